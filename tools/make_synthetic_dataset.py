#!/usr/bin/env python
"""Writes a seeded procedural stand-in for an AiMDoom split (no dataset is available offline):
    python tools/make_synthetic_dataset.py --out data/AiMDoom_synth_simple --scenes 8 [--hard]
Each scene directory holds <name>.obj + settings.json in the reference's schema."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextbestpath_amd.simulator.mesh import make_maze_scene  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="data/AiMDoom_synth_simple")
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--hard", action="store_true", help="bigger mazes, finer tessellation (20-50 k faces)")
    ap.add_argument("--starts", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hull", choices=["slab", "shell"], default="slab",
                    help="shell: one closed surface around the free space (training scenes: check_camera_in_mesh)")
    a = ap.parse_args()
    for i in range(a.scenes):
        d = os.path.join(a.out, f"maze_{i:02d}")
        if a.hard:
            make_maze_scene(d, seed=a.seed + i, cells=12, size=7.2, height=1.2, tess=0.15, n_starts=a.starts, hull=a.hull)
        else:
            make_maze_scene(d, seed=a.seed + i, cells=10, size=6.0, height=1.2, tess=0.25, n_starts=a.starts, hull=a.hull)
        print(d)
