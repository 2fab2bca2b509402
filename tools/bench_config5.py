#!/usr/bin/env python
"""BASELINE.json configs[3] / configs[4] shaped rollout rates (reported in DESIGN.md, not bench lines):
    python tools/bench_config5.py [--grid 512] [--precision bf16] [--rollouts 8] [--hard] [--steps 30]
R concurrent rollouts on R procedural mazes, NBP forwards batched; prints exploration steps/s."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--rollouts", type=int, default=8)
    ap.add_argument("--hard", action="store_true")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    a = ap.parse_args()
    from nextbestpath_amd.networks.nbp_model import NBP
    from nextbestpath_amd.simulator import scene as sc
    from nextbestpath_amd.simulator.mesh import make_maze_scene
    from nextbestpath_amd.testers import nbp_planning as tp
    from nextbestpath_amd.utility.synthetic import make_explorer_state_dict
    dev = torch.device("cuda")
    params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
    net = NBP()
    net.load_state_dict(make_explorer_state_dict(9))
    net.conv_precision = a.precision
    net = net.to(dev).eval()
    tmp = tempfile.mkdtemp(prefix="nbp_c5_")
    for k in range(a.rollouts):
        if a.hard:
            make_maze_scene(os.path.join(tmp, f"maze{k}"), seed=200 + k, cells=12, size=7.2, height=1.2, tess=0.15)
        else:
            make_maze_scene(os.path.join(tmp, f"maze{k}"), seed=200 + k, cells=10, size=6.0, height=1.2, tess=0.25)
    ds = sc.SceneDataset(tmp)
    ros = [tp.build_rollout(params, net, ds, (k, 0), dev, seed=20 + k, grid=a.grid) for k in range(a.rollouts)]
    multi = tp.MultiRollout(ros, net, dev)
    for _ in range(a.warmup):
        multi.step()
    multi.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        multi.step()
    multi.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cov = [r.coverage_evolution(a.steps + a.warmup)[-1] for r in ros]
    print(json.dumps({"grid": a.grid, "precision": a.precision, "rollouts": a.rollouts, "faces": int(ros[0].mesh.faces.shape[0]),
                      "steps_per_s": round(a.steps * a.rollouts / dt, 2), "ms_per_lockstep": round(dt / a.steps * 1e3, 3),
                      "final_coverage_mean": round(float(np.mean(cov)), 4)}))


if __name__ == "__main__":
    main()
