#!/usr/bin/env python
"""Map accumulation: append-order kernel (LDS hash per 8192 points) vs the tile-binned build (utils.CloudBins), on a wall-like
cloud: us per build (steady state: everything filed), us for filing one step's ~29 k new points, equality of the maps.
    python tools/diag/map_bins_ab.py [--points 2300000]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.utility import utils as hu  # noqa: E402


def walls(n, seed, extent=30.0):
    rng = np.random.default_rng(seed)
    k = 40
    a = rng.uniform(-extent, extent, (k, 2)).astype(np.float32)
    d = rng.uniform(-1, 1, (k, 2)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    w = rng.integers(0, k, n)
    s = rng.uniform(0, 20, n).astype(np.float32)
    xz = a[w] + d[w] * s[:, None] + rng.normal(0, 0.02, (n, 2)).astype(np.float32)
    y = rng.uniform(0, 12, n).astype(np.float32)
    return torch.from_numpy(np.stack([xz[:, 0], y, xz[:, 1]], 1).astype(np.float32))


def ev(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="+", default=[1_300_000, 2_300_000, 3_000_000])
    a = ap.parse_args()
    ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
    pose = torch.tensor([1.0, 5.05, -2.0, 0, 0])
    for n in a.points:
        pc = walls(n + 8 * 29_180, seed=1).cuda().contiguous()
        n_dev = torch.tensor([n], dtype=torch.int64, device="cuda")
        bins = hu.CloudBins((-52.0, -52.0), (52.0, 52.0), pc.shape[0], "cuda")
        out_a, out_b = torch.empty(6, 256, 256, device="cuda"), torch.empty(6, 256, 256, device="cuda")
        hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out_a)
        hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out_b, bins=bins)
        same = bool(torch.equal(out_a, out_b))
        us_a = ev(lambda: hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out_a))
        us_b = ev(lambda: hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out_b, bins=bins))
        # builds that each meet one step's worth of new points (29180: five frames' kept pixels), as a rollout's do
        evs = []
        for k in range(8):
            n_dev.add_(29_180)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out_b, bins=bins)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        us_new = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)[len(evs) // 2]
        hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out_a)
        same2 = bool(torch.equal(out_a, out_b))
        alg = 12 * n + 24 * 256 * 256
        h = bins.header()
        print(f"N={n}: append-order {us_a:.1f} us ({alg / us_a / 1e3:.0f} GB/s)  binned {us_b:.1f} us ({alg / us_b / 1e3:.0f} GB/s, "
              f"{alg / us_b / 1e3 / 8000:.3f} of 8 TB/s)  build meeting 29180 new points {us_new:.1f} us (median of 8)  equal={same},{same2}  "
              f"pages={h['n_pages']} overflow={h['n_overflow']}", flush=True)


if __name__ == "__main__":
    main()
