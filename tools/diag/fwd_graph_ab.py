#!/usr/bin/env python
"""Eval forward eager vs replayed hipGraph (packing.ForwardGraph): ms per forward at the given batches, bit equality of the outputs.
    python tools/diag/fwd_graph_ab.py [--batches 1 2 24]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import packing  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_count_maps, make_explorer_state_dict  # noqa: E402


def ev_time(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 24])
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="fp32_split")
    a = ap.parse_args()
    dev = torch.device("cuda")
    packed = packing.pack_state_dict(make_explorer_state_dict(9), dev, precision=a.precision)
    for B in a.batches:
        x = make_count_maps(B, a.size, seed=B).to(dev)
        e1, e2 = packing.forward_packed(packed, x)
        g = packing.ForwardGraph(packed, x)
        g1, g2 = g()
        torch.cuda.synchronize()
        same = bool(torch.equal(e1, g1) and torch.equal(e2, g2))
        reps = 200 if B <= 4 else 30
        ms_e = ev_time(lambda: packing.forward_packed(packed, x), reps)
        ms_g = ev_time(g, reps)
        # interleaved repeats: clock / power state must not decide the comparison
        rep2 = [(ev_time(lambda: packing.forward_packed(packed, x), reps), ev_time(g, reps)) for _ in range(3)]
        print("   repeats (eager, graph): " + "  ".join(f"({a:.4f}, {b:.4f})" for a, b in rep2), flush=True)
        # wall clock including the host's enqueue time (what a step loop sees when nothing else overlaps)
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            packing.forward_packed(packed, x)
        t_host_e = (time.perf_counter() - t0) / reps * 1e3
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            g()
        t_host_g = (time.perf_counter() - t0) / reps * 1e3
        torch.cuda.synchronize()
        print(f"B={B:3d} S={a.size}: eager {ms_e:.4f} ms  graph {ms_g:.4f} ms  ({ms_e / ms_g:.3f}x)  bit-identical={same}  "
              f"host enqueue per forward: eager {t_host_e:.3f} ms, graph {t_host_g:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
