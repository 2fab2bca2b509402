#!/usr/bin/env python
"""Per-forward wall time of N back-to-back forwards (event pairs): looks for rare long launches.
    python tools/diag/forward_jitter.py [--precision fp32_split] [--batch 8] [--n 300]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import packing  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_count_maps, make_nbp_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp32_split")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--n", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda")
pk = packing.pack_state_dict(make_nbp_state_dict(9), dev, precision=a.precision)
x = make_count_maps(a.batch, 256, seed=1).to(dev)
for _ in range(3):
    packing.forward_packed(pk, x)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.n + 1)]
ev[0].record()
for i in range(a.n):
    packing.forward_packed(pk, x)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.n))
print(f"{a.precision} B={a.batch}: n={a.n} min {ms[0]:.3f} median {ms[a.n // 2]:.3f} p99 {ms[int(a.n * 0.99)]:.3f} max {ms[-1]:.3f} ms")
