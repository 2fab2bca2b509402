#!/usr/bin/env python
"""Whole-network forward timing through nbp_forward_timed_{f32,split_f32,bf16}: per-layer table + totals.
    python tools/bench_forward.py [--bf16 | --split] [--batch 8] [--size 512] [--reps 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextbestpath_amd import _lib  # noqa: E402
from nextbestpath_amd.networks import packing  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_count_maps, make_nbp_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--split", action="store_true", help="fp32 tensors, 3x3 layers as three exact fp16 MFMAs per product (two-piece operands)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda")
    prec = "bf16" if a.bf16 else ("fp32_split" if a.split else "fp32")
    packed = packing.pack_state_dict(make_nbp_state_dict(9), dev, precision=prec)
    B, S = a.batch, a.size
    x = make_count_maps(B, S, seed=1).to(dev)
    o1 = torch.empty(B, 8, S // 4, S // 4, device=dev)
    o2 = torch.empty(B, 1, S, S, device=dev)
    nws = getattr(L, packing._FWD[prec][1])(B, S)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fwd = getattr(L, packing._FWD[prec][0])
    timed = getattr(L, {"fp32": "nbp_forward_timed_f32", "bf16": "nbp_forward_timed_bf16", "fp32_split": "nbp_forward_timed_split_f32"}[prec])

    def run():
        _lib.check(fwd(packed.handle, x.data_ptr(), B, S, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(),
                       _lib.current_stream()), "forward")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    fl = L.nbp_forward_flops(B, S)
    acc = {}
    for _ in range(a.reps):
        arr = (_lib.LayerTiming * 128)()
        n = C.c_int(0)
        _lib.check(timed(packed.handle, x.data_ptr(), B, S, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(),
                         _lib.current_stream(), arr, 128, C.byref(n)), "timed")
        for i, t in enumerate(arr[:n.value]):
            r = acc.setdefault(i, dict(name=t.name.decode(), flops=t.flops, ms=[], tile=t.tile, split=t.split_k, M=t.M,
                                       N=t.N, K=t.K))
            r["ms"].append(t.ms)
    if not a.quiet:
        for i in sorted(acc):
            r = acc[i]
            m = sorted(r["ms"])[len(r["ms"]) // 2]
            tf = r["flops"] / m / 1e9 if m > 0 else 0
            print(f"{r['name']:28s} M={r['M']:8d} N={r['N']:5d} K={r['K']:5d} tile={r['tile']:2d} split={r['split']:2d} "
                  f"{m*1e3:9.1f} us {tf:8.1f} TF")
    igemm = sum(sorted(r["ms"])[len(r["ms"]) // 2] for r in acc.values() if r["tile"] > 0)
    other = sum(sorted(r["ms"])[len(r["ms"]) // 2] for r in acc.values() if r["tile"] <= 0)
    print(f"{prec} B={B} S={S}: {ms:.3f} ms/forward  {B/ms*1e3:.1f} maps/s  {fl/ms/1e9:.1f} TF "
          f"(event-bracketed: igemm {igemm:.3f} ms, other kernels {other:.3f} ms; workspace {nws/2**20:.0f} MiB)")


if __name__ == "__main__":
    main()
