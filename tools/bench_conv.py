#!/usr/bin/env python
"""Micro-benchmark of single NBP conv layers through nbp_conv_igemm_f32 (HIP events).
    python tools/bench_conv.py [--tile T] [--split K] [--batch B]
Prints time / TFLOP/s per representative layer shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextbestpath_amd import _lib  # noqa: E402

# name, H(=W) of output, C0, C1, N, ksize, ups
SHAPES = [("Conv1.3", 256, 64, 0, 64, 3, 0), ("Conv2.0", 128, 64, 0, 128, 3, 0), ("Conv2.3", 128, 128, 0, 128, 3, 0),
          ("Conv3.3", 64, 256, 0, 256, 3, 0), ("Conv4.3", 32, 512, 0, 512, 3, 0), ("Conv5.0", 16, 512, 0, 1024, 3, 0),
          ("Conv5.3", 16, 1024, 0, 1024, 3, 0), ("Up5", 32, 1024, 0, 512, 3, 1), ("Upc5.0", 32, 512, 512, 512, 3, 0),
          ("Up4", 64, 512, 0, 256, 3, 1), ("Up3", 128, 256, 0, 128, 3, 1), ("Up2", 256, 128, 0, 64, 3, 1),
          ("Upc2.0", 256, 64, 64, 64, 3, 0), ("Att5", 32, 512, 512, 256, 1, 0), ("Att2", 256, 64, 64, 32, 1, 0)]


def sweep(a):
    L = _lib.lib()
    dev = "cuda"
    for name, H, C0, C1, N, k, ups in SHAPES:
        if a.only and a.only not in name:
            continue
        B = a.batch
        Hs = H // 2 if ups else H
        s0 = torch.randn(B, Hs, Hs, C0, device=dev)
        s1 = torch.randn(B, Hs, Hs, C1, device=dev) if C1 else None
        wpk = torch.randn((C0 + C1) // 32 * k * k * N * 32, device=dev) * 0.02
        sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        out = torch.empty(B, H, H, N, device=dev)
        ws = torch.empty(max(L.nbp_conv_igemm_workspace_bytes(B, H, H, N, 64), 256), dtype=torch.uint8, device=dev)
        chunks = (C0 + C1) // 32 * k * k
        cfgs = [(0, 0)] + [(t, sk) for t in (1, 2, 3, 4, 5, 6, 7, 8, 9) for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32) if sk <= chunks]
        res = {}
        for rnd in range(3):
            for cfg in cfgs:
                def run():
                    return L.nbp_conv_igemm_f32(s0.data_ptr(), C0, _lib.ptr(s1), C1, ups, B, H, H, k, wpk.data_ptr(), N,
                                                sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), cfg[1], cfg[0],
                                                ws.data_ptr(), ws.numel(), _lib.current_stream())
                if run() != 0:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(cfg, []).append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * B * H * H * N * (C0 + C1) * k * k
        best = sorted((min(v), c) for c, v in res.items())
        auto = min(res[(0, 0)])
        txt = "  ".join(f"t{c[0]}/s{c[1]}:{t*1e3:.1f}" for t, c in best[:5])
        print(f"{name:8s} M={B*H*H:6d} N={N:4d} K={(C0+C1)*k*k:5d} auto {auto*1e3:7.1f} us ({fl/auto/1e9:6.1f} TF) | best {txt}")


def bf16(a):
    L = _lib.lib()
    dev = "cuda"
    tot_t = tot_f = 0.0
    for name, H, C0, C1, N, k, ups in SHAPES:
        if a.only and a.only not in name:
            continue
        if (C0 % 64) or (C1 % 64):
            continue
        B = a.batch
        Hs = H // 2 if ups else H
        s0 = torch.randn(B, Hs, Hs, C0, device=dev).to(torch.bfloat16)
        s1 = torch.randn(B, Hs, Hs, C1, device=dev).to(torch.bfloat16) if C1 else None
        wpk = (torch.randn((C0 + C1) // 64 * k * k * N * 64, device=dev) * 0.02).to(torch.bfloat16)
        sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        out = torch.empty(B, H, H, N, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(max(L.nbp_conv_igemm_bf16_workspace_bytes(B, H, H, N, 64), 256), dtype=torch.uint8, device=dev)
        chunks = (C0 + C1) // 64 * k * k
        cfgs = [(a.tile, a.split)]
        if a.sweep:
            cfgs = [(0, 0)] + [(t, sk) for t in (1, 2, 3, 4, 5, 6, 7) for sk in (1, 2, 3, 4, 6, 8, 16) if sk <= chunks]
        res = {}
        for rnd in range(3 if a.sweep else 1):
            for cfg in cfgs:
                def run():
                    return L.nbp_conv_igemm_bf16(s0.data_ptr(), C0, _lib.ptr(s1), C1, ups, B, H, H, k, wpk.data_ptr(), N,
                                                 sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), cfg[1], cfg[0],
                                                 ws.data_ptr(), ws.numel(), _lib.current_stream())
                if run() != 0:
                    continue
                for _ in range(3):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(cfg, []).append(e0.elapsed_time(e1) / a.reps)
        fl = 2.0 * B * H * H * N * (C0 + C1) * k * k
        first = min(res[cfgs[0]])
        tot_t += first; tot_f += fl
        best = sorted((min(v), c) for c, v in res.items())
        txt = "  ".join(f"t{c[0]}/s{c[1]}:{t*1e3:.1f}" for t, c in best[:5]) if a.sweep else ""
        print(f"{name:8s} M={B*H*H:7d} N={N:4d} K={(C0+C1)*k*k:5d}  {first*1e3:8.1f} us  {fl/first/1e9:7.1f} TF  {txt}")
    if tot_t:
        print(f"TOTAL {tot_t*1e3:.1f} us  {tot_f/tot_t/1e9:.2f} TF")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--sweep", action="store_true", help="try every (tile, split) and report the best per shape")
    ap.add_argument("--bf16", action="store_true", help="the bf16 kernels (nbp_conv_igemm_bf16)")
    ap.add_argument("--size", type=int, default=256, help="grid size S the shapes are scaled to (256 or 512)")
    a = ap.parse_args()
    global SHAPES
    SHAPES = [(n, H * a.size // 256, C0, C1, N, k, u) for n, H, C0, C1, N, k, u in SHAPES]
    if a.bf16:
        return bf16(a)
    if a.sweep:
        return sweep(a)
    L = _lib.lib()
    dev = "cuda"
    tot_t = tot_f = 0.0
    for name, H, C0, C1, N, k, ups in SHAPES:
        if a.only and a.only not in name:
            continue
        B = a.batch
        Hs = H // 2 if ups else H
        s0 = torch.randn(B, Hs, Hs, C0, device=dev)
        s1 = torch.randn(B, Hs, Hs, C1, device=dev) if C1 else None
        wpk = torch.randn((C0 + C1) // 32 * k * k * N * 32, device=dev) * 0.02
        sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        out = torch.empty(B, H, H, N, device=dev)
        ws = torch.empty(max(L.nbp_conv_igemm_workspace_bytes(B, H, H, N, a.split), 256), dtype=torch.uint8, device=dev)

        def run():
            rc = L.nbp_conv_igemm_f32(s0.data_ptr(), C0, _lib.ptr(s1), C1, ups, B, H, H, k, wpk.data_ptr(), N,
                                      sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), a.split, a.tile, ws.data_ptr(),
                                      ws.numel(), _lib.current_stream())
            assert rc == 0, rc
        rc0 = L.nbp_conv_igemm_f32(s0.data_ptr(), C0, _lib.ptr(s1), C1, ups, B, H, H, k, wpk.data_ptr(), N, sc.data_ptr(),
                                   sh.data_ptr(), 1, out.data_ptr(), a.split, a.tile, ws.data_ptr(), ws.numel(),
                                   _lib.current_stream())
        if rc0 != 0:
            print(f"{name:8s} unsupported with tile {a.tile} (rc {rc0})")
            continue
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        fl = 2.0 * B * H * H * N * (C0 + C1) * k * k
        tot_t += ms; tot_f += fl
        print(f"{name:8s} M={B*H*H:6d} N={N:4d} K={(C0+C1)*k*k:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TF")
    if tot_t:
        print(f"TOTAL {tot_t*1e3:.1f} us  {tot_f/tot_t/1e9:.2f} TF")


if __name__ == "__main__":
    main()
