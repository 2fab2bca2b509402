#!/usr/bin/env python
"""bench.py -- NextBestPath hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--batch 1]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the per-step device hot path of one exploration rollout
(BASELINE.json configs[1]: AiMDoom_simple-like rollout, 256x256 grid, B=1, fp32):
the fused map accumulation over the rollout's accumulated point cloud + one NBP forward.
Inputs are synthetic and resident in HBM before the timed region.  Rollouts are independent
(SURVEY.md 8e), so N ranks run N independent rollouts: weak scaling, no data-path collective.

Prints ONE JSON line (rank 0) with the `roofline` object for the dominant kernel (the fp32
MFMA implicit-GEMM convolution) and the `cpu_baseline` object (the reference's arithmetic --
stock PyTorch CPU convolutions on the same weights -- timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_HBM_GBS = 8000.0
TILE_NAMES = {1: "igemm_conv_kernel<2,2,2,2>(128x128)", 2: "igemm_conv_kernel<4,1,2,2>(256x64)",
              3: "igemm_conv_kernel<4,1,2,1>(256x32)", 4: "igemm_conv_kernel<2,2,2,1>(128x64)",
              5: "igemm_conv_kernel<1,4,2,1>(64x128)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1, help="concurrent rollouts batched per forward on each GPU")
    ap.add_argument("--points", type=int, default=1_500_000, help="accumulated cloud size (mid-rollout)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="print the per-layer timing table to stderr")
    return ap.parse_args()


def timed_layers(net_packed, x, out1, out2, ws):
    from nextbestpath_amd import _lib
    L = _lib.lib()
    arr = (_lib.LayerTiming * 128)()
    n = C.c_int(0)
    B, _, S, _ = x.shape
    rc = L.nbp_forward_timed_f32(net_packed.handle, x.data_ptr(), B, S, out1.data_ptr(), out2.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _lib.current_stream(), arr, 128, C.byref(n))
    _lib.check(rc, "nbp_forward_timed_f32")
    return [dict(name=a.name.decode(), flops=a.flops, ms=a.ms, tile=a.tile, split_k=a.split_k, M=a.M, N=a.N, K=a.K)
            for a in arr[:n.value]]


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from nextbestpath_amd import _lib
    from nextbestpath_amd.networks import packing
    from nextbestpath_amd.networks.nbp_model import NBP
    from nextbestpath_amd.utility import utils as hu
    from nextbestpath_amd.utility.synthetic import make_count_maps, make_nbp_state_dict, make_point_cloud

    L = _lib.lib()
    S, B = args.size, args.batch
    sd = make_nbp_state_dict(9)
    net = NBP()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x = make_count_maps(B, S, seed=100 + rank).to(dev)
    pc = make_point_cloud(args.points, seed=200 + rank, extent=0.21 * S).to(dev)
    pose = torch.tensor([0.0, 13.3, 0.0, 0.0, 0.0])
    ybins = torch.arange(0.5, 29.5 + 7.25, 7.25)
    rng = (-40 * S / 256, 40 * S / 256)

    def step():
        maps = hu.accumulate_step_maps(pc, pose, ybins, S, rng)
        with torch.no_grad():
            o1, o2 = net(x)
        return maps, o1, o2

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel durations with HIP events on the launch stream (rank 0, N=1 semantics)
    roofline = None
    stage = {}
    layer_rows = []
    if rank == 0:
        packed = net._ensure_packed(dev)
        o1 = torch.empty(B, 8, S // 4, S // 4, device=dev)
        o2 = torch.empty(B, 1, S, S, device=dev)
        ws = packing._workspace(B, S, dev)
        reps = 5
        acc = {}
        for r in range(reps + 1):
            rows = timed_layers(packed, x, o1, o2, ws)
            if r == 0:
                continue        # warm
            for i, row in enumerate(rows):
                a = acc.setdefault(i, dict(row, ms=0.0))
                a["ms"] += row["ms"] / reps
        layer_rows = [acc[i] for i in sorted(acc)]
        by_tile = {}
        for row in layer_rows:
            if row["tile"] > 0:
                t = by_tile.setdefault(row["tile"], dict(flops=0.0, ms=0.0, launches=0))
                t["flops"] += row["flops"]; t["ms"] += row["ms"]; t["launches"] += 1
        dom = max(by_tile, key=lambda k: by_tile[k]["flops"])
        d = by_tile[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        conv_flops = sum(r["flops"] for r in layer_rows if r["tile"] > 0)
        conv_ms = sum(r["ms"] for r in layer_rows if r["tile"] > 0)
        roofline = {"bound": "mfma", "kernel": TILE_NAMES[dom], "achieved": round(achieved, 3),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": None, "launches_per_forward": d["launches"],
                    "avg_launch_ms": round(d["ms"] / d["launches"], 5),
                    "flops_per_launch": d["flops"] / d["launches"],
                    "all_igemm_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 3),
                    "all_igemm_frac": round(conv_flops / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        # scatter kernel: HIP events around the fused map accumulation
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hu.accumulate_step_maps(pc, pose, ybins, S, rng)
        e0.record()
        for _ in range(20):
            hu.accumulate_step_maps(pc, pose, ybins, S, rng)
        e1.record(); torch.cuda.synchronize()
        ms_scatter = e0.elapsed_time(e1) / 20
        alg_bytes = 12 * args.points + 6 * S * S * 4
        stage["map_accumulate"] = {"ms": round(ms_scatter, 4), "points": args.points,
                                   "algorithmic_bytes": alg_bytes,
                                   "achieved_GBps": round(alg_bytes / (ms_scatter * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(alg_bytes / (ms_scatter * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        e0.record()
        for _ in range(20):
            with torch.no_grad():
                net(x)
        e1.record(); torch.cuda.synchronize()
        ms_fwd = e0.elapsed_time(e1) / 20
        fl = L.nbp_forward_flops(B, S)
        stage["nbp_forward"] = {"ms": round(ms_fwd, 4), "maps_per_s": round(B / (ms_fwd * 1e-3), 2),
                                "tflops": round(fl / (ms_fwd * 1e-3) / 1e12, 3),
                                "frac_of_f32_mfma_peak": round(fl / (ms_fwd * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        if args.layers:
            for r in layer_rows:
                tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
                print(f'{r["name"]:24s} M={r["M"]:7d} N={r["N"]:5d} K={r["K"]:5d} tile={r["tile"]:2d} '
                      f'sk={r["split_k"]:2d} {r["ms"]*1e3:9.1f} us {tf:7.2f} TF', file=sys.stderr)

    # ---- CPU baseline: the reference's arithmetic (stock PyTorch CPU convs) on the host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import maps as omaps
        from oracle import nbp_net
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        xc = x[:1].cpu()
        # pick the thread count that is fastest on this host (all logical CPUs oversubscribes badly)
        best = None
        with torch.no_grad():
            for nt in sorted({min(avail, c) for c in (8, 16, 32, 64, 128)}):
                torch.set_num_threads(nt)
                nbp_net.nbp_forward(sd, xc)
                t0c = time.perf_counter()
                nbp_net.nbp_forward(sd, xc)
                tt = time.perf_counter() - t0c
                if best is None or tt < best[1]:
                    best = (nt, tt)
            cores = best[0]
            torch.set_num_threads(cores)
            n_it, t0c = 0, time.perf_counter()
            while time.perf_counter() - t0c < 10.0 and n_it < 50:
                nbp_net.nbp_forward(sd, xc)
                n_it += 1
            cpu_fwd = (time.perf_counter() - t0c) / n_it
        sub = pc[:200_000].cpu().numpy()
        t0c = time.perf_counter()
        omaps.accumulate_step_maps(sub, pose.numpy(), ybins.numpy(), S, rng)
        cpu_map = (time.perf_counter() - t0c) * (args.points / 200_000)
        cpu = {"value": round(1.0 / (cpu_fwd + cpu_map), 3), "unit": "steps/s", "cores": cores, "kind": "port",
               "sample": f"{n_it} NBP forwards at {S}x{S} B=1 (torch CPU convs, {cores} threads: "
                         f"{cpu_fwd*1e3:.1f} ms each) + numpy map accumulation of 200k points scaled to "
                         f"{args.points} ({cpu_map*1e3:.1f} ms)",
               "nbp_maps_per_s": round(1.0 / cpu_fwd, 3)}

    if rank == 0:
        steps_total = args.steps * world * B
        out = {
            "metric": "exploration hot-path steps/s (map accumulation + NBP forward, 256x256)",
            "value": round(steps_total / dt, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: AiMDoom_simple-like rollout step, 256x256 grid, 1 rollout/GPU, "
                                   "seeded synthetic NBP weights + synthetic point cloud",
                       "grid": S, "rollouts_per_gpu": B, "cloud_points": args.points},
            "nbp_maps_per_s": round(world * stage["nbp_forward"]["maps_per_s"], 2) if stage else None,
            "stages": stage, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
