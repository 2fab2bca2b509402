#!/usr/bin/env python
"""bench.py -- NextBestPath exploration hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one exploration step of one rollout (S1-S14 of SURVEY.md section 3.1; BASELINE.json
configs[1]: AiMDoom_simple-like scene, 256x256 grid, B=1, fp32): coverage of the cloud so far,
un-projection of 5 depth frames (256x456) into the cloud, fused map accumulation, ONE NBP forward,
replanning when the path is exhausted / blocked, 4 rasterised frames along the move.  The scene is
a seeded procedural maze (no AiMDoom data offline), the NBP weights are seeded synthetic; mesh,
weights and buffers are resident in HBM before the timed region.  Rollouts are independent
(SURVEY.md 8e): N ranks run N rollouts on N scenes -> weak scaling, no data-path collective.

Rank 0 prints ONE JSON line.  `value` = exploration steps/s over all ranks; `nbp_maps_per_s` = NBP
forwards/s (the second half of BASELINE.json's metric); `roofline` = the dominant kernel (fp32 MFMA
implicit-GEMM convolution) timed live with HIP events on the launch stream; `roofline_scatter` =
the HBM-bound map accumulation; `cpu_baseline` = the reference's arithmetic on the host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_HBM_GBS = 8000.0
TILE_NAMES = {1: "igemm_conv_kernel<2,2,2,2>(128x128)", 2: "igemm_conv_kernel<4,1,2,2>(256x64)",
              3: "igemm_conv_kernel<4,1,2,1>(256x32)", 4: "igemm_conv_kernel<2,2,2,1>(128x64)",
              5: "igemm_conv_kernel<1,4,2,1>(64x128)", 6: "conv3x3_halo_f32_kernel<4,2>(8x32 px x 128 ch)",
              7: "conv3x3_halo_f32_kernel<2,2>(8x32 px x 64 ch)", 8: "conv3x3_halo_f32_kernel<4,1>(4x32 px x 128 ch)",
              9: "conv3x3_halo_f32_kernel<2,1>(4x32 px x 64 ch)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="per-layer timing table to stderr")
    ap.add_argument("--faces", choices=["simple", "hard"], default="simple")
    ap.add_argument("--rollouts-per-gpu", type=int, default=8,
                    help="independent rollouts stepped in lock-step per GPU (their NBP forwards are one batched launch)")
    return ap.parse_args()


def timed_layers(packed, x, out1, out2, ws):
    from nextbestpath_amd import _lib
    arr = (_lib.LayerTiming * 128)()
    n = C.c_int(0)
    B, _, S, _ = x.shape
    rc = _lib.lib().nbp_forward_timed_f32(packed.handle, x.data_ptr(), B, S, out1.data_ptr(), out2.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.current_stream(), arr, 128, C.byref(n))
    _lib.check(rc, "nbp_forward_timed_f32")
    return [dict(name=a.name.decode(), flops=a.flops, ms=a.ms, tile=a.tile, split_k=a.split_k, M=a.M, N=a.N, K=a.K)
            for a in arr[:n.value]]


def pmc_traffic(kernel_prefix):
    """HBM-side bytes per launch of `kernel_prefix` from the committed rocprofv3 PMC passes over the same workload
    (tools/profile.sh -> profiles/r01/forward_f32_pmc_summary.csv): 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; the
    factor 2 is the gfx950 correction for 16-B-per-lane loads, MI355X_MICROARCH.md).  None if no profile."""
    import csv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "forward_f32_pmc_summary.csv")
    if not os.path.exists(path):
        return None
    want = kernel_prefix.replace(" ", "")
    vals = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["kernel"].replace(" ", "").startswith(want):
                vals[row["counter"]] = float(row["mean_per_dispatch"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def ev_time(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    backend = os.environ.get("NBP_DIST_BACKEND", "nccl")      # "gloo" only to smoke-test N > 1 on a 1-GPU box
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from nextbestpath_amd import _lib
    from nextbestpath_amd.networks import packing
    from nextbestpath_amd.networks.nbp_model import NBP
    from nextbestpath_amd.simulator import scene as sc
    from nextbestpath_amd.simulator.mesh import make_maze_scene
    from nextbestpath_amd.testers import nbp_planning as tp
    from nextbestpath_amd.utility import hipops
    from nextbestpath_amd.utility import utils as hu
    from nextbestpath_amd.utility.synthetic import make_explorer_state_dict

    L = _lib.lib()
    S = 256
    params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
    tmp = tempfile.mkdtemp(prefix=f"nbp_bench_r{rank}_")
    R = max(1, args.rollouts_per_gpu)
    sd = make_explorer_state_dict(9)     # silent obstacle head: the observed walls do the blocking
    net = NBP()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()

    def make_rollout(k):
        name = f"maze{k}"
        if args.faces == "hard":
            make_maze_scene(os.path.join(tmp, name), seed=100 + 16 * rank + k, cells=12, size=7.2, height=1.2, tess=0.15)
        else:
            make_maze_scene(os.path.join(tmp, name), seed=100 + 16 * rank + k, cells=10, size=6.0, height=1.2, tess=0.25)
        ds = sc.SceneDataset(tmp, [name])
        settings = sc.Settings(ds[0]["settings"], params.scene_scale_factor)
        mesh = sc.load_scene(os.path.join(tmp, name, ds[0]["obj_name"]), params.scene_scale_factor, dev)
        y_bins = sc.y_bins_for(mesh.verts_host, 4)
        _, gt = sc.setup_gt_scene(params, settings, mesh, dev, 0.05, seed=rank)
        cam = tp.setup_test_camera(params, mesh, settings.camera.start_positions[0], settings, dev, seed=rank)
        return tp.Rollout(params, net, cam, gt, mesh, mesh, y_bins, dev, seed=8 + 16 * rank + k)

    # ---- single-rollout rate (B = 1 forward, latency-style) on a short separate run, reported beside `value`
    single = None
    if rank == 0:
        r1 = make_rollout(15)
        for _ in range(5):
            r1.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r1.step()
        torch.cuda.synchronize()
        single = 20 / (time.perf_counter() - t0)
        del r1
    rollouts = [make_rollout(k) for k in range(R)]
    multi = tp.MultiRollout(rollouts, net, dev)
    ro, cam, mesh, y_bins, gt = rollouts[0], rollouts[0].camera, rollouts[0].mesh, rollouts[0].y_bins, rollouts[0].gt

    for _ in range(args.warmup):
        multi.step()
    multi.flush()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    replans0 = sum(r.n_replans for r in rollouts)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        multi.step()
    multi.flush()                      # every rollout has completed exactly `steps` exploration steps
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    roofline = scatter = None
    stage = {}
    if rank == 0:
        # ---- the dominant kernel, per launch, with HIP events on the launch stream
        x = torch.cat(multi.net_in)
        packed = net._ensure_packed(dev)
        o1 = torch.empty(R, 8, S // 4, S // 4, device=dev)
        o2 = torch.empty(R, 1, S, S, device=dev)
        ws = packing._workspace(R, S, dev)
        reps, acc = 5, {}
        for r in range(reps + 1):
            rows = timed_layers(packed, x, o1, o2, ws)
            if r:
                for i, row in enumerate(rows):
                    acc.setdefault(i, dict(row, ms=0.0))["ms"] += row["ms"] / reps
        layer_rows = [acc[i] for i in sorted(acc)]
        by_tile = {}
        for row in layer_rows:
            if row["tile"] > 0:
                t = by_tile.setdefault(row["tile"], dict(flops=0.0, ms=0.0, launches=0))
                t["flops"] += row["flops"]; t["ms"] += row["ms"]; t["launches"] += 1
        dom = max(by_tile, key=lambda k: by_tile[k]["flops"])
        d = by_tile[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        cf = sum(r["flops"] for r in layer_rows if r["tile"] > 0)
        cm = sum(r["ms"] for r in layer_rows if r["tile"] > 0)
        # algorithmic bytes of those launches: sources + packed weights + output, each once
        def layer_bytes(r):
            taps = 1 if ".W_g" in r["name"] else 9
            src = r["M"] * (r["K"] // taps) / (4 if ".up.1" in r["name"] else 1)     # fused x2 upsample reads H/2 x W/2
            return 4.0 * (src + r["K"] * r["N"] + r["M"] * r["N"]) * (2 if "{1,2}" in r["name"] else 1)
        alg_bytes = sum(layer_bytes(r) for r in layer_rows if r["tile"] == dom)
        traffic = pmc_traffic(TILE_NAMES[dom].split("(")[0]) if (R, S) == (8, 256) else None
        roofline = {"bound": "mfma", "kernel": TILE_NAMES[dom], "achieved": round(achieved, 3),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, "
                    "profiles/r01/forward_f32_pmc_summary.csv, same B=8 S=256 forward)",
                    "algorithmic_bytes_per_launch": round(alg_bytes / d["launches"]),
                    "launches_per_forward": d["launches"],
                    "avg_launch_ms": round(d["ms"] / d["launches"], 5), "flops_per_launch": d["flops"] / d["launches"],
                    "note": "avg_launch_ms: HIP event pair per layer on the launch stream (a split-K layer includes its reduce)",
                    "all_igemm_tflops": round(cf / (cm * 1e-3) / 1e12, 3),
                    "all_igemm_frac": round(cf / (cm * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        if args.layers:
            for r in layer_rows:
                tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
                print(f'{r["name"]:24s} M={r["M"]:7d} N={r["N"]:5d} K={r["K"]:5d} tile={r["tile"]:2d} '
                      f'sk={r["split_k"]:2d} {r["ms"]*1e3:9.1f} us {tf:7.2f} TF', file=sys.stderr)
        # ---- stage timings on the state the rollout reached
        n_pts = int(ro.st.cloud_count.item())
        pose, _ = cam.get_pose_from_idx(cam.cam_idx)
        ms_fwd = ev_time(lambda: net(x))
        fl = L.nbp_forward_flops(R, S)
        x1 = x[:1].contiguous()
        ms_fwd1 = ev_time(lambda: net(x1))
        stage["nbp_forward_b1"] = {"ms": round(ms_fwd1, 4), "maps_per_s": round(1e3 / ms_fwd1, 2),
                                   "tflops": round(L.nbp_forward_flops(1, S) / (ms_fwd1 * 1e-3) / 1e12, 3)}
        stage["nbp_forward"] = {"ms": round(ms_fwd, 4), "batch": R, "maps_per_s": round(R * 1e3 / ms_fwd, 2),
                                "tflops": round(fl / (ms_fwd * 1e-3) / 1e12, 3),
                                "frac_of_f32_mfma_peak": round(fl / (ms_fwd * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        ms_sc = ev_time(lambda: hu.accumulate_step_maps(ro.st.cloud, pose, y_bins, S, (-40, 40), n_dev=ro.st.cloud_count,
                                                        out=ro.st.maps6))
        alg = 12 * n_pts + 6 * S * S * 4
        scatter = {"bound": "hbm", "kernel": "map_accumulate_kernel", "achieved": round(alg / (ms_sc * 1e-3) / 1e9, 1),
                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(alg / (ms_sc * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                   "traffic": None, "points": n_pts, "algorithmic_bytes": alg, "ms": round(ms_sc, 4)}
        cams4 = np.stack([f[1] for f in cam.frames[-4:]])
        zb = torch.empty(4, params.image_height, params.image_width, device=dev)
        ms_r = ev_time(lambda: hipops.raster_zbuf(mesh.verts, mesh.faces, cams4, params.image_height, params.image_width,
                                                  bin_cap=mesh.bin_cap, out=zb))
        stage["raster_4_frames"] = {"ms": round(ms_r, 4), "faces": int(mesh.faces.shape[0]),
                                    "frames_per_s": round(4e3 / ms_r, 1)}
        scratch = torch.zeros(200_000, 3, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)

        def unproj():
            cnt.zero_()
            hipops.unproject_append(zb, None, cams4, scratch, cnt, 0.05, 70.0, seed=1)
        stage["unproject_4_frames"] = {"ms": round(ev_time(unproj), 4)}
        bbox = (gt.min(0).values.tolist(), gt.max(0).values.tolist())
        out = torch.zeros(2, dtype=torch.int32, device=dev)
        stage["coverage"] = {"ms": round(ev_time(lambda: hipops.coverage_count(gt, ro.st.cloud, n_dev=ro.st.cloud_count,
                                                                                n=ro.st.cloud.shape[0], bbox=bbox,
                                                                                out=out)), 4),
                             "gt_points": int(gt.shape[0])}
        # BASELINE configs[4] forward (reported beside the headline, not part of `value`): 8 maps of 512x512
        # through the bf16 network; fraction of the dense bf16 MFMA peak (2.5 PFLOP/s)
        sd16 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        pk16 = packing.pack_state_dict(sd16, dev, bf16=True)
        x5 = torch.zeros(8, 5, 512, 512, device=dev)
        x5[:, :, 128:384, 128:384] = x[:1].expand(8, -1, -1, -1) if S == 256 else 0.0
        ms16 = ev_time(lambda: packing.forward_packed(pk16, x5), reps=10)
        fl16 = L.nbp_forward_flops(8, 512)
        stage["config5_forward_bf16_512_b8"] = {"ms": round(ms16, 4), "maps_per_s": round(8e3 / ms16, 2),
                                                "tflops": round(fl16 / (ms16 * 1e-3) / 1e12, 2),
                                                "frac_of_bf16_mfma_peak": round(fl16 / (ms16 * 1e-3) / 1e12 / 2500.0, 4)}
        pk16.free()
        del x5, pk16
        stage["replans_in_timed_region"] = sum(r.n_replans for r in rollouts) - replans0
        stage["single_rollout_steps_per_s"] = round(single, 2)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import maps as omaps
        from oracle import nbp_net
        from oracle import planner as opl
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        xc = multi.net_in[0][:1].cpu()
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
            while time.perf_counter() - t0c < 8.0 and n_it < 50:
                nbp_net.nbp_forward(sd, xc)
                n_it += 1
            cpu_fwd = (time.perf_counter() - t0c) / n_it
        n_pts = int(ro.st.cloud_count.item())
        sub = ro.st.cloud[:min(n_pts, 200_000)].cpu().numpy()
        t0c = time.perf_counter()
        omaps.accumulate_step_maps(sub, pose, y_bins.numpy(), S, (-40, 40))
        cpu_map = (time.perf_counter() - t0c) * (n_pts / max(len(sub), 1))
        gts = gt[:2000].cpu().numpy()
        t0c = time.perf_counter()
        opl.coverage(gts, ro.st.cloud[:min(n_pts, 100_000)].cpu().numpy())
        cpu_cov = (time.perf_counter() - t0c) * (gt.shape[0] / 2000)
        cpu = {"value": round(1.0 / (cpu_fwd + cpu_map + cpu_cov), 3), "unit": "steps/s", "cores": cores,
               "kind": "port",
               "sample": f"{n_it} NBP forwards at 256x256 B=1 with stock PyTorch CPU convs = the reference's own "
                         f"arithmetic ({cores} threads, {cpu_fwd*1e3:.1f} ms each) + numpy map accumulation of "
                         f"{len(sub)} points scaled to {n_pts} ({cpu_map*1e3:.1f} ms) + brute-force coverage of 2000 GT "
                         f"points scaled to {gt.shape[0]} ({cpu_cov*1e3:.0f} ms); rendering/un-projection excluded "
                         f"(PyTorch3D is not installable here), so this is an upper bound on the CPU step rate",
               "nbp_maps_per_s": round(1.0 / cpu_fwd, 3)}

    if rank == 0:
        out = {
            "metric": "exploration steps/s (+ NBP maps/s) at 256x256", "value": round(args.steps * world * R / dt, 3),
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "note": f"ms_per_step is one lock-step step of {R} concurrent rollouts per GPU ({R} exploration steps)",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: AiMDoom_simple-like rollout (seeded procedural maze, "
                                   f"{int(mesh.faces.shape[0])} faces), 256x256 grid, {R} concurrent rollouts per GPU on "
                                   f"{R} scenes (NBP forwards batched), 5 depth frames of 256x456 per step per "
                                   "rollout, seeded synthetic NBP weights",
                       "grid": S, "rollouts_per_gpu": R, "image": [params.image_height, params.image_width]},
            "nbp_maps_per_s": round(world * stage["nbp_forward"]["maps_per_s"], 2),
            "stages": stage, "roofline": roofline, "roofline_scatter": scatter, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()                  # keep every rank alive until rank 0 has printed
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
