#!/usr/bin/env python
"""Stand-alone timings (HIP events) of the small per-step kernels on a rollout state: map accumulation at several cloud
sizes, coverage, raster, un-projection.   python tools/bench_small_kernels.py [--steps 45] [--faces simple|hard]"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextbestpath_amd.networks.nbp_model import NBP  # noqa: E402
from nextbestpath_amd.simulator import scene as sc  # noqa: E402
from nextbestpath_amd.simulator.mesh import make_maze_scene  # noqa: E402
from nextbestpath_amd.testers import nbp_planning as tp  # noqa: E402
from nextbestpath_amd.utility import hipops  # noqa: E402
from nextbestpath_amd.utility import utils as hu  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_explorer_state_dict  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ev(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--faces", default="simple")
    a = ap.parse_args()
    dev = torch.device("cuda")
    params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
    tmp = tempfile.mkdtemp()
    if a.faces == "hard":
        make_maze_scene(os.path.join(tmp, "m"), seed=100, cells=12, size=7.2, height=1.2, tess=0.15)
    else:
        make_maze_scene(os.path.join(tmp, "m"), seed=100, cells=10, size=6.0, height=1.2, tess=0.25)
    ds = sc.SceneDataset(tmp, ["m"])
    net = NBP(); net.load_state_dict(make_explorer_state_dict(9)); net = net.to(dev).eval()
    ro = tp.build_rollout(params, net, ds, (0, 0), dev, seed=8)
    for _ in range(a.steps):
        ro.step()
    torch.cuda.synchronize()
    n = int(ro.st.cloud_count.item())
    cam, mesh = ro.camera, ro.mesh
    pose, _ = cam.get_pose_from_idx(cam.cam_idx)
    print(f"state: step {a.steps}, cloud {n} points, {mesh.faces.shape[0]} faces, GT {ro.gt.shape[0]}")
    clear = ev(lambda: ro.st.maps6.zero_())
    print(f"clear of 6x256x256 alone: {clear:.1f} us")
    for npts in (n // 4, n, 3_000_000):
        cloud = ro.st.cloud[:n].repeat((npts + n - 1) // n, 1)[:npts].contiguous()
        for var in os.environ.get("SCATTER_VARIANTS", "cur").split(","):
            us = ev(lambda: hu.accumulate_step_maps(cloud, pose, ro.y_bins, 256, (-40, 40), out=ro.st.maps6))
            alg = 12 * npts + 6 * 256 * 256 * 4
            print(f"map_accumulate N={npts:8d}: {us:7.1f} us incl. clear -> {alg/us/1e3:7.1f} GB/s ({alg/us/1e3/8000:.3f} of 8 TB/s)")
    out = torch.zeros(2, dtype=torch.int32, device=dev)
    print(f"coverage (planned, 1 kernel): {ev(lambda: ro.cov_plan.count(ro.st.cloud, out, n_dev=ro.st.cloud_count, n=ro.st.cloud.shape[0], seed=1)):.1f} us")
    print(f"coverage (one-shot, sort per call): {ev(lambda: hipops.coverage_count(ro.gt, ro.st.cloud, n_dev=ro.st.cloud_count, n=ro.st.cloud.shape[0], bbox=ro.bbox, out=out)):.1f} us")
    cams4 = np.stack([f[1] for f in cam.frames[-4:]])
    H, W = params.image_height, params.image_width
    zb = torch.empty(4, H, W, device=dev)
    print(f"raster 4 frames: {ev(lambda: hipops.raster_zbuf(mesh.verts, mesh.faces, cams4, H, W, out=zb)):.1f} us")
    rgbbuf = torch.empty(4, H, W, 3, device=dev)
    print(f"raster+colours 4 frames: {ev(lambda: hipops.raster_rgbz(mesh.verts, mesh.faces, mesh.colors, cams4, H, W, out_z=zb, out_rgb=rgbbuf)):.1f} us")
    print(f"raster 1 frame : {ev(lambda: hipops.raster_zbuf(mesh.verts, mesh.faces, cams4[:1], H, W, out=zb[:1])):.1f} us")
    scratch = torch.zeros(400_000, 3, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def unproj(k):
        cnt.zero_()
        hipops.unproject_append(zb[:k], None, cams4[:k], scratch, cnt, 0.05, 70.0, seed=1)
    crgb = torch.zeros(400_000, 3, device=dev)

    def unproj_rgb(k):
        cnt.zero_()
        hipops.unproject_append(zb[:k], None, cams4[:k], scratch, cnt, 0.05, 70.0, seed=1, rgb=rgbbuf[:k], cloud_rgb=crgb)
    print(f"unproject+colours 4 frames: {ev(lambda: unproj_rgb(4)):.1f} us")
    print(f"unproject 4 frames: {ev(lambda: unproj(4)):.1f} us (incl. a 8-B clear);  1 frame: {ev(lambda: unproj(1)):.1f} us")
    x = ro.st.net_in
    print(f"NBP forward B=1: {ev(lambda: net(x), 20):.1f} us")
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        ro.step()
    t1.record(); torch.cuda.synchronize()
    print(f"full single-rollout step: {t0.elapsed_time(t1)/20*1e3:.1f} us")


if __name__ == "__main__":
    main()
