"""Thin tensor-level wrappers over the simulator / planner entry points of libnbp_hip.so
(pointers + sizes only; all arithmetic is in the HIP kernels).  Used by the rollout driver,
the reference-API mirrors and the GPU parity tests."""
from __future__ import annotations

import ctypes as C
import math

import torch

from .. import _lib

TAN_HALF_FOV = float(torch.tensor(math.tan(math.radians(30.0)), dtype=torch.float32))   # FoV 60 deg (pytorch3d default)
Z_CLIP = 0.5                                                                            # znear / 2


def _st():
    return _lib.current_stream()


_ws = {}


def _workspace(tag, nbytes, device):
    # scratch is per (kind, device, STREAM): rollouts stepped on different streams run their kernels concurrently
    key = (tag, device.index if device.index is not None else -1, _st())
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


_ws_sizes = {}


def _workspace_for(tag, size_key, size_fn, device):
    """Workspace whose size only depends on `size_key`: the C size query runs once per key."""
    n = _ws_sizes.get((tag, size_key))
    if n is None:
        n = _ws_sizes[(tag, size_key)] = int(size_fn())
    return _workspace(tag, n, device)


def cams12(R, T, device=None):
    """[n,3,3], [n,3] -> HOST numpy [n,12] fp32 (R row-major then T); cameras travel as kernel arguments."""
    import numpy as np
    R = np.asarray(R, np.float32).reshape(-1, 9)
    T = np.asarray(T, np.float32).reshape(-1, 3)
    return np.ascontiguousarray(np.concatenate([R, T], 1))


def _cam_arg(cams):
    import numpy as np
    a = np.ascontiguousarray(np.asarray(cams, np.float32).reshape(-1, 12))
    return a, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0]


def unproject_files(depth, mask):
    """Can unproject_append(..., bins=) file this call's points?  (The filing rides on the three-launch form: H W % 4 == 0, 16-byte
    aligned frames.)"""
    F_, H, W = depth.shape
    return (H * W) % 4 == 0 and depth.data_ptr() % 16 == 0 and (mask is None or mask.data_ptr() % 4 == 0)


def unproject_append(depth, mask, cams, cloud, cloud_count, gathering_factor=0.05, fov_range=70.0, seed=0,
                     tan_half_fov=TAN_HALF_FOV, rgb=None, cloud_rgb=None, shade=None, bins=None, clear=None):
    """depth [F,H,W] fp32 (F <= 8), mask [F,H,W] uint8|None, cams host [F,12]; appends to cloud [cap,3] at the
    device counter cloud_count (int64[1]).  Returns counts [F,2] int32 (device): a VIEW into this stream's scratch, valid
    until the next unproject_append on the same stream -- clone it to keep it.
    bins (utils.CloudBins of `cloud`): the launch also files the appended points into the tile-binned store; clear = (maps6 [6,S,S],
    traj [S,S]): ... and zeroes what the map build behind it accumulates into (utils.step_maps(..., prefiled=True): one launch)."""
    F_, H, W = depth.shape
    if F_ > 8:
        raise ValueError("unproject_append: at most 8 frames per call")
    L = _lib.lib()
    keep, cam_ptr, _ = _cam_arg(cams)
    ws = _workspace_for("unproject", (F_, H, W), lambda: L.nbp_unproject_workspace_bytes(F_, H, W) + 256, depth.device)
    counts = ws[-64:].view(torch.int32).reshape(8, 2)[:F_]          # per-stream scratch: no allocation per call
    if bins is not None:
        zface = verts = faces = vcolors = None
        ambient = 0.85
        if shade is not None:
            zface, verts, faces, vcolors, ambient = shade
            rgb = None
        z6, z1, S = (None, None, 0) if clear is None else (clear[0], clear[1], int(clear[0].shape[-1]))
        rc = L.nbp_unproject_append_filed_f32(_lib.ptr(depth), _lib.ptr(mask), _lib.ptr(rgb), _lib.ptr(zface), _lib.ptr(verts),
                                              _lib.ptr(faces), _lib.ptr(vcolors), cam_ptr, F_, H, W, tan_half_fov, float(fov_range),
                                              float(gathering_factor), int(seed) & 0xFFFFFFFF, float(ambient), _lib.ptr(counts),
                                              _lib.ptr(cloud), _lib.ptr(cloud_rgb), _lib.ptr(cloud_count), cloud.shape[0],
                                              bins.store.data_ptr(), _lib.ptr(z6), _lib.ptr(z1), S, _lib.ptr(ws), ws.numel(), _st())
        _lib.check(rc, "nbp_unproject_append_filed_f32")
        return counts
    if shade is not None:      # deferred shading: (zface [F,H,W] int64, verts, faces, vcolors, ambient) -> colours of kept pixels
        zface, verts, faces, vcolors, ambient = shade
        rc = L.nbp_unproject_append_shaded_f32(_lib.ptr(depth), _lib.ptr(mask), _lib.ptr(zface), _lib.ptr(verts), _lib.ptr(faces),
                                               _lib.ptr(vcolors), cam_ptr, F_, H, W, tan_half_fov, float(fov_range),
                                               float(gathering_factor), int(seed) & 0xFFFFFFFF, float(ambient),
                                               _lib.ptr(counts), _lib.ptr(cloud), _lib.ptr(cloud_rgb), _lib.ptr(cloud_count),
                                               cloud.shape[0], _lib.ptr(ws), ws.numel(), _st())
        _lib.check(rc, "nbp_unproject_append_shaded_f32")
        return counts
    if rgb is not None:        # colours of the kept pixels ride along (compute_partial_point_cloud with images)
        rc = L.nbp_unproject_append_rgb_f32(_lib.ptr(depth), _lib.ptr(mask), _lib.ptr(rgb), cam_ptr, F_, H, W, tan_half_fov,
                                            float(fov_range), float(gathering_factor), int(seed) & 0xFFFFFFFF,
                                            _lib.ptr(counts), _lib.ptr(cloud), _lib.ptr(cloud_rgb), _lib.ptr(cloud_count),
                                            cloud.shape[0], _lib.ptr(ws), ws.numel(), _st())
        _lib.check(rc, "nbp_unproject_append_rgb_f32")
        return counts
    rc = L.nbp_unproject_append_f32(_lib.ptr(depth), _lib.ptr(mask), cam_ptr, F_, H, W, tan_half_fov,
                                    float(fov_range), float(gathering_factor), int(seed) & 0xFFFFFFFF,
                                    _lib.ptr(counts), _lib.ptr(cloud), _lib.ptr(cloud_count), cloud.shape[0],
                                    _lib.ptr(ws), ws.numel(), _st())
    _lib.check(rc, "nbp_unproject_append_f32")
    return counts


def raster_zbuf(verts, faces, cams, H, W, bin_cap=2048, tan_half_fov=TAN_HALF_FOV, z_clip=Z_CLIP, out=None,
                overflow=None):
    """verts [V,3] fp32, faces [F,3] int32, cams host [n,12] -> zbuf [n,H,W] (-1 background)."""
    L = _lib.lib()
    keep, cam_ptr, n = _cam_arg(cams)
    if out is None:
        out = torch.empty(n, H, W, dtype=torch.float32, device=verts.device)
    if overflow is None:
        overflow = torch.zeros(1, dtype=torch.int32, device=verts.device)
    nf = faces.shape[0]
    ws = _workspace_for("raster", (nf, n, H, W), lambda: L.nbp_raster_workspace_bytes(nf, n, H, W, bin_cap), verts.device)
    rc = L.nbp_raster_zbuf_f32(_lib.ptr(verts), verts.shape[0], _lib.ptr(faces), faces.shape[0], cam_ptr, n, H,
                               W, tan_half_fov, z_clip, bin_cap, _lib.ptr(out), _lib.ptr(overflow), _lib.ptr(ws),
                               ws.numel(), _st())
    _lib.check(rc, "nbp_raster_zbuf_f32")
    return out, overflow


def raster_rgbz(verts, faces, vcolors, cams, H, W, ambient=0.85, contrast=1.0, tan_half_fov=TAN_HALF_FOV, z_clip=Z_CLIP,
                out_z=None, out_rgb=None):
    """Camera.capture_image with colours -> (zbuf [n,H,W], rgb [n,H,W,3])."""
    L = _lib.lib()
    keep, cam_ptr, n = _cam_arg(cams)
    if out_z is None:
        out_z = torch.empty(n, H, W, dtype=torch.float32, device=verts.device)
    if out_rgb is None:
        out_rgb = torch.empty(n, H, W, 3, dtype=torch.float32, device=verts.device)
    nf = faces.shape[0]
    ws = _workspace_for("raster_rgb", (nf, n, H, W), lambda: L.nbp_raster_rgb_workspace_bytes(nf, n, H, W), verts.device)
    rc = L.nbp_raster_rgbz_f32(_lib.ptr(verts), verts.shape[0], _lib.ptr(faces), nf, _lib.ptr(vcolors), cam_ptr, n, H, W,
                               tan_half_fov, z_clip, float(ambient), float(contrast), _lib.ptr(out_z), _lib.ptr(out_rgb),
                               _lib.ptr(ws), ws.numel(), _st())
    _lib.check(rc, "nbp_raster_rgbz_f32")
    return out_z, out_rgb


def raster_zface(verts, faces, cams, H, W, tan_half_fov=TAN_HALF_FOV, z_clip=Z_CLIP, out_z=None, out_zface=None):
    """Depth render that also returns zface [n,H,W] int64 = (depth bits << 32 | nearest face), -1 for background: the
    input of the deferred colour evaluation (shade_image / unproject_append(shade=...))."""
    L = _lib.lib()
    keep, cam_ptr, n = _cam_arg(cams)
    if out_z is None:
        out_z = torch.empty(n, H, W, dtype=torch.float32, device=verts.device)
    if out_zface is None:
        out_zface = torch.empty(n, H, W, dtype=torch.int64, device=verts.device)
    nf = faces.shape[0]
    ws = _workspace_for("raster", (nf, n, H, W), lambda: L.nbp_raster_workspace_bytes(nf, n, H, W, 0), verts.device)
    rc = L.nbp_raster_zface_f32(_lib.ptr(verts), verts.shape[0], _lib.ptr(faces), nf, cam_ptr, n, H, W, tan_half_fov, z_clip,
                                _lib.ptr(out_z), _lib.ptr(out_zface), _lib.ptr(ws), ws.numel(), _st())
    _lib.check(rc, "nbp_raster_zface_f32")
    return out_z, out_zface


def shade_image(zface, verts, faces, vcolors, cams, ambient=0.85, tan_half_fov=TAN_HALF_FOV, out=None):
    """rgb [n,H,W,3] of frames rendered by raster_zface (bit-identical to raster_rgbz with contrast 1)."""
    L = _lib.lib()
    keep, cam_ptr, n = _cam_arg(cams)
    _, H, W = zface.shape
    if out is None:
        out = torch.empty(n, H, W, 3, dtype=torch.float32, device=verts.device)
    rc = L.nbp_shade_image_f32(_lib.ptr(zface), _lib.ptr(verts), _lib.ptr(faces), _lib.ptr(vcolors), cam_ptr, n, H, W,
                               tan_half_fov, float(ambient), 1.0, _lib.ptr(out), _st())
    _lib.check(rc, "nbp_shade_image_f32")
    return out


def segments_hit_mesh(verts, faces, segs):
    """segs [E,6] fp32 device -> int32 [E] (1 = the segment hits the mesh)."""
    hit = torch.empty(segs.shape[0], dtype=torch.int32, device=verts.device)
    rc = _lib.lib().nbp_segments_hit_mesh_f32(_lib.ptr(verts), _lib.ptr(faces), faces.shape[0], _lib.ptr(segs),
                                              segs.shape[0], _lib.ptr(hit), _st())
    _lib.check(rc, "nbp_segments_hit_mesh_f32")
    return hit


def axis_ray_counts(verts, faces, pts):
    cnt = torch.empty(pts.shape[0], 3, dtype=torch.int32, device=verts.device)
    rc = _lib.lib().nbp_axis_ray_counts_f32(_lib.ptr(verts), _lib.ptr(faces), faces.shape[0], _lib.ptr(pts),
                                            pts.shape[0], _lib.ptr(cnt), _st())
    _lib.check(rc, "nbp_axis_ray_counts_f32")
    return cnt


def slice_obstacle(verts, faces, y0, cx, cz, S=256, grid_range=(-40, 40), half_width_px=1.04, out=None):
    """GT obstacle label [S,S] fp32 {0,1} around (cx, cz) at height y0 (ref utils.py:226-262)."""
    if out is None:
        out = torch.empty(S, S, dtype=torch.float32, device=verts.device)
    rc = _lib.lib().nbp_slice_obstacle_f32(_lib.ptr(verts), _lib.ptr(faces), faces.shape[0], float(y0), float(cx), float(cz),
                                           S, float(grid_range[0]), float(grid_range[1]), float(half_width_px),
                                           _lib.ptr(out), _st())
    _lib.check(rc, "nbp_slice_obstacle_f32")
    return out


def reference_figure_geometry(S=256, view_size=80.0):
    """(half_u, scale_u, half_v, scale_v, half_width_px, cap_px) of the reference's label image (utils.py:232-258): a 2.56 in figure
    at 100 dpi whose default axes box is 0.775 x 0.77 of it (198.4 x 197.12 px), x limits view_size wide, y limits shrunk to the box's
    aspect, the box saved as 198 x 197 px ('tight' bbox, pad 0), resized to S x S, flipped left-right; lines 1.5 pt wide with
    projecting caps.  Pinned by tests/golden/obstacle_label.npz."""
    fig = 2.56 * 100.0
    ax_w, ax_h = 0.775 * fig, 0.77 * fig                   # matplotlib's default subplot box: left .125 right .9 bottom .11 top .88
    png_w, png_h = int(ax_w), int(ax_h)                    # the saved crop
    px_per_unit = ax_w / view_size                         # equal aspect: the same on both axes before the resize
    scale_u, scale_v = px_per_unit * S / png_w, px_per_unit * S / png_h
    centre_u = S - (ax_w / 2.0) * S / png_w                # after the left-right flip
    centre_v = (png_h - ax_h / 2.0) * S / png_h            # rows count from the top of the 197-px crop
    half_width = 0.5 * (1.5 * 100.0 / 72.0) * 0.5 * (S / png_w + S / png_h)
    return centre_u / scale_u, scale_u, centre_v / scale_v, scale_v, half_width, half_width


def slice_obstacle_fig(verts, faces, y0, cx, cz, S=256, view_size=80.0, out=None):
    """The GT obstacle label on the reference's own pixel grid (nbp_slice_obstacle_fig_f32)."""
    if out is None:
        out = torch.empty(S, S, dtype=torch.float32, device=verts.device)
    hu, su, hv, sv, hw, cap = reference_figure_geometry(S, view_size)
    rc = _lib.lib().nbp_slice_obstacle_fig_f32(_lib.ptr(verts), _lib.ptr(faces), faces.shape[0], float(y0), float(cx), float(cz),
                                               S, hu, su, hv, sv, hw, cap, _lib.ptr(out), _st())
    _lib.check(rc, "nbp_slice_obstacle_fig_f32")
    return out


def fuse_obstacle(out2, maps6, traj, threshold=0.13):
    S = maps6.shape[-1]
    obst = torch.empty(S, S, dtype=torch.float32, device=maps6.device)
    fullproj = torch.empty(S, S, dtype=torch.float32, device=maps6.device)
    rc = _lib.lib().nbp_fuse_obstacle_f32(_lib.ptr(out2), _lib.ptr(maps6), _lib.ptr(traj), float(threshold), S,
                                          _lib.ptr(obst), _lib.ptr(fullproj), _st())
    _lib.check(rc, "nbp_fuse_obstacle_f32")
    return obst, fullproj


def score_candidates(pos, pose_xyz, out1, fullproj, skip=None, grid_range=(-40, 40)):
    """pos [P,3] device; out1 [8,V,V] device; -> (valid u8 [P], cell int32 [P,2], score f64 [P]) on device."""
    P, V, S = pos.shape[0], out1.shape[-1], fullproj.shape[-1]
    dev = pos.device
    valid = torch.empty(P, dtype=torch.uint8, device=dev)
    cell = torch.empty(P, 2, dtype=torch.int32, device=dev)
    score = torch.empty(P, dtype=torch.float64, device=dev)
    rc = _lib.lib().nbp_score_candidates_f32(_lib.ptr(pos), P, float(pose_xyz[0]), float(pose_xyz[2]), _lib.ptr(out1),
                                             V, _lib.ptr(fullproj), S, float(grid_range[0]), float(grid_range[1]),
                                             _lib.ptr(skip), _lib.ptr(valid), _lib.ptr(cell), _lib.ptr(score), _st())
    _lib.check(rc, "nbp_score_candidates_f32")
    return valid, cell, score


def edges_blocked(obst, pose_xyz, pos, edges, grid_range=(-40, 40)):
    E = edges.shape[0]
    out = torch.empty(E, dtype=torch.uint8, device=pos.device)
    rc = _lib.lib().nbp_edges_blocked_u8(_lib.ptr(obst), obst.shape[-1], float(grid_range[0]), float(grid_range[1]),
                                         float(pose_xyz[0]), float(pose_xyz[2]), _lib.ptr(pos), _lib.ptr(edges), E,
                                         _lib.ptr(out), _st())
    _lib.check(rc, "nbp_edges_blocked_u8")
    return out


def coverage_count(gt, pc, n_dev=None, n=None, weight=2, seed=0, threshold=1.0, bbox=None, out=None):
    """-> (count int32[1], m int32[1]) device tensors; coverage = count / len(gt)."""
    L = _lib.lib()
    G = gt.shape[0]
    k = int(G * weight)
    if bbox is None:
        lo, hi = gt.min(0).values.tolist(), gt.max(0).values.tolist()
    else:
        lo, hi = bbox
    lo_a, hi_a = (C.c_float * 3)(*lo), (C.c_float * 3)(*hi)
    ws = _workspace("coverage", L.nbp_coverage_workspace_bytes(lo_a, hi_a, float(threshold), k), gt.device)
    if out is None:
        out = torch.empty(2, dtype=torch.int32, device=gt.device)
    N = pc.shape[0] if n is None else int(n)
    rc = L.nbp_coverage_count_f32(_lib.ptr(gt), G, _lib.ptr(pc), N, _lib.ptr(n_dev), k, int(seed) & 0xFFFFFFFF,
                                  float(threshold), lo_a, hi_a, out[0:1].data_ptr(), out[1:2].data_ptr(), _lib.ptr(ws),
                                  ws.numel(), _st())
    _lib.check(rc, "nbp_coverage_count_f32")
    return out


class CoveragePlan:
    """GT cloud sorted once into the coverage grid (one per rollout); `count` then costs one kernel per call."""

    def __init__(self, gt, threshold=1.0, weight=2, bbox=None):
        L = _lib.lib()
        self.gt, self.G, self.thr, self.k = gt, gt.shape[0], float(threshold), int(gt.shape[0] * weight)
        lo, hi = (gt.min(0).values.tolist(), gt.max(0).values.tolist()) if bbox is None else bbox
        self.lo, self.hi = (C.c_float * 3)(*lo), (C.c_float * 3)(*hi)
        self.plan = torch.empty(L.nbp_coverage_plan_bytes(self.lo, self.hi, self.thr, self.G), dtype=torch.uint8,
                                device=gt.device)
        ws = torch.empty(L.nbp_coverage_plan_workspace_bytes(self.lo, self.hi, self.thr, self.G), dtype=torch.uint8,
                         device=gt.device)
        rc = L.nbp_coverage_plan_build_f32(_lib.ptr(gt), self.G, self.thr, self.lo, self.hi, _lib.ptr(self.plan),
                                           self.plan.numel(), _lib.ptr(ws), ws.numel(), _st())
        _lib.check(rc, "nbp_coverage_plan_build_f32")
        self.epoch = 0
        self._m = torch.zeros(1, dtype=torch.int32, device=gt.device)

    def count(self, pc, out, n_dev=None, n=None, seed=0, out_is_zero=False):
        """Adds the covered-GT count to out[0] (int32 device; zeroed first unless out_is_zero), out[1] = sample size."""
        if not out_is_zero:
            out[0:1].zero_()
        self.epoch += 1
        N = pc.shape[0] if n is None else int(n)
        rc = _lib.lib().nbp_coverage_count_planned_f32(_lib.ptr(self.plan), self.G, self.thr, self.lo, self.hi, _lib.ptr(pc), N,
                                                       _lib.ptr(n_dev), self.k, int(seed) & 0xFFFFFFFF, self.epoch,
                                                       out[0:1].data_ptr(), out[1:2].data_ptr(), _st())
        _lib.check(rc, "nbp_coverage_count_planned_f32")
        return out


def replan_batch(items, S, V, grid_range=(-40, 40), threshold=0.13):
    """fuse_obstacle + score_candidates + edges_blocked for several rollouts in two launches; items = one tuple per
    rollout from LatticePlanner.replan_item.  More than 16 items go in chunks of 16."""
    import numpy as np
    if len(items) > 16:
        for i in range(0, len(items), 16):
            replan_batch(items[i:i + 16], S, V, grid_range, threshold)
        return
    n = len(items)
    VP, I = C.c_void_p, C.c_int
    a = [(VP * n)() for _ in range(13)]
    P, E = (I * n)(), (I * n)()
    xz = np.zeros((n, 2), np.float32)
    for i, (out2, maps6, traj, obst, fullproj, pos, cxz, out1, skip, valid, cell, score, edges, blocked) in enumerate(items):
        for arr, t in zip(a, (out2, maps6, traj, obst, fullproj, pos, out1, skip, valid, cell, score, edges, blocked)):
            arr[i] = None if t is None else t.data_ptr()
        P[i], E[i] = pos.shape[0], edges.shape[0]
        xz[i] = cxz
    o2, m6, tr, ob, fp, ps, o1, sk, va, ce, sc, ed, bl = a
    rc = _lib.lib().nbp_replan_batch_f32(n, o2, m6, tr, float(threshold), int(S), ob, fp, ps, P, xz.ctypes.data, o1, int(V),
                                         float(grid_range[0]), float(grid_range[1]), sk, va, ce, sc, ed, E, bl, _st())
    _lib.check(rc, "nbp_replan_batch_f32")


def coverage_count_batch(items):
    """CoveragePlan.count for several rollouts in two launches: items = [(plan, pc, out[2] int32, n_dev, n, seed, out_is_zero)]
    (more than 16 go in chunks of 16); the plans share the threshold."""
    import numpy as np
    if len(items) > 16:
        for i in range(0, len(items), 16):
            coverage_count_batch(items[i:i + 16])
        return
    n = len(items)
    VP, LL, U, I = C.c_void_p, C.c_longlong, C.c_uint, C.c_int
    plans, G, pc, N, ndev, k, seed, epoch, cnt, mout = (VP * n)(), (I * n)(), (VP * n)(), (LL * n)(), (VP * n)(), (LL * n)(), (U * n)(), \
        (U * n)(), (VP * n)(), (VP * n)()
    lo, hi = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    for i, (plan, cloud, out, n_dev, nn, sd, out_is_zero) in enumerate(items):
        if not out_is_zero:
            out[0:1].zero_()
        plan.epoch += 1
        plans[i], G[i], pc[i], N[i] = plan.plan.data_ptr(), plan.G, cloud.data_ptr(), cloud.shape[0] if nn is None else int(nn)
        ndev[i], k[i], seed[i], epoch[i] = _lib.ptr(n_dev), plan.k, int(sd) & 0xFFFFFFFF, plan.epoch
        cnt[i], mout[i] = out[0:1].data_ptr(), out[1:2].data_ptr()
        lo[i], hi[i] = list(plan.lo), list(plan.hi)
    rc = _lib.lib().nbp_coverage_count_planned_batch_f32(n, plans, G, items[0][0].thr, lo.ctypes.data, hi.ctypes.data, pc, N, ndev, k, seed,
                                                         epoch, cnt, mout, _st())
    _lib.check(rc, "nbp_coverage_count_planned_batch_f32")


def _item_ws(tag, owner, key, nbytes, device):
    """Per-item scratch of the batched stages (items run concurrently: no sharing between rollouts).  The scratch belongs to
    `owner` (the rollout: any object with a __dict__) and is released with it -- a module-level cache keyed by id(rollout) kept
    ~1.2 KB per face per rollout alive for the life of the process."""
    store = owner.__dict__.setdefault("_hip_scratch", {})
    w = store.get((tag, key))
    if w is None or w.numel() < nbytes:
        w = store[(tag, key)] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return w


def unproject_append_batch(items, H, W, n_frames, gathering_factor=0.05, fov_range=70.0, tan_half_fov=TAN_HALF_FOV):
    """unproject_append for several rollouts in three launches.  items = [(key, depth frames [F tensors [H,W]], cams host
    [F,12], cloud, cloud_count, seed, cloud_rgb | None, shade | None)] with shade = (zface frames [F tensors [H,W] int64], verts,
    faces, vcolors, ambient); the frames need not be adjacent in memory; `key` is the rollout OBJECT: it owns its scratch.
    More than 12 items go in chunks of 12."""
    import numpy as np
    if len(items) > 12:
        for i in range(0, len(items), 12):
            unproject_append_batch(items[i:i + 12], H, W, n_frames, gathering_factor, fov_range, tan_half_fov)
        return
    n = len(items)
    L = _lib.lib()
    VP, LL, U = C.c_void_p, C.c_longlong, C.c_uint
    wsb = int(L.nbp_unproject_workspace_bytes(n_frames, H, W)) + 512
    dep, zf = (VP * (n * n_frames))(), (VP * (n * n_frames))()
    ve, fa, vc, cnts, cl, crgb, ccount, cap, wsp, seeds = (VP * n)(), (VP * n)(), (VP * n)(), (VP * n)(), (VP * n)(), (VP * n)(), (VP * n)(), \
        (LL * n)(), (VP * n)(), (U * n)()
    cams = np.zeros((n, n_frames, 12), np.float32)
    ambient = 0.85
    for i, (key, depth, cam, cloud, cloud_count, seed, cloud_rgb, shade) in enumerate(items):
        ws = _item_ws("unproject", key, (n_frames, H, W), wsb, cloud.device)
        cl[i], ccount[i], cap[i], wsp[i], seeds[i] = cloud.data_ptr(), cloud_count.data_ptr(), cloud.shape[0], ws.data_ptr(), \
            int(seed) & 0xFFFFFFFF
        cnts[i] = ws.data_ptr() + wsb - 256               # 2 n_frames ints in the tail of the item's scratch
        cams[i] = np.asarray(cam, np.float32).reshape(n_frames, 12)
        for f in range(n_frames):
            dep[i * n_frames + f] = depth[f] if isinstance(depth[f], int) else depth[f].data_ptr()
        if shade is not None and cloud_rgb is not None:
            zface, verts, faces, vcolors, ambient = shade
            for f in range(n_frames):
                zf[i * n_frames + f] = zface[f] if isinstance(zface[f], int) else zface[f].data_ptr()
            ve[i], fa[i], vc[i], crgb[i] = verts.data_ptr(), faces.data_ptr(), vcolors.data_ptr(), cloud_rgb.data_ptr()
    rc = L.nbp_unproject_append_shaded_batch_f32(n, dep, zf, ve, fa, vc, cams.ctypes.data, n_frames, H, W, tan_half_fov, float(fov_range),
                                                 float(gathering_factor), seeds, float(ambient), cnts, cl, crgb, ccount, cap, wsp,
                                                 wsb - 256, _st())
    _lib.check(rc, "nbp_unproject_append_shaded_batch_f32")


def raster_zface_batch(items, H, W, n_frames, tan_half_fov=TAN_HALF_FOV, z_clip=Z_CLIP):
    """raster_zface for several rollouts, each its own mesh, in four launches.  items = [(key = the rollout object, which owns the
    scratch, verts, faces, cams host [F,12], out_z [F,H,W], out_zface [F,H,W] int64)].  More than 12 items go in chunks of 12."""
    import numpy as np
    if len(items) > 12:
        for i in range(0, len(items), 12):
            raster_zface_batch(items[i:i + 12], H, W, n_frames, tan_half_fov, z_clip)
        return
    n = len(items)
    L = _lib.lib()
    VP, I, SZ = C.c_void_p, C.c_int, C.c_size_t
    ve, nv, fa, nf, zb, zf, wsp, wsn = (VP * n)(), (I * n)(), (VP * n)(), (I * n)(), (VP * n)(), (VP * n)(), (VP * n)(), (SZ * n)()
    cams = np.zeros((n, n_frames, 12), np.float32)
    for i, (key, verts, faces, cam, out_z, out_zface) in enumerate(items):
        F_ = faces.shape[0]
        nb = _ws_sizes.get(("raster", (F_, n_frames, H, W)))
        if nb is None:
            nb = _ws_sizes[("raster", (F_, n_frames, H, W))] = int(L.nbp_raster_workspace_bytes(F_, n_frames, H, W, 0))
        ws = _item_ws("raster", key, (F_, n_frames, H, W), nb, verts.device)
        ve[i], nv[i], fa[i], nf[i], zb[i], zf[i], wsp[i], wsn[i] = verts.data_ptr(), verts.shape[0], faces.data_ptr(), F_, out_z.data_ptr(), \
            out_zface.data_ptr(), ws.data_ptr(), ws.numel()
        cams[i] = np.asarray(cam, np.float32).reshape(n_frames, 12)
    rc = L.nbp_raster_zface_batch_f32(n, ve, nv, fa, nf, cams.ctypes.data, n_frames, H, W, tan_half_fov, z_clip, zb, zf, wsp, wsn, _st())
    _lib.check(rc, "nbp_raster_zface_batch_f32")


def carve_update(proxy_pts, depth, mask, cam12_host, zfar, fov_range, tol, score_threshold, n_inside, n_behind, occ,
                 out_of_field, tan_half_fov=TAN_HALF_FOV):
    """A20 (macarons_utils.py:2849-2949, 3329-3363): in-place update of the per-proxy-point carving state."""
    H, W = depth.shape[-2:]
    cam = (C.c_float * 12)(*[float(x) for x in cam12_host])
    rc = _lib.lib().nbp_carve_update_f32(_lib.ptr(proxy_pts), proxy_pts.shape[0], _lib.ptr(depth), _lib.ptr(mask), cam, H, W,
                                         tan_half_fov, float(zfar), float(fov_range), float(tol), float(score_threshold),
                                         _lib.ptr(n_inside), _lib.ptr(n_behind), _lib.ptr(occ), _lib.ptr(out_of_field),
                                         _st())
    _lib.check(rc, "nbp_carve_update_f32")


def camera_center(cam12_host):
    """World position of the camera of a [12] host camera (R row-major, T): C = -T R^T in fp32 (get_camera_center)."""
    import numpy as np
    c = np.asarray(cam12_host, np.float32)
    R, T = c[:9].reshape(3, 3), c[9:]
    return np.array([-((T[0] * R[j, 0] + T[1] * R[j, 1]) + T[2] * R[j, 2]) for j in range(3)], np.float32)


def view_state_update(pts, x_view_host, n_elev, n_azim, view_states, mask=None, sd=None, distance_to_surface=0.0):
    """compute_view_state (scone_utils.py:799-862) OR-ed into view_states [P, n_elev * n_azim] (fp32 0 / 1) for the selected
    points: mask [P] uint8 (None = all) and sd [P] < distance_to_surface (None = no test); x_view_host [n_view <= 8, 3]."""
    import numpy as np
    xv = np.ascontiguousarray(x_view_host, np.float32).reshape(-1, 3)
    rc = _lib.lib().nbp_view_state_update_f32(_lib.ptr(pts), pts.shape[0], _lib.ptr(mask), _lib.ptr(sd), float(distance_to_surface),
                                              xv.ctypes.data_as(C.POINTER(C.c_float)), len(xv), int(n_elev), int(n_azim),
                                              _lib.ptr(view_states), _st())
    _lib.check(rc, "nbp_view_state_update_f32")


def view_gain(proxy_pts, occ, view_states, cams12_host, x_cams_host, n_elev, n_azim, H, W, fov_range, tan_half_fov=TAN_HALF_FOV):
    """gains [n_cams] int32 (device): occupied proxy points in each candidate's field of view not yet seen from its direction."""
    import numpy as np
    cams = np.ascontiguousarray(cams12_host, np.float32).reshape(-1, 12)
    xs = np.ascontiguousarray(x_cams_host, np.float32).reshape(-1, 3)
    gains = torch.empty(len(cams), dtype=torch.int32, device=proxy_pts.device)
    fp = C.POINTER(C.c_float)
    rc = _lib.lib().nbp_view_gain_i32(_lib.ptr(proxy_pts), proxy_pts.shape[0], _lib.ptr(occ), _lib.ptr(view_states),
                                      cams.ctypes.data_as(fp), xs.ctypes.data_as(fp), len(cams), int(n_elev), int(n_azim), int(H), int(W),
                                      tan_half_fov, float(fov_range), _lib.ptr(gains), _st())
    _lib.check(rc, "nbp_view_gain_i32")
    return gains


def carve_view_update(proxy_pts, depth, mask, cam12_host, zfar, fov_range, tol, score_threshold, n_inside, n_behind, occ,
                      out_of_field, x_cam_host, n_elev, n_azim, distance_to_surface, view_states, fov_mask=None, sd=None,
                      tan_half_fov=TAN_HALF_FOV):
    """carve_update + Scene.update_proxy_view_states (macarons_utils.py:3268-3327) for the frame's camera in ONE launch."""
    H, W = depth.shape[-2:]
    cam = (C.c_float * 12)(*[float(x) for x in cam12_host])
    xc = (C.c_float * 3)(*[float(x) for x in x_cam_host])
    rc = _lib.lib().nbp_carve_view_update_f32(_lib.ptr(proxy_pts), proxy_pts.shape[0], _lib.ptr(depth), _lib.ptr(mask), cam, H, W,
                                              tan_half_fov, float(zfar), float(fov_range), float(tol), float(score_threshold),
                                              _lib.ptr(n_inside), _lib.ptr(n_behind), _lib.ptr(occ), _lib.ptr(out_of_field), xc,
                                              int(n_elev), int(n_azim), float(distance_to_surface), _lib.ptr(view_states),
                                              _lib.ptr(fov_mask), _lib.ptr(sd), _st())
    _lib.check(rc, "nbp_carve_view_update_f32")


def _scene_geom(scene):
    import numpy as np
    box = np.ascontiguousarray(scene.box6(), np.float32)
    grid = np.ascontiguousarray(scene.grid3(), np.int32)
    return box, grid, box.ctypes.data_as(C.POINTER(C.c_float)), grid.ctypes.data_as(C.POINTER(C.c_int))


def scene_fill_cells(scene, pts, n_point_min=0, seed=0, n_dev=None):
    """Scene.fill_cells on the device store of `scene` (simulator/scene.py::Scene); pts [n,3] fp32 device."""
    L = _lib.lib()
    box, grid, bp, gp = _scene_geom(scene)
    n = pts.shape[0]
    ws = _workspace("scene_fill", L.nbp_scene_fill_workspace_bytes(bp, gp, scene.cell_capacity, n, scene.cell_resolution),
                    pts.device)
    rc = L.nbp_scene_fill_cells_f32(_lib.ptr(pts), n, _lib.ptr(n_dev), bp, gp, scene.cell_capacity,
                                    float(scene.cell_resolution), int(n_point_min), int(seed) & 0xFFFFFFFF,
                                    _lib.ptr(scene.cell_pts), _lib.ptr(scene.cell_count), _lib.ptr(ws), ws.numel(), _st())
    _lib.check(rc, "nbp_scene_fill_cells_f32")


def scene_gather(scene):
    """Scene.return_entire_pt_cloud -> [G,3] device tensor (one sync to size the result)."""
    cap_total = scene.n_cells * scene.cell_capacity
    out = torch.empty(cap_total, 3, dtype=torch.float32, device=scene.cell_pts.device)
    n_out = torch.zeros(1, dtype=torch.int64, device=out.device)
    rc = _lib.lib().nbp_scene_gather_f32(_lib.ptr(scene.cell_pts), _lib.ptr(scene.cell_count), scene.n_cells,
                                         scene.cell_capacity, _lib.ptr(out), cap_total, _lib.ptr(n_out), _st())
    _lib.check(rc, "nbp_scene_gather_f32")
    return out[:int(n_out.item())].clone()


def scene_coverage(gt_scene, rec_scene, epsilon, out=None):
    """-> int32 [2] device: (covered GT points, stored GT points)."""
    L = _lib.lib()
    box, grid, bp, gp = _scene_geom(gt_scene)
    if out is None:
        out = torch.empty(2, dtype=torch.int32, device=gt_scene.cell_pts.device)
    ws = _workspace("scene_cov", L.nbp_scene_coverage_workspace_bytes(bp, gp, rec_scene.cell_capacity, float(epsilon)),
                    out.device)
    rc = L.nbp_scene_coverage_f32(_lib.ptr(gt_scene.cell_pts), _lib.ptr(gt_scene.cell_count), gt_scene.cell_capacity,
                                  _lib.ptr(rec_scene.cell_pts), _lib.ptr(rec_scene.cell_count), rec_scene.cell_capacity,
                                  bp, gp, float(epsilon), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _st())
    _lib.check(rc, "nbp_scene_coverage_f32")
    return out


def points_in_fov(pts, cams, H, W, fov_range, want_mask=True, tan_half_fov=TAN_HALF_FOV):
    """Camera.get_points_in_fov for n cameras (host [n,12]) -> (mask uint8 [n,P] or None, any int32 [n])."""
    keep, cam_ptr, n = _cam_arg(cams)
    P = pts.shape[0]
    mask = torch.empty(n, P, dtype=torch.uint8, device=pts.device) if want_mask else None
    any_ = torch.empty(n, dtype=torch.int32, device=pts.device)
    rc = _lib.lib().nbp_points_in_fov_u8(_lib.ptr(pts), P, cam_ptr, n, H, W, tan_half_fov, float(fov_range), _lib.ptr(mask),
                                         _lib.ptr(any_), _st())
    _lib.check(rc, "nbp_points_in_fov_u8")
    return mask, any_


def sample_points(pc, k, seed=0, n_dev=None, n=None):
    """The first k points of a seeded random permutation of pc[:n] -> (out [k,3] device, m int64[1] device)."""
    N = pc.shape[0] if n is None else int(n)
    out = torch.empty(min(int(k), N), 3, dtype=torch.float32, device=pc.device)
    m = torch.zeros(1, dtype=torch.int64, device=pc.device)
    rc = _lib.lib().nbp_sample_points_f32(_lib.ptr(pc), N, _lib.ptr(n_dev), int(k), int(seed) & 0xFFFFFFFF, _lib.ptr(out),
                                          _lib.ptr(m), _st())
    _lib.check(rc, "nbp_sample_points_f32")
    return out, m


def append_points(dst, offset, pts_host):
    """dst[offset:offset+n] = pts (n <= 8 host points, passed in the kernel arguments)."""
    import numpy as np
    a = np.ascontiguousarray(np.asarray(pts_host, np.float32).reshape(-1, 3))
    rc = _lib.lib().nbp_append_points_f32(_lib.ptr(dst), int(offset), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], _st())
    _lib.check(rc, "nbp_append_points_f32")
