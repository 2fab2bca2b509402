"""Map accumulation -- drop-in for next_best_path/utility/utils.py:160-223.

Same function names, positional arguments and return shapes as the reference; the work is
done by the HIP kernels of csrc/nbp_maps.hip (no torch arithmetic, no CPU fallback).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib


def _need_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a cuda tensor (no CPU fallback)")


def _pose_xyz(camera_pose):
    # the reference reads the pose through .tolist() (one device sync); accept host floats too
    if isinstance(camera_pose, torch.Tensor):
        vals = camera_pose.detach().flatten()[:3].tolist()
    else:
        vals = list(camera_pose)[:3]
    return float(vals[0]), float(vals[1]), float(vals[2])


def get_point_position_in_the_img(points_2d, grid_size, grid_range):
    """ref utils.py:160-164 -- [...,2] fp32 -> stack((i0, i1)).squeeze() int64, no bounds check."""
    _need_cuda(points_2d, "get_point_position_in_the_img")
    p = points_2d.reshape(-1, 2).contiguous().float()
    K = p.shape[0]
    out = torch.empty(2, K, dtype=torch.int64, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.lib().nbp_point_position_i64(p.data_ptr(), K, int(grid_size[0]), int(grid_size[1]),
                                               float(grid_range[0]), float(grid_range[1]), out.data_ptr(),
                                               _lib.current_stream())
    _lib.check(rc, "nbp_point_position_i64")
    return out.reshape((2,) + tuple(points_2d.shape[:-1])).squeeze()


def transform_points_to_n_pieces(points, camera_pose, device=None, no_rotation=True):
    """ref utils.py:166-196 -- world [N,3] -> agent-centred [1,N,2] = (-(z-cz), -(x-cx))."""
    if not no_rotation:
        raise NotImplementedError("the reference only ever calls this with no_rotation=True")
    _need_cuda(points, "transform_points_to_n_pieces")
    cx, cy, cz = _pose_xyz(camera_pose)
    p = points.contiguous().float()
    N = p.shape[0]
    out = torch.empty(1, N, 2, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.lib().nbp_transform_points_f32(p.data_ptr(), N, cx, cy, cz, out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "nbp_transform_points_f32")
    return out


def map_points_to_n_imgs(points_2d_batch, grid_size, grid_range, device=None):
    """ref utils.py:198-223 -- [n,m,2] -> [n,S0,S1] fp32 point counts."""
    _need_cuda(points_2d_batch, "map_points_to_n_imgs")
    p = points_2d_batch.contiguous().float()
    n, m, _ = p.shape
    out = torch.empty((n, int(grid_size[0]), int(grid_size[1])), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.lib().nbp_map_points_to_imgs_f32(p.data_ptr(), n, m, int(grid_size[0]), int(grid_size[1]),
                                                   float(grid_range[0]), float(grid_range[1]), out.data_ptr(),
                                                   _lib.current_stream())
    _lib.check(rc, "nbp_map_points_to_imgs_f32")
    return out


class CloudBins:
    """Tile-binned shadow copy of ONE cloud tensor (include/nbp_hip.h "Tile-binned shadow copy"): a map build files the points it
    has not seen yet into 2048-point pages of their 2.5-unit (x, z) tile and builds the six maps from the pages -- one workgroup
    per page on a dense LDS histogram, tiles outside the window never read.  The maps are bit-identical to the
    unbinned kernel's.  `lo_xz` / `hi_xz`: the scene's horizontal extent (points outside it are still counted, through a slower
    side list).  The store follows the cloud from zero points: call reset() whenever the cloud is emptied."""

    def __init__(self, lo_xz, hi_xz, capacity, device):
        L = _lib.lib()
        self.lo = (C.c_float * 2)(float(lo_xz[0]), float(lo_xz[1]))
        self.hi = (C.c_float * 2)(float(hi_xz[0]), float(hi_xz[1]))
        self.capacity = int(capacity)
        nbytes = int(L.nbp_cloud_bins_bytes(self.lo, self.hi, self.capacity))
        if nbytes == 0:
            raise ValueError(f"CloudBins: bad extent {tuple(lo_xz)} .. {tuple(hi_xz)} / capacity {capacity}")
        g, t = (C.c_int * 3)(), (C.c_float * 3)()
        _lib.check(L.nbp_cloud_bins_geometry(self.lo, self.hi, self.capacity, g, t), "nbp_cloud_bins_geometry")
        self.nx, self.nz, self.max_pages = int(g[0]), int(g[1]), int(g[2])
        self.x0, self.z0, self.tile = float(t[0]), float(t[1]), float(t[2])
        self.store = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.reset()

    def reset(self):
        with torch.cuda.device(self.store.device):
            rc = _lib.lib().nbp_cloud_bins_init(self.store.data_ptr(), self.store.numel(), self.lo, self.hi, self.capacity,
                                                _lib.current_stream())
        _lib.check(rc, "nbp_cloud_bins_init")

    def page_bound(self, n_upper):
        """Page workgroups to launch for a cloud of at most n_upper points: the full pages plus room for 512 tiles' partial ones (a
        rollout's cloud sits on a few hundred tiles).  NOT a correctness bound: the kernel's page workgroups stride over every
        page that exists, so a cloud on more tiles than that only costs time."""
        return max(1, min(self.max_pages, int(n_upper) // 2048 + min(self.nx * self.nz, 512) + 1))

    def header(self):
        """{n_pages, n_overflow, error, n_binned} (synchronises: tests and end-of-rollout checks only)."""
        h = self.store[:24].cpu()
        w = h.view(torch.int32)
        return {"n_pages": int(w[0]), "n_overflow": int(w[1]), "error": int(w[2]), "n_binned": int(h[16:24].view(torch.int64)[0])}


def accumulate_step_maps(full_pc, camera_pose, y_bins, grid_size=256, grid_range=(-40, 40), band=0.1, n_dev=None,
                         out=None, bins=None, prefiled=False):
    """One fused pass replacing nbp_planning.py:114-127 + :172-183.

    Returns [6,S,S]: four height slabs (torch.bucketize(p_y, y_bins[:-1]) - 1 semantics),
    the points in no slab, and the +-`band` height band around the camera.
    """
    _need_cuda(full_pc, "accumulate_step_maps")
    cx, cy, cz = _pose_xyz(camera_pose)
    yb = [float(v) for v in (y_bins.tolist() if isinstance(y_bins, torch.Tensor) else y_bins)]
    bounds = yb[:-1]
    if len(bounds) > 8:
        raise ValueError("at most 8 slab boundaries")
    S = int(grid_size)
    if out is None:
        out = torch.empty(6, S, S, dtype=torch.float32, device=full_pc.device)
    p = full_pc.contiguous().float()
    n_dev = None if n_dev is None else n_dev.data_ptr()
    arr = (C.c_float * max(len(bounds), 1))(*bounds)
    # thresholds exactly as the reference forms them: python double +-0.1, then fp32 compare
    band_hi = torch.tensor(cy + band, dtype=torch.float32).item()
    band_lo = torch.tensor(cy - band, dtype=torch.float32).item()
    with torch.cuda.device(p.device):
        if bins is not None:          # on the tile-binned shadow copy of THIS cloud tensor (CloudBins): same maps, bit for bit
            if p.data_ptr() != full_pc.data_ptr():
                raise ValueError("accumulate_step_maps(bins=...): the cloud must be the contiguous fp32 tensor the bins follow")
            # prefiled: the launch that appended the points filed them and cleared `out` (hipops.unproject_append(bins=, clear=)):
            # the page build alone, one launch
            fn = _lib.lib().nbp_step_maps_prefiled_f32 if prefiled else _lib.lib().nbp_step_maps_binned_f32
            rc = fn(bins.store.data_ptr(), bins.page_bound(p.shape[0]), p.data_ptr(), p.shape[0], n_dev,
                    cx, cy, cz, arr, len(bounds), band_lo, band_hi, S, float(grid_range[0]),
                    float(grid_range[1]), None, 0, None, 0, out.data_ptr(), None, _lib.current_stream())
            _lib.check(rc, "nbp_step_maps_prefiled_f32" if prefiled else "nbp_step_maps_binned_f32")
            return out
        rc = _lib.lib().nbp_map_accumulate_f32(p.data_ptr(), p.shape[0], n_dev, cx, cy, cz, arr, len(bounds), band_lo,
                                               band_hi, S, float(grid_range[0]), float(grid_range[1]),
                                               out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "nbp_map_accumulate_f32")
    return out


def step_maps(full_pc, camera_pose, y_bins, grid_size, grid_range, traj_dev, n_traj_old, traj_fresh, out6, net_in5,
              band=0.1, n_dev=None, bins=None, n_upper=None, prefiled=False):
    """accumulate_step_maps + the trajectory channel + the copy of the four slabs into the network input, in one call
    (nbp_step_maps_f32: two memsets, one kernel, one copy instead of seven launches).  traj_dev: device [cap,3] history
    of camera positions, n_traj_old of them valid; traj_fresh: host [k<=8,3] new positions, appended by the kernel.
    out6 [6,S,S]; net_in5 [5,S,S] (one map of the network's input batch)."""
    _need_cuda(full_pc, "step_maps")
    cx, cy, cz = _pose_xyz(camera_pose)
    bounds = [float(v) for v in (y_bins.tolist() if isinstance(y_bins, torch.Tensor) else y_bins)][:-1]
    if len(bounds) > 8:
        raise ValueError("at most 8 slab boundaries")
    S = int(grid_size)
    if tuple(out6.shape) != (6, S, S) or tuple(net_in5.shape) != (5, S, S) or not net_in5.is_contiguous():
        raise ValueError("step_maps: out6 [6,S,S] and a contiguous net_in5 [5,S,S] expected")
    arr = (C.c_float * max(len(bounds), 1))(*bounds)
    fresh = np.ascontiguousarray(np.asarray(traj_fresh, np.float32).reshape(-1, 3))
    if n_traj_old + len(fresh) > traj_dev.shape[0]:
        raise ValueError("step_maps: the trajectory buffer is too small")
    # thresholds exactly as the reference forms them: python double +-0.1, then fp32 compare
    band_hi, band_lo = float(np.float32(cy + band)), float(np.float32(cy - band))
    if bins is not None:              # the tile-binned shadow copy of this cloud (CloudBins): same maps, bit for bit
        nu = full_pc.shape[0] if n_upper is None else min(int(n_upper), full_pc.shape[0])
        # prefiled: every point was filed by the launch that appended it, and that launch cleared out6 and the trajectory channel
        # (hipops.unproject_append(bins=, clear=(out6, net_in5[4]))): the page build alone
        fn = _lib.lib().nbp_step_maps_prefiled_f32 if prefiled else _lib.lib().nbp_step_maps_binned_f32
        rc = fn(bins.store.data_ptr(), bins.page_bound(nu), full_pc.data_ptr(), full_pc.shape[0],
                None if n_dev is None else n_dev.data_ptr(), cx, cy, cz, arr, len(bounds), band_lo,
                band_hi, S, float(grid_range[0]), float(grid_range[1]), traj_dev.data_ptr(),
                int(n_traj_old), fresh.ctypes.data, len(fresh), out6.data_ptr(), net_in5.data_ptr(),
                _lib.current_stream())
        _lib.check(rc, "nbp_step_maps_prefiled_f32" if prefiled else "nbp_step_maps_binned_f32")
        return
    rc = _lib.lib().nbp_step_maps_f32(full_pc.data_ptr(), full_pc.shape[0], None if n_dev is None else n_dev.data_ptr(),
                                      cx, cy, cz, arr, len(bounds), band_lo, band_hi, S, float(grid_range[0]),
                                      float(grid_range[1]), traj_dev.data_ptr(), int(n_traj_old), fresh.ctypes.data,
                                      len(fresh), out6.data_ptr(), net_in5.data_ptr(), _lib.current_stream())
    _lib.check(rc, "nbp_step_maps_f32")


def step_maps_batch(items, grid_size, grid_range, out6_all, net_in_all, band=0.1):
    """step_maps for the rollouts of a lock-step group in ONE kernel launch (nbp_step_maps_batch_f32): `items` = one tuple per
    rollout (full_pc, n_upper, n_dev, camera_pose, y_bins, traj_dev, n_traj_old, traj_fresh[, bins]); rollout i writes out6_all[i]
    ([n,6,S,S]) and net_in_all[i] ([n,5,S,S]); n_upper = a host-side upper bound of the cloud size (sizes the grid).  With a
    ninth entry (CloudBins) in EVERY item the group runs on the binned shadow copies (nbp_step_maps_binned_batch_f32)."""
    n, S = len(items), int(grid_size)
    if n > 16:                                   # the kernel arguments hold 16 rollouts: larger groups go in chunks
        for i in range(0, n, 16):
            step_maps_batch(items[i:i + 16], grid_size, grid_range, out6_all[i:i + 16], net_in_all[i:i + 16], band)
        return
    if n < 1:
        raise ValueError("step_maps_batch: at least one rollout")
    if tuple(out6_all.shape) != (n, 6, S, S) or tuple(net_in_all.shape) != (n, 5, S, S) or not out6_all.is_contiguous() \
            or not net_in_all.is_contiguous():
        raise ValueError("step_maps_batch: contiguous out6_all [n,6,S,S] and net_in_all [n,5,S,S] expected")
    _need_cuda(out6_all, "step_maps_batch")
    VP, LL = C.c_void_p, C.c_longlong
    pts, ncap, ndev, traj = (VP * n)(), (LL * n)(), (VP * n)(), (VP * n)()
    poses, bounds, nb = np.zeros((n, 3), np.float32), np.zeros((n, 8), np.float32), (C.c_int * n)()
    bands, fresh, n_old, n_fresh = np.zeros((n, 2), np.float32), np.zeros((n, 24), np.float32), (C.c_int * n)(), (C.c_int * n)()
    binned = all(len(it) > 8 and it[8] is not None for it in items)
    stores, pbound = (VP * n)(), (C.c_int * n)()
    for i, it in enumerate(items):
        full_pc, n_upper, n_dev, pose, y_bins, traj_dev, n_traj_old, traj_fresh = it[:8]
        if binned:
            stores[i], pbound[i] = it[8].store.data_ptr(), it[8].page_bound(min(int(n_upper), full_pc.shape[0]))
        cx, cy, cz = _pose_xyz(pose)
        b = [float(v) for v in (y_bins.tolist() if isinstance(y_bins, torch.Tensor) else y_bins)][:-1]
        if len(b) > 8:
            raise ValueError("at most 8 slab boundaries")
        f = np.asarray(traj_fresh, np.float32).reshape(-1)
        if n_traj_old + len(f) // 3 > traj_dev.shape[0]:
            raise ValueError("step_maps_batch: the trajectory buffer is too small")
        pts[i], ncap[i], ndev[i], traj[i] = full_pc.data_ptr(), min(int(n_upper), full_pc.shape[0]), n_dev.data_ptr(), traj_dev.data_ptr()
        poses[i] = (cx, cy, cz)
        bounds[i, :len(b)] = b
        nb[i] = len(b)
        bands[i] = (np.float32(cy - band), np.float32(cy + band))       # as the reference forms them: python double +-0.1, then fp32
        fresh[i, :len(f)] = f
        n_old[i], n_fresh[i] = int(n_traj_old), len(f) // 3
    if binned:
        rc = _lib.lib().nbp_step_maps_binned_batch_f32(n, stores, pbound, pts, ncap, ndev, poses.ctypes.data, bounds.ctypes.data, nb,
                                                       bands.ctypes.data, S, float(grid_range[0]), float(grid_range[1]), traj, n_old,
                                                       fresh.ctypes.data, n_fresh, out6_all.data_ptr(), net_in_all.data_ptr(),
                                                       _lib.current_stream())
        _lib.check(rc, "nbp_step_maps_binned_batch_f32")
        return
    rc = _lib.lib().nbp_step_maps_batch_f32(n, pts, ncap, ndev, poses.ctypes.data, bounds.ctypes.data, nb, bands.ctypes.data, S,
                                            float(grid_range[0]), float(grid_range[1]), traj, n_old, fresh.ctypes.data, n_fresh,
                                            out6_all.data_ptr(), net_in_all.data_ptr(), _lib.current_stream())
    _lib.check(rc, "nbp_step_maps_batch_f32")
