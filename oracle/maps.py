"""numpy restatement of the map accumulation (fp32 arithmetic, bit-exact cells).

Follows next_best_path/utility/utils.py: get_point_position_in_the_img :160-164,
transform_points_to_n_pieces :166-196 (no_rotation=True => R = I, so the matmul reduces to
one fp32 subtraction per coordinate), map_points_to_n_imgs :198-223; and the slab split /
projections of next_best_path/testers/nbp_planning.py:114-127,172-183.
Pinned by tests/golden/maps_*.npz (outputs of the reference functions themselves).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def transform_points_to_n_pieces(points, camera_pose):
    p = np.asarray(points, dtype=f32).reshape(-1, 3)
    cx, cz = f32(camera_pose[0]), f32(camera_pose[2])
    out = np.empty((1, p.shape[0], 2), dtype=f32)
    out[0, :, 0] = -(p[:, 2] - cz)
    out[0, :, 1] = -(p[:, 0] - cx)
    return out


def _cells(v, size, grid_range):
    lo, hi = grid_range
    scale = f32(size / (hi - lo))           # python double division, then fp32 (torch scalar rule)
    return np.rint((np.asarray(v, dtype=f32) - f32(lo)) * scale)


def get_point_position_in_the_img(points_2d, grid_size, grid_range):
    p = np.asarray(points_2d, dtype=f32)
    x = _cells(p[..., 0], grid_size[0], grid_range).astype(np.int64)
    y = _cells(p[..., 1], grid_size[1], grid_range).astype(np.int64)
    return np.squeeze(np.stack((x, y)))


def map_points_to_n_imgs(points_2d_batch, grid_size, grid_range):
    p = np.asarray(points_2d_batch, dtype=f32)
    n, m, _ = p.shape
    out = np.zeros((n,) + tuple(grid_size), dtype=f32)
    fx = _cells(p[..., 0], grid_size[0], grid_range)
    fy = _cells(p[..., 1], grid_size[1], grid_range)
    ok = (fx >= 0) & (fx < grid_size[0]) & (fy >= 0) & (fy < grid_size[1])
    img = np.broadcast_to(np.arange(n)[:, None], (n, m))[ok]
    np.add.at(out, (img, fx[ok].astype(np.int64), fy[ok].astype(np.int64)), f32(1))
    return out


def slab_bins(y, y_bins):
    """torch.bucketize(y, y_bins[:-1]) - 1 with right=False == searchsorted(side='left') - 1."""
    bounds = np.asarray(y_bins, dtype=f32)[:-1]
    return np.searchsorted(bounds, np.asarray(y, dtype=f32), side="left") - 1


def accumulate_step_maps(full_pc, camera_pose, y_bins, S=256, grid_range=(-40, 40), band=0.1, n_pieces=4):
    """[6,S,S]: 4 slabs, 'no slab' remainder, +-band height band (see include/nbp_hip.h)."""
    p = np.asarray(full_pc, dtype=f32).reshape(-1, 3)
    out = np.zeros((6, S, S), dtype=f32)
    if p.shape[0] == 0:
        return out
    bins = slab_bins(p[:, 1], y_bins)
    for k in range(n_pieces):
        sel = p[bins == k]
        if len(sel):
            out[k] = map_points_to_n_imgs(transform_points_to_n_pieces(sel, camera_pose), (S, S), grid_range)[0]
    rest = p[(bins < 0) | (bins >= n_pieces)]
    if len(rest):
        out[4] = map_points_to_n_imgs(transform_points_to_n_pieces(rest, camera_pose), (S, S), grid_range)[0]
    cy = float(f32(camera_pose[1]))
    hi_t, lo_t = f32(cy + band), f32(cy - band)
    sel = p[(p[:, 1] < hi_t) & (p[:, 1] > lo_t)]
    if len(sel):
        out[5] = map_points_to_n_imgs(transform_points_to_n_pieces(sel, camera_pose), (S, S), grid_range)[0]
    return out
