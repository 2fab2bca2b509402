"""numpy fp32 restatement of the proxy points' view-state vectors (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows macarons/utility/scone_utils.py:799-862 (compute_view_state), macarons/utility/CustomGeometry.py:27-45
(get_spherical_coords), macarons/utility/utils.py:113-117 (floor_divide) and Scene.update_proxy_view_states
(macarons/utility/macarons_utils.py:3268-3327) as the NBV driver calls it (macarons/testers/scene.py:598-601).
Pinned by tests/golden/viewstate.npz (outputs of the reference functions themselves).  The direction of a ray is binned
into n_elev x n_azim cells; asin / acos are evaluated by numpy here and by Sleef inside torch, so a ray within a few ulp
of a bin boundary may fall on either side: `boundary_distance` returns the margin so that tests can say which rays those are.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _pymod(x, d):
    """torch's `tensor % scalar` (remainder with the divisor's sign), fp32"""
    return np.remainder(x.astype(f32), f32(d)).astype(f32)


def spherical_coords(X):
    """CustomGeometry.get_spherical_coords: (r, elev, azim), fp32."""
    X = np.asarray(X, f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.sqrt((X[:, 0] * X[:, 0] + X[:, 1] * X[:, 1]) + X[:, 2] * X[:, 2], dtype=f32)
        s = (X[:, 1] / r).astype(f32)
        elev = np.arcsin(s, dtype=f32)
        elev[s <= -1] = f32(-np.pi / 2)
        elev[s >= 1] = f32(np.pi / 2)
        c = (X[:, 2] / (r * np.cos(elev, dtype=f32))).astype(f32)
        azim = np.arccos(c, dtype=f32)
        azim[c <= -1] = f32(np.pi)
        azim[c >= 1] = f32(0.0)
    azim[X[:, 0] < 0] *= f32(-1)
    return r, elev, azim


def view_bins(pts, X_view, n_elev, n_azim, return_angles=False):
    """[n_pts, n_view] int64 bin index of the ray from every point to every camera (compute_view_state's `indices`)."""
    pts, X_view = np.asarray(pts, f32)[:, :3], np.asarray(X_view, f32).reshape(-1, 3)
    n_pts, n_view = len(pts), len(X_view)
    rays = (X_view[None, :, :] - pts[:, None, :]).reshape(-1, 3)
    _, elev, azim = spherical_coords(rays)
    elev_step, azim_step = np.pi / (n_elev + 1), 2 * np.pi / n_azim
    me, ma = _pymod(elev, elev_step), _pymod(azim, azim_step)
    idx_e = ((elev - me) / f32(elev_step)).astype(f32)            # floor_divide = (x - x % d) / d
    idx_a = ((azim - ma) / f32(azim_step)).astype(f32)
    idx_e[me > f32(elev_step / 2.0)] += 1
    idx_a[ma > f32(azim_step / 2.0)] += 1
    idx_e[idx_e >= n_elev] = n_elev - 1
    idx_e[idx_e < -n_elev // 2] = -n_elev // 2                     # (-n) // 2, as the reference's precedence has it
    idx_a[idx_a > n_azim // 2] = -n_azim // 2
    idx_e += n_elev // 2
    idx_a[idx_a < 0] += n_azim
    ind = (idx_e.astype(np.int64) * n_azim + idx_a.astype(np.int64)) % (n_elev * n_azim)
    ind = ind.reshape(n_pts, n_view)
    if return_angles:
        return ind, elev.reshape(n_pts, n_view), azim.reshape(n_pts, n_view)
    return ind


def compute_view_state(pts, X_view, n_elev, n_azim):
    """[n_pts, n_elev * n_azim] float32 in {0, 1} (the reference's [n_cloud, seq_len, .] with n_cloud = 1 squeezed)."""
    ind = view_bins(pts, X_view, n_elev, n_azim)
    vs = np.zeros((len(ind), n_elev * n_azim), f32)
    vs[np.arange(len(ind))[:, None], ind] = 1.0
    return vs


def boundary_distance(pts, X_view, n_elev, n_azim):
    """[n_pts, n_view] distance (radians, float64 evaluation) of each ray's (elev, azim) to the nearest rounding boundary of its
    bin -- rays closer than a few fp32 ulp of the angle may be binned differently by another libm."""
    pts, X_view = np.asarray(pts, np.float64)[:, :3], np.asarray(X_view, np.float64).reshape(-1, 3)
    rays = X_view[None] - pts[:, None]
    r = np.linalg.norm(rays, axis=2)
    elev = np.arcsin(np.clip(rays[..., 1] / r, -1, 1))
    azim = np.arctan2(rays[..., 0], rays[..., 2])
    es, as_ = np.pi / (n_elev + 1), 2 * np.pi / n_azim
    de = np.abs(np.remainder(elev, es) - es / 2)
    da = np.abs(np.remainder(azim, as_) - as_ / 2)
    return np.minimum(de, da)


def update_proxy_view_states(view_states, proxy_points, fov_mask, signed_distances, X_cam, n_elev, n_azim, distance_to_surface):
    """Scene.update_proxy_view_states with signed distances: the vectors of the proxy points inside the field of view whose
    signed distance is below distance_to_surface (default 3 x distance_between_proxy_points, mu:3298-3299) gain the bit of
    the direction towards X_cam; view_states [P, n_elev * n_azim] is updated in place (sum + heaviside(., 0) == OR).
    signed_distances: [P] (entries outside fov_mask are ignored)."""
    upd = np.asarray(fov_mask, bool) & (np.asarray(signed_distances, f32) < f32(distance_to_surface))
    idx = np.nonzero(upd)[0]
    if len(idx):
        ind = view_bins(np.asarray(proxy_points, f32)[idx], np.asarray(X_cam, f32).reshape(1, 3), n_elev, n_azim)[:, 0]
        view_states[idx, ind] = 1.0
    return upd


def view_gain(proxy_points, occ, view_states, cams_RT, X_cams, n_elev, n_azim, H, W, fov_range):
    """The product's geometric coverage-gain model (include/nbp_hip.h::nbp_view_gain_i32; the reference's predictor is the
    unreleased SCONE network): per candidate, the occupied proxy points in its field of view whose view-state bit for the
    direction towards the candidate is 0."""
    from . import camera as ocam
    P = np.asarray(proxy_points, f32)
    out = []
    for (R, T), X in zip(cams_RT, X_cams):
        inf = ocam.points_in_fov(P, R, T, H, W, fov_range) & (np.asarray(occ, f32).reshape(-1) > 0.5)
        idx = np.nonzero(inf)[0]
        b = view_bins(P[idx], np.asarray(X, f32).reshape(1, 3), n_elev, n_azim)[:, 0] if len(idx) else np.zeros(0, np.int64)
        out.append(int((np.asarray(view_states)[idx, b] == 0).sum()))
    return out
