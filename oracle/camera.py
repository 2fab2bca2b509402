"""numpy restatement of the camera arithmetic on the NBP path.  PARITY UNPINNED against the
libraries: pytorch3d 0.7.4 (environment.yml:204) is a third-party dependency absent from
/root/reference and from this image; this file restates its documented conventions
(row vectors, X_view = X_world R + T; NDC +X left, +Y up, short side in [-1,1];
FoVPerspectiveCameras defaults fov=60 deg, znear=1, aspect 1) at the reference's call sites:

  * get_camera_RT (macarons/utility/macarons_utils.py:940-957) = get_cartesian_coords
    (macarons/utility/CustomGeometry.py:5-24) + pytorch3d look_at_view_transform(eye, at)
  * Camera NDC tables (macarons_utils.py:2270-2279) + project_depth_in_3D (:2788-2809) +
    compute_partial_point_cloud (:2811-2847)
  * Camera.__init__ pose lattice (:2283-2327) and update_camera interpolation (:2590-2632)

fp32 operation order matches nextbestpath_amd/csrc/nbp_sim.hip exactly (parity with the HIP
path is bit-exact); known-answer tests (tests/test_oracle_sim.py) validate the geometry."""
import numpy as np

from . import sampling

f32 = np.float32
TAN_HALF_FOV = f32(np.tan(np.deg2rad(30.0)))       # FoVPerspectiveCameras default fov = 60 deg


def cartesian(r, elev_deg, azim_deg):
    """CustomGeometry.get_cartesian_coords: x = r cos(e) sin(a), y = r sin(e), z = r cos(e) cos(a)."""
    e, a = np.deg2rad(elev_deg), np.deg2rad(azim_deg)
    return np.stack([r * np.cos(e) * np.sin(a), r * np.sin(e), r * np.cos(e) * np.cos(a)], -1)


def look_at_RT(eye, at, up=(0.0, 1.0, 0.0), direction=None):
    """pytorch3d look_at_view_transform: z = normalize(at-eye), x = normalize(up x z), y = z x x;
    R has those axes as COLUMNS (row-vector convention), T = -eye R.  `direction` (= at - eye in exact arithmetic)
    can be given directly: forming at = eye + d and subtracting eye again perturbs d by ~1e-16 in float64 (by ~1e-7 in
    the library's fp32), which only shows where a rotation entry is exactly 0."""
    eye, at, up = np.asarray(eye, np.float64), np.asarray(at, np.float64), np.asarray(up, np.float64)
    z = at - eye if direction is None else np.asarray(direction, np.float64)
    z = z / max(np.linalg.norm(z), 1e-5)
    x = np.cross(up, z)
    nx = np.linalg.norm(x)
    if nx < 5e-3:                       # up parallel to view axis: pytorch3d falls back to y x z
        y0 = np.cross(z, np.array([1.0, 0.0, 0.0]))
        y0 = y0 / np.linalg.norm(y0)
        x = np.cross(y0, z)
        nx = np.linalg.norm(x)
    x = x / nx
    y = np.cross(z, x)
    y = y / max(np.linalg.norm(y), 1e-5)
    R = np.stack([x, y, z], axis=1)
    T = -(eye @ R)
    return R.astype(f32), T.astype(f32)


def camera_RT(X_cam, V_cam):
    """get_camera_RT (mu:940-957): rays = -cartesian(1, -elev, 180 + azim); look at X + rays."""
    rays = -cartesian(1.0, -float(V_cam[0]), 180.0 + float(V_cam[1]))
    X = np.asarray(X_cam, np.float64)
    return look_at_RT(X, X + rays, direction=rays)


def ndc_tables(H, W):
    s = min(H, W)
    col = np.arange(W, dtype=f32)[None, :].repeat(H, 0)
    row = np.arange(H, dtype=f32)[:, None].repeat(W, 1)
    ndc_x = f32(W / s) - (col / f32(s - 1)) * f32(2)
    ndc_y = f32(H / s) - (row / f32(s - 1)) * f32(2)
    return ndc_x, ndc_y


def unproject(depth, R, T):
    """All pixels: [H,W] view-space depth -> [H*W,3] world points (fp32, kernel op order)."""
    H, W = depth.shape
    ndc_x, ndc_y = ndc_tables(H, W)
    z = depth.astype(f32)
    xv = (ndc_x * z) * TAN_HALF_FOV
    yv = (ndc_y * z) * TAN_HALF_FOV
    dx, dy, dz = xv - f32(T[0]), yv - f32(T[1]), z - f32(T[2])
    R = R.astype(f32)
    out = np.empty((H, W, 3), f32)
    for j in range(3):
        out[..., j] = (dx * R[j, 0] + dy * R[j, 1]) + dz * R[j, 2]
    return out.reshape(-1, 3)


def partial_point_cloud(depth, mask, R, T, gathering_factor, fov_range, seed, frame_index=0, rgb=None):
    """compute_partial_point_cloud with the seeded bijection instead of torch.randperm.
    Returns (points [n_keep,3], n_valid), or (points, n_valid, colours [n_keep,3]) when the image rgb [H,W,3] is given."""
    H, W = depth.shape
    m = (depth > -1) if mask is None else (mask != 0)
    valid = m.reshape(-1) & (depth.reshape(-1) < f32(fov_range))
    lst = np.nonzero(valid)[0]
    n_valid = len(lst)
    n_keep = int(n_valid * gathering_factor)
    sd = (seed + 0x632BE5AB * (frame_index + 1)) & sampling.M32
    sel = lst[sampling.perm_index(np.arange(n_keep), n_valid, sd)] if n_keep else lst[:0]
    if rgb is not None:
        return unproject(depth, R, T)[sel], n_valid, np.asarray(rgb, f32).reshape(-1, 3)[sel]
    return unproject(depth, R, T)[sel], n_valid


def pose_lattice(x_min, pose_l, pose_w, pose_h, n_elev, n_azim):
    """Camera.__init__ (mu:2283-2327): poses [L,W,H,E,A,5] (x,y,z,elev,azim), i-major order.
    NOTE the reference offsets from the *scene* x_min argument (not self.x_min = x_min + 3)."""
    x_min = np.asarray(x_min, f32)
    idx = np.stack(np.meshgrid(np.arange(pose_l), np.arange(pose_w), np.arange(pose_h), np.arange(n_elev),
                               np.arange(n_azim), indexing="ij"), -1).reshape(-1, 5)
    poses = np.zeros((len(idx), 5), f32)
    poses[:, 0] = x_min[0] + (idx[:, 0] * 3).astype(f32)
    poses[:, 1] = x_min[1] + f32(3.3)
    poses[:, 2] = x_min[2] + (idx[:, 2] * 3).astype(f32)
    poses[:, 3] = f32(-90.0) + (f32(180.0) * (1 + idx[:, 3]).astype(f32)) / f32(n_elev + 1)
    poses[:, 4] = (f32(360.0) * idx[:, 4].astype(f32)) / f32(n_azim)
    return idx, poses


def camera_center(R, T):
    """World position of a camera given (R, T) of X_view = X_world R + T: C = -T R^T (fp32, the op order of points_in_fov)."""
    R, T = np.asarray(R, f32), np.asarray(T, f32)
    return np.array([-((T[0] * R[j, 0] + T[1] * R[j, 1]) + T[2] * R[j, 2]) for j in range(3)], f32)


def points_in_fov(pts, R, T, H, W, fov_range):
    """Camera.get_points_in_fov (mu:2849-2884): boolean mask (fp32, kernel op order; same test as carve_update)."""
    s = min(H, W)
    R, T = np.asarray(R, f32), np.asarray(T, f32)
    p = np.asarray(pts, f32)
    v = np.empty_like(p)
    for j in range(3):
        v[:, j] = ((p[:, 0] * R[0, j] + p[:, 1] * R[1, j]) + p[:, 2] * R[2, j]) + T[j]
    C = np.array([-((T[0] * R[j, 0] + T[1] * R[j, 1]) + T[2] * R[j, 2]) for j in range(3)], f32)
    d = p - C
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        nx = v[:, 0] / (v[:, 2] * TAN_HALF_FOV)
        ny = v[:, 1] / (v[:, 2] * TAN_HALF_FOV)
    max_x = f32(W / s); min_x = max_x - (f32(W - 1) / f32(s - 1)) * f32(2)
    max_y = f32(H / s); min_y = max_y - (f32(H - 1) / f32(s - 1)) * f32(2)
    return (nx >= min_x) & (nx <= max_x) & (ny >= min_y) & (ny <= max_y) & (v[:, 2] > 0) & (dist < f32(fov_range))


def carve_update(pts, depth, mask, R, T, zfar, fov_range, tol, score_thr, n_inside, n_behind, occ, out_of_field):
    """Depth-map space carving (A20): Camera.get_points_in_fov (mu:2849-2884) +
    get_signed_distance_to_depth_maps (mu:2900-2949; F.grid_sample bilinear, border padding,
    align_corners=False) + Scene.update_proxy_supervision_occ / update_proxy_out_of_field
    (mu:3329-3363).  Arrays are updated in place; fp32, kernel op order."""
    H, W = depth.shape
    s = min(H, W)
    R, T = np.asarray(R, f32), np.asarray(T, f32)
    p = np.asarray(pts, f32)
    v = np.empty_like(p)
    for j in range(3):
        v[:, j] = ((p[:, 0] * R[0, j] + p[:, 1] * R[1, j]) + p[:, 2] * R[2, j]) + T[j]
    C = np.array([-((T[0] * R[j, 0] + T[1] * R[j, 1]) + T[2] * R[j, 2]) for j in range(3)], f32)
    d = p - C
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        nx = v[:, 0] / (v[:, 2] * TAN_HALF_FOV)
        ny = v[:, 1] / (v[:, 2] * TAN_HALF_FOV)
    max_x = f32(W / s); min_x = max_x - (f32(W - 1) / f32(s - 1)) * f32(2)
    max_y = f32(H / s); min_y = max_y - (f32(H - 1) / f32(s - 1)) * f32(2)
    inf = (nx >= min_x) & (nx <= max_x) & (ny >= min_y) & (ny <= max_y) & (v[:, 2] > 0) & (dist < f32(fov_range))
    gx = (-f32(s) / f32(W)) * nx
    gy = (-f32(s) / f32(H)) * ny
    ix = np.clip(((gx + f32(1)) * f32(W) - f32(1)) * f32(0.5), 0, W - 1).astype(f32)
    iy = np.clip(((gy + f32(1)) * f32(H) - f32(1)) * f32(0.5), 0, H - 1).astype(f32)
    ix, iy = np.where(inf, ix, 0).astype(f32), np.where(inf, iy, 0).astype(f32)
    x0, y0 = np.floor(ix).astype(int), np.floor(iy).astype(int)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    wx, wy = ix - x0.astype(f32), iy - y0.astype(f32)
    ok = (depth > -1) if mask is None else (mask != 0)
    dd = np.where(ok, depth, f32(1.1) * f32(zfar)).astype(f32)
    one = f32(1)
    ds = (dd[y0, x0] * (one - wx) * (one - wy) + dd[y0, x1] * wx * (one - wy)) + \
         (dd[y1, x0] * (one - wx) * wy + dd[y1, x1] * wx * wy)
    sd = v[:, 2] - ds
    n_inside[inf] += 1
    n_behind[inf] += (sd[inf] >= -f32(tol)).astype(f32)
    occ[inf] = ((n_behind[inf] / n_inside[inf]) >= f32(score_thr)).astype(f32)
    out_of_field[inf] = 0
    return inf, sd
