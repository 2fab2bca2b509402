"""Camera for the exploration rollout -- host-side mirror of the reference's ``Camera``
(macarons/utility/macarons_utils.py:2193-2949) restricted to what the NBP drivers use.

Differences that are deliberate:
  * the pose lattice is index based (the reference keys a dict by ``str(list(np.int64 row))``,
    which breaks under numpy >= 2);
  * frames stay on the device (the reference torch.save()s every frame and torch.load()s it
    back: nbp_planning.py:66,271-282); rendering is the HIP tile rasteriser, un-projection the
    HIP kernel, both through libnbp_hip.so;
  * the 4 interpolated frames of one move are rasterised in ONE launch.
Pose arithmetic follows the reference line by line (fp32): lattice :2283-2327, interpolation
and azimuth wrap :2590-2632, look-at :940-957 (pytorch3d look_at_view_transform conventions).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib
from ..utility import hipops

f32 = np.float32


def _cartesian(elev_deg, azim_deg):
    e, a = np.deg2rad(elev_deg), np.deg2rad(azim_deg)
    return np.array([np.cos(e) * np.sin(a), np.sin(e), np.cos(e) * np.cos(a)])


_R_CACHE = {}


def _look_rotation(V_cam):
    """Rotation of get_camera_RT (mu:940-957): look along -cartesian(1, -elev, 180 + azim); float64.
    pytorch3d convention: R columns = camera x / y / z axes (row vectors)."""
    z = -_cartesian(-float(V_cam[0]), 180.0 + float(V_cam[1]))
    z = z / max(np.linalg.norm(z), 1e-5)
    up = np.array([0.0, 1.0, 0.0])
    x = np.cross(up, z)
    nx = np.linalg.norm(x)
    if nx < 5e-3:                                  # looking straight up / down
        y0 = np.cross(z, np.array([1.0, 0.0, 0.0]))
        x = np.cross(y0 / np.linalg.norm(y0), z)
        nx = np.linalg.norm(x)
    x = x / nx
    y = np.cross(z, x)
    y = y / max(np.linalg.norm(y), 1e-5)
    return np.stack([x, y, z], axis=1)


def camera_RT(X_cam, V_cam):
    """get_camera_RT (mu:940-957): R as above, T = -X R.  The rotation depends on (elev, azim) only and the lattice has
    a few dozen distinct view directions (8 azimuths x 4 interpolation steps): it is memoised (the numpy cross / norm
    calls were 20 % of the host time of a step); the arithmetic is unchanged."""
    key = (float(V_cam[0]), float(V_cam[1]))
    hit = _R_CACHE.get(key)
    if hit is None:
        R64 = _look_rotation(V_cam)
        hit = (R64, R64.astype(f32))
        if len(_R_CACHE) < 4096:
            _R_CACHE[key] = hit
    X = np.asarray(X_cam, np.float64).reshape(3)
    return hit[1], (-(X @ hit[0])).astype(f32)


class Camera:
    def __init__(self, x_min, x_max, pose_l, pose_w, pose_h, pose_n_elev, pose_n_azim, n_interpolation_steps,
                 zfar, image_height, image_width, device, gathering_factor=0.05, sensor_range=70.0, seed=0,
                 ambient_light_intensity=0.85, contrast_factor=1.0, render_rgb=True):
        self.x_min_arg = np.asarray(x_min, f32)
        self.x_min = self.x_min_arg + f32(3)             # mu:2230-2231 (kept for parity of attributes)
        self.x_max = np.asarray(x_max, f32) - f32(3)
        self.pose_l, self.pose_w, self.pose_h = int(pose_l), int(pose_w), int(pose_h)
        self.pose_n_elev, self.pose_n_azim = int(pose_n_elev), int(pose_n_azim)
        self.n_interpolation_steps = int(n_interpolation_steps)
        self.zfar = zfar
        self.image_height, self.image_width = int(image_height), int(image_width)
        self.device = device
        self.gathering_factor, self.sensor_range = gathering_factor, sensor_range
        self.seed = int(seed)
        self.ambient, self.contrast_factor = float(ambient_light_intensity), float(contrast_factor)
        self.render_rgb = render_rgb and _lib.tune("NBP_RENDER_RGB", "1") != "0"      # A/B switch
        self._rgb_ring = None
        self._zface_ring = None
        self._mesh = None
        self.l_step = self.h_step = 3
        # ---- lattice (mu:2295-2320); flat order = cartesian product order (i, j, k, e, a)
        self.dims = (self.pose_l, self.pose_w, self.pose_h, self.pose_n_elev, self.pose_n_azim)
        # ---- state
        self.cam_idx = None
        self.X_cam = None
        self.V_cam = None
        self.R_cam = None
        self.T_cam = None
        self.cam_idx_history = []                       # list of 5-int tuples
        self._X_hist, self._V_hist = [], []             # rows; X_cam_history / V_cam_history stack them on demand
        self._hist_cache = None
        self._pose_cache = {}
        self.visited = set()
        self.n_frames_captured = 0
        self.frames = []                                # last frames: (zbuf [H,W] device view, cam12 host row)
        self._zbuf_ring = None
        self._cursor = 0
        self._traj_dev = torch.zeros(512, 3, dtype=torch.float32, device=device)
        self._traj_n = 0
        self._overflow = torch.zeros(1, dtype=torch.int32, device=device)

    # ------------------------------------------------------------------ lattice
    @property
    def X_cam_history(self):
        if self._hist_cache is None or self._hist_cache[0] != len(self._X_hist):
            self._hist_cache = (len(self._X_hist), np.asarray(self._X_hist, f32).reshape(-1, 3),
                                np.asarray(self._V_hist, f32).reshape(-1, 2))
        return self._hist_cache[1]

    @property
    def V_cam_history(self):
        _ = self.X_cam_history
        return self._hist_cache[2]

    def pose_from_idx(self, idx):
        """5-D pose (x, y, z, elev, azim) of lattice index (i, j, k, e, a) -- fp32 like mu:2315-2320."""
        idx = tuple(int(v) for v in idx)
        hit = self._pose_cache.get(idx)
        if hit is None:
            hit = self._pose_from_idx(idx)
            self._pose_cache[idx] = hit
        return hit.copy()

    def _pose_from_idx(self, idx):
        i, j, k, e, a = idx
        x_min = self.x_min_arg
        return np.array([x_min[0] + f32(i * self.l_step), x_min[1] + f32(3.3), x_min[2] + f32(k * self.h_step),
                         f32(-90.0) + (f32(180.0) * f32(1 + e)) / f32(self.pose_n_elev + 1),
                         (f32(360.0) * f32(a)) / f32(self.pose_n_azim)], f32)

    def get_pose_from_idx(self, idx):
        idx = tuple(int(v) for v in idx)
        return self.pose_from_idx(idx), (idx in self.visited)

    def in_lattice(self, ijk):
        return 0 <= ijk[0] < self.pose_l and 0 <= ijk[1] < self.pose_w and 0 <= ijk[2] < self.pose_h

    def positions(self, elev_index=2):
        """Position lattice used by the planner (scene.py:465 keeps elevation index 2 only;
        long_term_utils.py:420-433 collapses to positions): (idx3 [P,3] int, xyz [P,3] fp32), i-major."""
        idx = [(i, j, k) for i in range(self.pose_l) for j in range(self.pose_w) for k in range(self.pose_h)]
        xyz = np.stack([self.pose_from_idx((i, j, k, elev_index, 0))[:3] for i, j, k in idx])
        return np.asarray(idx, np.int64), xyz.astype(f32)

    def get_neighboring_poses(self, pose_idx):
        """mu:2473-2498: +-1 along x or z (clamped), same elevation, azimuth shifted by -3..3; sorted unique."""
        i, j, k, e, a = (int(v) for v in pose_idx)
        out = set()
        for di, dk in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            ni = min(max(i + di, 0), self.pose_l - 1)
            nk = min(max(k + dk, 0), self.pose_h - 1)
            if (ni, nk) == (i, k):
                continue
            for da in range(-3, 4):
                out.add((ni, j, nk, e, (a + da) % self.pose_n_azim))
        return sorted(out)

    def get_neighboring_poses_2d(self, pose_idx=None):
        """mu:2447-2471: same arithmetic as get_neighboring_poses (the y and elevation shifts are undone)."""
        return self.get_neighboring_poses(self.cam_idx if pose_idx is None else pose_idx)

    def cam12_of_pose(self, pose):
        R, T = camera_RT(pose[:3], pose[3:])
        return np.concatenate([R.reshape(-1), T.reshape(-1)]).astype(f32)

    def get_valid_neighbors(self, neighbor_indices, mesh):
        """mu:2528-2556: the unvisited neighbours whose field of view contains a mesh vertex (is_fov_empty, :2672-2688,
        range 5 zfar) -- all of them tested in ONE launch -- or, when there is none, the visited ones."""
        new = [n for n in neighbor_indices if tuple(n) not in self.visited]
        visited = [n for n in neighbor_indices if tuple(n) in self.visited]
        if new:
            cams = np.stack([self.cam12_of_pose(self.pose_from_idx(n)) for n in new])
            _, any_ = hipops.points_in_fov(mesh.verts, cams, self.image_height, self.image_width, 5 * self.zfar,
                                           want_mask=False)
            new = [n for n, a in zip(new, any_.cpu().tolist()) if a]
        return new if new else visited

    def get_points_in_fov(self, pts, return_mask=False, cam12=None, fov_range=None):
        """mu:2849-2884 for the current camera (or a [12] host camera): the points inside the field of view."""
        cam12 = self.cam12_of_pose(np.concatenate([self.X_cam, self.V_cam])) if cam12 is None else cam12
        rng = 1.1 * self.zfar if fov_range is None else fov_range
        mask, _ = hipops.points_in_fov(pts.contiguous(), cam12[None], self.image_height, self.image_width,
                                       1e30 if fov_range is None else rng)
        m = mask[0].bool()
        return (pts[m], m) if return_mask else pts[m]

    # ------------------------------------------------------------------ motion
    def _interp(self, new_idx, step):
        """Pose at interpolation step `step` of the move cam_idx -> new_idx (mu:2590-2632), fp32."""
        n = self.n_interpolation_steps
        old_pose = self.pose_from_idx(self.cam_idx if step < n else new_idx)
        new_pose = self.pose_from_idx(new_idx)
        if step == n:
            off = f32(0.0)
        elif self.cam_idx[4] == 0 and new_idx[4] == self.pose_n_azim - 1:
            off = f32(-360.0)
        elif self.cam_idx[4] == self.pose_n_azim - 1 and new_idx[4] == 0:
            off = f32(360.0)
        else:
            off = f32(0.0)
        X = old_pose[:3] + (new_pose[:3] - old_pose[:3]) * f32(step) / f32(n)
        V = old_pose[3:] + (new_pose[3:] - old_pose[3:]) * f32(step) / f32(n)
        V[1] = V[1] + off * f32(step) / f32(n)
        return X.astype(f32), V.astype(f32)

    def update_camera(self, new_cam_index, interpolation_step=None):
        new_idx = tuple(int(v) for v in new_cam_index)
        n = self.n_interpolation_steps
        step = n if interpolation_step is None else int(interpolation_step)
        if step > n:
            raise ValueError("interpolation_step is too large")
        if self.cam_idx is None:
            self.cam_idx = new_idx
        X, V = self._interp(new_idx, step)
        if step == n:
            self.cam_idx = new_idx
            self.cam_idx_history.append(new_idx)
            self.visited.add(new_idx)
        self.X_cam, self.V_cam = X, V
        self._X_hist.append(X)
        self._V_hist.append(V)
        self.R_cam, self.T_cam = camera_RT(X, V)

    def initialize_camera(self, start_cam_idx):
        self.update_camera(start_cam_idx)

    # ------------------------------------------------------------------ rendering (device resident frames)
    def _ring(self):
        if self._zbuf_ring is None:
            self._zbuf_ring = torch.empty(16, self.image_height, self.image_width, dtype=torch.float32,
                                          device=self.device)
        return self._zbuf_ring

    def _reserve(self, n):
        ring = self._ring()
        if self._cursor + n > 16:                       # explicit cursor: the 8 newest frames stay intact
            self._cursor = 0
        slot = self._cursor
        self._cursor += n
        return ring[slot:slot + n], slot

    def _commit(self, out, cams_host, slot):
        n = len(cams_host)
        for i in range(n):
            self.frames.append((out[i], cams_host[i].copy(), slot + i))
        self.frames = self.frames[-8:]
        self.n_frames_captured += n

    def deferred_colours(self, mesh):
        """True when this camera renders depth + nearest face and evaluates colours where they are consumed (the default for a
        mesh with vertex colours and contrast factor 1)."""
        return self.render_rgb and getattr(mesh, "colors", None) is not None and self.contrast_factor == 1.0

    def capture_begin(self, mesh, cams_host):
        """First half of capture_images for a batched render of several cameras (hipops.raster_zface_batch): reserves the ring
        slots and returns (out_z, out_zface, slot); capture_commit registers the frames once the launch is enqueued."""
        assert self.deferred_colours(mesh)
        out, slot = self._reserve(len(cams_host))
        self._mesh = mesh
        if self._zface_ring is None:
            self._zface_ring = torch.empty(16, self.image_height, self.image_width, dtype=torch.int64, device=self.device)
        return out, self._zface_ring[slot:slot + len(cams_host)], slot

    def capture_commit(self, out, cams_host, slot):
        self._commit(out, cams_host, slot)

    def capture_images(self, mesh, cams_host):
        """Rasterises len(cams_host) frames in one launch; cams_host [n,12] fp32 (R row-major, T)."""
        n = len(cams_host)
        out, slot = self._reserve(n)
        self._mesh = mesh
        if self.deferred_colours(mesh):
            # depth AND the nearest face per pixel: the colours of the reference's renderer (mu:2743-2763) are a pure
            # function of (face, pixel, camera, mesh), so they are evaluated where they are consumed -- for the ~5 % of
            # pixels the un-projection keeps (colour_source), or as whole images on request (frames_rgb)
            if self._zface_ring is None:
                self._zface_ring = torch.empty(16, self.image_height, self.image_width, dtype=torch.int64, device=self.device)
            hipops.raster_zface(mesh.verts, mesh.faces, cams_host, self.image_height, self.image_width, out_z=out,
                                out_zface=self._zface_ring[slot:slot + n])
        elif self.render_rgb and getattr(mesh, "colors", None) is not None:
            # a contrast change needs every pixel's luminance: eager colour render, colours in a ring beside the depths
            if self._rgb_ring is None:
                self._rgb_ring = torch.empty(16, self.image_height, self.image_width, 3, dtype=torch.float32, device=self.device)
            hipops.raster_rgbz(mesh.verts, mesh.faces, mesh.colors, cams_host, self.image_height, self.image_width,
                               self.ambient, self.contrast_factor, out_z=out, out_rgb=self._rgb_ring[slot:slot + n])
        else:
            hipops.raster_zbuf(mesh.verts, mesh.faces, cams_host, self.image_height, self.image_width, bin_cap=mesh.bin_cap,
                               out=out, overflow=self._overflow)
        self._commit(out, cams_host, slot)
        return out

    def capture_image(self, mesh):
        cam = np.concatenate([self.R_cam.reshape(-1), self.T_cam.reshape(-1)]).astype(f32)
        return self.capture_images(mesh, cam[None])

    def move_poses(self, next_idx):
        """The 4 interpolated camera updates of a move (nbp_planning.py:269-274) -> cams host [4,12] to render."""
        cams = []
        for step in range(1, self.n_interpolation_steps + 1):
            self.update_camera(next_idx, interpolation_step=step)
            cams.append(np.concatenate([self.R_cam.reshape(-1), self.T_cam.reshape(-1)]))
        return np.asarray(cams, f32)

    def move_and_capture(self, mesh, next_idx):
        """The 4 interpolated updates + captures of nbp_planning.py:269-274 with ONE raster launch."""
        return self.capture_images(mesh, self.move_poses(next_idx))

    def frames_batch(self, which):
        """Stacks frames by negative offsets (e.g. [-1] = current, [-5,-4,-3,-2] = supervision batch):
        (depth [n,H,W] device, cams [n,12] HOST -- cameras travel as kernel arguments)."""
        sel = [self.frames[w] for w in which]
        slots = [s[2] for s in sel]
        if all(b == a + 1 for a, b in zip(slots, slots[1:])):
            z = self._zbuf_ring[slots[0]:slots[0] + len(slots)]          # consecutive ring slots: a view, no copy kernel
        else:
            z = torch.stack([s[0] for s in sel])
        return z, np.stack([s[1] for s in sel]).astype(f32)

    def _ring_view(self, ring, which):
        slots = [self.frames[w][2] for w in which]
        if all(b == a + 1 for a, b in zip(slots, slots[1:])):
            return ring[slots[0]:slots[0] + len(slots)]
        return torch.stack([ring[k] for k in slots])

    def frames_rgb(self, which):
        """Colour images of the same frames [n,H,W,3] (None when the camera renders depth only)."""
        if self._zface_ring is not None:
            m = self._mesh
            cams = np.stack([self.frames[w][1] for w in which]).astype(f32)
            return hipops.shade_image(self._ring_view(self._zface_ring, which), m.verts, m.faces, m.colors, cams, self.ambient)
        if self._rgb_ring is None:
            return None
        return self._ring_view(self._rgb_ring, which)

    def colour_source(self, which):
        """Keyword arguments that make hipops.unproject_append carry the colours of these frames: {} when the camera renders
        depth only, shade=(nearest faces, mesh, ambient) in the deferred form, rgb=images after an eager colour render."""
        if self._zface_ring is not None:
            m = self._mesh
            return {"shade": (self._ring_view(self._zface_ring, which), m.verts, m.faces, m.colors, self.ambient)}
        if self._rgb_ring is not None:
            return {"rgb": self._ring_view(self._rgb_ring, which)}
        return {}

    @property
    def renders_colours(self):
        return self._zface_ring is not None or self._rgb_ring is not None

    def trajectory_pending(self):
        """(device history buffer, number of valid points in it, host array of the <= 8 positions not on the device yet) for
        utils.step_maps, whose kernel appends them; a longer backlog is flushed here first."""
        n = len(self._X_hist)
        if n > self._traj_dev.shape[0]:
            grown = torch.zeros(2 * n, 3, dtype=torch.float32, device=self.device)
            grown[:self._traj_n] = self._traj_dev[:self._traj_n]
            self._traj_dev = grown
        while n - self._traj_n > 8:
            hipops.append_points(self._traj_dev, self._traj_n, np.asarray(self._X_hist[self._traj_n:self._traj_n + 8], f32))
            self._traj_n += 8
        n_old = self._traj_n
        fresh = np.asarray(self._X_hist[n_old:n], f32).reshape(-1, 3)
        self._traj_n = n
        return self._traj_dev, n_old, fresh

    def trajectory_points(self):
        """X_cam_history on the device (for the trajectory channel); new poses are appended by a kernel whose
        arguments carry the points, so there is no blocking host->device copy in the step loop."""
        n = len(self._X_hist)
        if n > self._traj_dev.shape[0]:
            grown = torch.zeros(2 * n, 3, dtype=torch.float32, device=self.device)
            grown[:self._traj_n] = self._traj_dev[:self._traj_n]
            self._traj_dev = grown
        while self._traj_n < n:
            k = min(8, n - self._traj_n)
            hipops.append_points(self._traj_dev, self._traj_n, np.asarray(self._X_hist[self._traj_n:self._traj_n + k], f32))
            self._traj_n += k
        return self._traj_dev[:n]
