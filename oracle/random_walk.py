"""CPU restatement of the random-walk baseline rollout (TEST INFRASTRUCTURE): compute_random_walk_trajectory
(macarons/testers/random_walk_planning.py:25-400) restricted to what runs without the unreleased MACARONS / SCONE
networks -- covered / surface scene fills, per-cell scene coverage, depth-map carving of the proxy points, valid
neighbours, uniformly random moves -- composed from oracle/scene_store.py, oracle/camera.py and csim.raster_zbuf, with
the product's documented seeds.  tests/test_gpu_random_walk.py steps it beside the HIP rollout."""
import random

import numpy as np

from . import camera as ocam
from . import sampling
from . import scene_store as oss
from . import view_state as ovs
from .rollout import OracleCamera

f32 = np.float32


class OracleRandomWalk:
    def __init__(self, verts, faces, cam_x_min, dims, start_idx, first_idx, scene_box, grid, params, proxy_points, gt_pts,
                 seed, scene_seeds, H=256, W=456):
        self.verts, self.faces = np.asarray(verts, f32), np.asarray(faces, np.int32)
        self.p = params
        self.cam = OracleCamera(cam_x_min, dims[0], dims[1], dims[2], dims[3], dims[4], params["n_interp"], H, W)
        self.cam.update(first_idx)
        self.cam.capture(self.verts, self.faces)
        self.cam.move_and_capture(self.verts, self.faces, start_idx)
        self.visited = set(self.cam.cam_idx_history)
        x_min, x_max = scene_box
        cap, res = params["surface_cell_capacity"], params["test_resolution"] * params["scale"]
        self.gt = oss.Scene(x_min, x_max, *grid, cap, res)
        self.gt.fill_cells(gt_pts, seed=scene_seeds[0] + 7919)
        self.covered = oss.Scene(x_min, x_max, *grid, cap, res)
        self.surface = oss.Scene(x_min, x_max, *grid, cap, None)
        self.seeds = {"covered": scene_seeds[1], "surface": scene_seeds[2]}
        self.fills = {"covered": 0, "surface": 0}
        self.proxy = np.asarray(proxy_points, f32)
        P = len(self.proxy)
        self.n_inside, self.n_behind = np.zeros(P, f32), np.zeros(P, f32)
        self.occ, self.oof = np.ones(P, f32), np.ones(P, f32)
        self.n_elev, self.n_azim = params.get("view_state_n_elev", 7), params.get("view_state_n_azim", 14)
        self.view_states = np.zeros((P, self.n_elev * self.n_azim), f32)
        # Scene.__init__ (mu:3110-3124): proxy_radius from the cell volume per proxy point
        n_cells = grid[0] * grid[1] * grid[2]
        d = (np.asarray(x_max, f32) - np.asarray(x_min, f32)) / np.asarray(grid, f32)
        vol = float(d[0] * d[1] * d[2]) / max(P / n_cells, 1e-30)
        self.dist_between = 2 * np.power(3 * vol / (4 * np.pi), 1.0 / 3.0)
        self.full_pc = np.zeros((0, 3), f32)
        self.rng = random.Random(seed)
        self.seed = seed * 1_000_003
        self.eps = 2 * params["test_resolution"] * params["scale"]
        self.coverage_evolution = []
        self.pose_i = 0

    def _fill(self, name, scene, pts, n_point_min=0):
        self.fills[name] += 1                    # the product counts every call (its scratch cloud is never zero-sized)
        if len(pts) == 0:
            return
        scene.fill_cells(pts, n_point_min, seed=self.seeds[name] + 7919 * self.fills[name])

    def _partial(self, which, seed):
        out = []
        for fi, w in enumerate(which):
            z, R, T = self.cam.frames[w][:3]
            pts, _ = ocam.partial_point_cloud(z, None, R, T, self.p["gathering_factor"], self.p["sensor_range"], seed & sampling.M32,
                                              frame_index=fi)
            out.append(pts)
        return np.concatenate(out, 0)

    def _carve(self, w, X_cam=None):
        """carving + update_proxy_view_states (random_walk_planning.py:140-166 with X_cam = the camera's pose position,
        :375-385 with the frame's camera centre)"""
        z, R, T = self.cam.frames[w][:3]
        inf, sd = ocam.carve_update(self.proxy, z, None, R, T, self.p["zfar"], self.p["sensor_range"], self.p["carving_tolerance"],
                                    self.p["score_threshold"], self.n_inside, self.n_behind, self.occ, self.oof)
        xc = ocam.camera_center(R, T) if X_cam is None else np.asarray(X_cam, f32)
        ovs.update_proxy_view_states(self.view_states, self.proxy, inf, sd, xc, self.n_elev, self.n_azim, 3 * self.dist_between)

    def _valid_neighbors(self):
        nbrs = self.cam.neighbors(self.cam.cam_idx)
        new = [n for n in nbrs if n not in self.visited]
        old = [n for n in nbrs if n in self.visited]
        ok = []
        for n in new:
            pose = self.cam.pose(n)
            R, T = ocam.camera_RT(pose[:3], pose[3:])
            if ocam.points_in_fov(self.verts, R, T, self.cam.H, self.cam.W, 5 * self.p["zfar"]).any():
                ok.append(n)
        return ok if ok else old

    def step(self):
        p, pose_i = self.p, self.pose_i
        if pose_i > 0 and pose_i % p["recompute_every"] == 0:
            N, k = len(self.full_pc), p["n_gt_surface_points"]
            m = min(N, k)
            sample = self.full_pc[sampling.perm_index(np.arange(m), N, (self.seed + 13 * pose_i) & sampling.M32)]
            for c in self.surface.cells.values():
                c.pts = np.zeros((0, 3), f32)
            step = p["max_points_per_fill"]
            n_fill = k // step + (1 if k % step else 0)
            for q in range(n_fill):
                lo = q * step
                chunk = sample[lo:-1] if q == k // step else sample[lo:lo + step]
                self._fill("surface", self.surface, chunk, 3)
        part = self._partial([-1], self.seed + 11 * pose_i)
        self._fill("covered", self.covered, part)
        covered, n_gt = oss.scene_coverage(self.gt, self.covered, self.eps)
        self.coverage_evolution.append(covered / n_gt if n_gt else 0.0)
        part = self._partial([-1], self.seed + 11 * pose_i + 3)
        self._fill("surface", self.surface, part)
        self.full_pc = np.concatenate([self.full_pc, part], 0)
        self._carve(-1, X_cam=self.cam.X)
        valid = self._valid_neighbors()
        next_idx = self.rng.choice(valid)
        self.cam.move_and_capture(self.verts, self.faces, next_idx)
        self.visited.add(tuple(next_idx))
        part = self._partial([-5, -4, -3, -2], self.seed + 11 * pose_i + 5)
        self._fill("surface", self.surface, part)
        self.full_pc = np.concatenate([self.full_pc, part], 0)
        for w in (-5, -4, -3, -2):
            self._carve(w)
        self.pose_i += 1
