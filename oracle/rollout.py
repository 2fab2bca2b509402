"""CPU exploration rollout composed from the oracle pieces (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, step by step, compute_nbp_trajectory (next_best_path/testers/nbp_planning.py:23-361), setup_test_camera
(macarons/testers/scene.py:410-488), Camera.update_camera / get_neighboring_poses
(macarons/utility/macarons_utils.py:2590-2632, 2473-2498) and generate_Dijkstra_path
(next_best_path/utility/long_term_utils.py:334-418) on numpy arrays: raster (csim.raster_zbuf == raster.py) ->
un-projection + sub-sampling (camera.partial_point_cloud) -> map accumulation (maps.py) -> NBP forward
(nbp_net.py, stock torch CPU ops) -> fusion / scoring / Bresenham edge test / heapq search (planner.py) -> move.

It is written against the REFERENCE's control flow (string-free: lattice positions are (i,j,k) tuples), not against
nextbestpath_amd/testers/nbp_planning.py; what it shares with the product are the documented deviations only
(DESIGN.md section 7): every random draw is seeded -- sub-sampling by the index bijection of sampling.py with
seed = step_seed + 11*pose_i (+5 for the supervision frames), coverage sub-sampling with step_seed + 7*pose_i,
headings from random.Random(seed) -- and "no path" turns in place instead of crashing on the unbound next_idx.
tests/test_gpu_rollout_parity.py steps this beside the HIP rollout and compares poses, cloud sizes and coverage."""
import heapq
import random

import numpy as np
import torch

from . import camera as ocam
from . import csim
from . import maps as omaps
from . import mesh_rays
from . import nbp_net
from . import planner as opl
from . import sampling

f32 = np.float32


class OracleCamera:
    """Pose lattice (mu:2283-2327), interpolated motion (mu:2590-2632), frame list (zbuf, R, T)."""

    def __init__(self, x_min, pose_l, pose_w, pose_h, n_elev, n_azim, n_interp, H, W, colors=None, ambient=0.85, contrast=1.0):
        self.colors, self.ambient, self.contrast = colors, ambient, contrast
        self.x_min = np.asarray(x_min, f32)
        self.dims = (int(pose_l), int(pose_w), int(pose_h), int(n_elev), int(n_azim))
        self.n_interp, self.H, self.W = int(n_interp), int(H), int(W)
        self.cam_idx = None
        self.cam_idx_history = []
        self.X_hist, self.V_hist = [], []
        self.frames = []

    def pose(self, idx):
        i, j, k, e, a = (int(v) for v in idx)
        _, _, _, n_e, n_a = self.dims
        return np.array([self.x_min[0] + f32(3 * i), self.x_min[1] + f32(3.3), self.x_min[2] + f32(3 * k),
                         f32(-90.0) + (f32(180.0) * f32(1 + e)) / f32(n_e + 1), (f32(360.0) * f32(a)) / f32(n_a)], f32)

    def neighbors(self, idx):
        """get_neighboring_poses: +-1 in x or z (clamped), azimuth shift -3..3, unique rows in sorted order."""
        i, j, k, e, a = (int(v) for v in idx)
        L, _, Hh, _, A = self.dims
        out = set()
        for di, dk in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            ni, nk = min(max(i + di, 0), L - 1), min(max(k + dk, 0), Hh - 1)
            if abs(ni - i) + abs(nk - k) == 0:
                continue
            for da in range(-3, 4):
                out.add((ni, j, nk, e, (a + da) % A))
        return sorted(out)

    def update(self, new_idx, step=None):
        new_idx = tuple(int(v) for v in new_idx)
        n = self.n_interp
        step = n if step is None else step
        if self.cam_idx is None:
            self.cam_idx = new_idx
        if step == n:
            self.cam_idx = new_idx
            self.cam_idx_history.append(new_idx)
        old, new = self.pose(self.cam_idx), self.pose(new_idx)
        A = self.dims[4]
        if step == n:
            off = f32(0)
        elif self.cam_idx[4] == 0 and new_idx[4] == A - 1:
            off = f32(-360)
        elif self.cam_idx[4] == A - 1 and new_idx[4] == 0:
            off = f32(360)
        else:
            off = f32(0)
        X = old[:3] + (new[:3] - old[:3]) * f32(step) / f32(n)
        V = old[3:] + (new[3:] - old[3:]) * f32(step) / f32(n)
        V[1] = V[1] + off * f32(step) / f32(n)
        self.X, self.V = X.astype(f32), V.astype(f32)
        self.X_hist.append(self.X.copy())
        self.V_hist.append(self.V.copy())
        self.R, self.T = ocam.camera_RT(self.X, self.V)

    def capture(self, verts, faces):
        if self.colors is not None:
            z, rgb = csim.raster_rgbz(verts, faces, self.colors, self.R, self.T, self.H, self.W, ocam.TAN_HALF_FOV, self.ambient,
                                      self.contrast)
            self.frames.append((z, self.R.copy(), self.T.copy(), rgb))
        else:
            z = csim.raster_zbuf(verts, faces, self.R, self.T, self.H, self.W, ocam.TAN_HALF_FOV)
            self.frames.append((z, self.R.copy(), self.T.copy()))
        self.frames = self.frames[-8:]

    def move_and_capture(self, verts, faces, next_idx):
        for s in range(1, self.n_interp + 1):
            self.update(next_idx, s)
            self.capture(verts, faces)


class OracleRollout:
    def __init__(self, sd, verts, faces, gt, y_bins, cam_x_min, dims, start_idx, seed, S=256, n_interp=4, H=256, W=456,
                 gathering_factor=0.05, sensor_range=70.0, colors=None, ambient=0.85, contrast=1.0):
        self.sd, self.verts, self.faces = sd, np.asarray(verts, f32), np.asarray(faces, np.int32)
        self.gt, self.y_bins = np.asarray(gt, f32), np.asarray(y_bins, f32)
        self.S, self.V, self.grid_range = S, S // 4, (-40 * S // 256, 40 * S // 256)
        self.gf, self.sensor_range = gathering_factor, sensor_range
        self.rng = random.Random(seed)
        self.step_seed = seed * 1_000_003
        cam = self.cam = OracleCamera(cam_x_min, dims[0], dims[1], dims[2], dims[3], dims[4], n_interp, H, W, colors, ambient,
                                      contrast)
        self.full_rgb = np.zeros((0, 3), f32)
        # ---- setup_test_camera (scene.py:465-488)
        start = tuple(int(v) for v in start_idx)
        first = None
        for nb in cam.neighbors(start):
            if not mesh_rays.segment_hits_mesh(cam.pose(nb)[:3], cam.pose(start)[:3], self.verts, self.faces):
                first = nb
                break
        if first is None:
            first = cam.neighbors(start)[0]
        cam.update(first)
        cam.capture(self.verts, self.faces)
        cam.move_and_capture(self.verts, self.faces, start)
        # ---- lattice positions at elevation index 2 (scene.py:465, long_term_utils.py:420-433), i-major
        self.nodes = [(i, j, k) for i in range(dims[0]) for j in range(dims[1]) for k in range(dims[2])]
        self.node_set = set(self.nodes)
        self.xyz = {n: cam.pose((n[0], n[1], n[2], 2, 0))[:3] for n in self.nodes}
        self.full_pc = np.zeros((0, 3), f32)
        self.path, self.path_record = [], 0
        self.collision_list, self.passable_list, self.idx_history = [], [], []
        self.coverage_counts, self.cloud_sizes = [], []
        self.pose_i = 0
        self.n_replans = 0
        self.net_inputs, self.net_outputs = [], []

    # ------------------------------------------------------------------ pieces
    def _append_frames(self, which, seed):
        for fi, w in enumerate(which):
            fr = self.cam.frames[w]
            z, R, T = fr[0], fr[1], fr[2]
            if len(fr) > 3:            # colours of the kept pixels ride along (full_pc_colors, nbp_planning.py:106,353)
                pts, _, col = ocam.partial_point_cloud(z, None, R, T, self.gf, self.sensor_range, seed, frame_index=fi, rgb=fr[3])
                self.full_rgb = np.concatenate([self.full_rgb, col], 0)
            else:
                pts, _ = ocam.partial_point_cloud(z, None, R, T, self.gf, self.sensor_range, seed, frame_index=fi)
            self.full_pc = np.concatenate([self.full_pc, pts], 0)

    def _segment_hits(self, a3, b3):
        if tuple(a3) == tuple(b3):
            return False
        return mesh_rays.segment_hits_mesh(self.xyz[tuple(a3)], self.xyz[tuple(b3)], self.verts, self.faces)

    def _dijkstra(self, start, goal, pose, obst, out1, blocked_cache):
        """generate_Dijkstra_path with training_flag=False (ltu:334-418).  Returns [[i,j,k,2,h], ...] without the
        start node, or None."""
        def edge_ok(a, b):
            if [list(a), list(b)] in self.passable_list:
                return True
            key = (a, b)
            if key not in blocked_cache:
                blocked_cache[key] = opl.edge_blocked(self.xyz[a], self.xyz[b], pose, obst, self.grid_range)
            return (not blocked_cache[key]) and [list(a), list(b)] not in self.collision_list

        frontier = [(0, start)]
        came_from, cost = {start: None}, {start: 0}
        while frontier:
            _, cur = heapq.heappop(frontier)
            if cur == goal:
                break
            x, y, z = cur
            for nb in ((x + 1, y, z), (x - 1, y, z), (x, y, z + 1), (x, y, z - 1)):
                if nb in self.node_set and edge_ok(cur, nb):
                    nc = cost[cur] + 1
                    if nb not in cost or nc < cost[nb]:
                        cost[nb] = nc
                        heapq.heappush(frontier, (nc, nb))
                        came_from[nb] = cur
        if goal not in came_from:
            return None
        nodes, cur = [], goal
        while cur:
            nodes.append(cur)
            cur = came_from[cur]
        nodes.reverse()
        hist = set(self.cam.cam_idx_history)
        out = []
        for n in nodes:
            g = opl._cell(self.xyz[n], pose, self.V, self.grid_range)
            if 0 <= g[0] < self.V and 0 <= g[1] < self.V:
                order = np.argsort(-out1[:, int(g[0]), int(g[1])], kind="stable")
                h = int(order[-1])
                for c in order.tolist():
                    h = int(c)
                    if (n[0], n[1], n[2], 2, h) not in hist:
                        break
            else:
                for _ in range(64):
                    h = self.rng.randint(0, 7)
                    if (n[0], n[1], n[2], 2, h) not in hist:
                        break
            out.append([n[0], n[1], n[2], 2, h])
        return out[1:]

    # ------------------------------------------------------------------ one step (nbp_planning.py:60-355)
    def step(self):
        cam, pose_i, S, gr = self.cam, self.pose_i, self.S, self.grid_range
        # coverage of the cloud so far (:71-74)
        if len(self.full_pc) == 0:
            cnt = 0
        else:
            pc = self.full_pc
            k = int(len(self.gt) * 2)
            if len(pc) > k:
                pc = pc[sampling.perm_index(np.arange(k), len(pc), (self.step_seed + 7 * pose_i) & sampling.M32)]
            cnt = csim.coverage_count(self.gt, pc, 1.0)
        self.coverage_counts.append(cnt)
        # current frame -> cloud (:96-109)
        self._append_frames([-1], (self.step_seed + 11 * pose_i) & sampling.M32)
        self.cloud_sizes.append(len(self.full_pc))
        pose = cam.pose(cam.cam_idx)
        # maps (:114-131)
        maps6 = omaps.accumulate_step_maps(self.full_pc, pose, self.y_bins, S, gr)
        traj2d = omaps.transform_points_to_n_pieces(np.stack(cam.X_hist), pose)
        traj_img = omaps.map_points_to_n_imgs(traj2d, (S, S), gr)[0]
        net_in = np.concatenate([maps6[:4], traj_img[None]], 0)[None]
        # replan condition (:133-155)
        path = self.path
        if pose_i == 0 or path is None or self.path_record + 1 > len(path):
            dij = True
        else:
            nxt = path[self.path_record]
            dij = self._segment_hits(cam.cam_idx[:3], nxt[:3])
            if dij:
                c3, n3 = list(cam.cam_idx[:3]), list(nxt[:3])
                self.collision_list += [[c3, n3], [n3, c3], list(path[-1][:3])]
        if len(self.idx_history) >= 2:
            p1, p2 = list(self.idx_history[-1][:3]), list(self.idx_history[-2][:3])
            self.passable_list += [[p1, p2], [p2, p1]]
        # one NBP forward per step (:166 or :252)
        with torch.no_grad():
            o1, o2 = nbp_net.nbp_forward(self.sd, torch.from_numpy(net_in))
        out1, out2 = o1[0].numpy(), o2[0, 0].numpy()
        self.net_inputs.append(net_in)
        self.net_outputs.append((out1, out2))
        if dij:
            self.n_replans += 1
            self.path_record = 0
            path = []
            obst, fullproj = opl.fuse_obstacle(out2, maps6, traj_img)
            positions = np.stack([self.xyz[n] for n in self.nodes])
            skip = np.array([list(n) in self.collision_list for n in self.nodes])
            valid, _, scores = opl.score_candidates(positions, pose, out1, fullproj, skip, self.V, gr)
            cand = [(self.nodes[i], scores[i]) for i in np.nonzero(valid)[0]]
            cand.sort(key=lambda t: t[1], reverse=True)                    # stable (:233)
            self.last_candidates = cand
            self.last_goal = None                                          # (introspection for the parity tests)
            start = tuple(cam.cam_idx[:3])
            cache, cache_version = {}, len(self.collision_list)
            for goal, _ in cand:
                path = self._dijkstra(start, goal, pose, obst, out1, cache)
                if path is not None and len(path) > 0:
                    if not self._segment_hits(cam.cam_idx[:3], path[0][:3]):
                        self.last_goal = goal
                        break
                    c3, n3 = list(cam.cam_idx[:3]), list(path[0][:3])
                    self.collision_list += [[c3, n3], [n3, c3]]
        # next pose (:254-265); "no path" turns in place (documented deviation)
        if not path or self.path_record >= len(path):
            next_idx = list(cam.cam_idx)
            next_idx[4] = self.rng.randrange(8)
            path = []
        else:
            next_idx = list(path[self.path_record])
            if tuple(next_idx) in set(self.idx_history):
                next_idx[4] = self.rng.randrange(8)
        self.path = path
        self.idx_history.append(tuple(cam.cam_idx))
        cam.move_and_capture(self.verts, self.faces, next_idx)
        self._append_frames([-5, -4, -3, -2], (self.step_seed + 11 * pose_i + 5) & sampling.M32)
        self.path_record += 1
        self.pose_i += 1
