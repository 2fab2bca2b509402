"""Planner glue -- mirrors of next_best_path/utility/long_term_utils.py for the NBP path, same
function names and argument meaning; the per-pixel / per-edge Python loops of the reference are
replaced by one kernel launch each (csrc/nbp_planner.hip, csrc/nbp_sim.hip)."""
from __future__ import annotations

import ctypes as C
import os
import random

import numpy as np
import torch

from .. import _lib
from . import hipops, planner_host
from .utils import _pose_xyz


def bresenham_line(x0, y0, x1, y1):
    """ref :277-298 (pure integer Python; kept for API parity, the kernels inline the same loop)."""
    pts = []
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err = dx - dy
    while True:
        pts.append((x0, y0))
        if x0 == x1 and y0 == y1:
            return pts
        e2 = 2 * err
        if e2 > -dy:
            err -= dy
            x0 += sx
        if e2 < dx:
            err += dx
            y0 += sy


def line_across_image_pixel(point_1, point_2, camera_current_pose, grid_size, grid_range, layout_image, device=None):
    """ref :300-331 -- True iff the edge is blocked (endpoint off-image or >= 2 obstacle pixels)."""
    pos = torch.stack([point_1[:3], point_2[:3]]).float().contiguous()
    ed = torch.tensor([[0, 1]], dtype=torch.int32, device=pos.device)
    obst = layout_image.reshape(grid_size[0], grid_size[1]).float().contiguous()
    out = hipops.edges_blocked(obst, _pose_xyz(camera_current_pose), pos, ed, grid_range)
    return bool(out.item())


def line_segment_mesh_intersection(start_point, end_point, mesh):
    """macarons_utils.py:120-151 on a DeviceMesh (brute-force GPU ray/triangle, no rtree)."""
    a = torch.as_tensor(start_point, dtype=torch.float32).reshape(3)
    b = torch.as_tensor(end_point, dtype=torch.float32).reshape(3)
    seg = torch.cat([a, b]).reshape(1, 6).to(mesh.verts.device)
    return bool(hipops.segments_hit_mesh(mesh.verts, mesh.faces, seg).item())


def check_camera_in_mesh(mesh_for_check, camera_position):
    """ref :158-170: inside iff the +Y, +X, +Z ray hit counts are all odd."""
    p = torch.as_tensor(camera_position, dtype=torch.float32).reshape(1, 3).to(mesh_for_check.verts.device)
    c = hipops.axis_ray_counts(mesh_for_check.verts, mesh_for_check.faces, p)[0].tolist()
    return all(v % 2 == 1 for v in c)


def calculate_coverage_percentage(pc1, pc2, threshold=1, weight=2, seed=0):
    """ref :457-468 (returns a Python float; one device sync, like the reference's .item())."""
    if len(pc2) == 0:
        return 0.0
    out = hipops.coverage_count(pc1.contiguous(), pc2.contiguous(), weight=weight, seed=seed, threshold=threshold)
    cnt = int(out[0].item())
    return float(np.float32(cnt) / np.float32(len(pc1)))


def compute_auc(y, dx=1 / 40):
    """ref :488-490."""
    y = np.asarray(y, np.float64)
    trap = getattr(np, "trapezoid", None) or np.trapz
    return float(trap(y, dx=dx) + y[0] * dx / 2.0)


class LatticePlanner:
    """Batched replacement of the replanning block of compute_nbp_trajectory
    (next_best_path/testers/nbp_planning.py:158-249) + generate_Dijkstra_path (ref :334-418).

    Per replan: 3 kernel launches (obstacle fusion, candidate scoring, all-edges Bresenham) and one
    device->host copy, then the reference's host logic (stable sort, uniform-cost search with
    heapq tie-breaking, heading choice, first-edge check against the real mesh)."""

    def __init__(self, camera, mesh, device, value_size=64, layout_size=256, grid_range=(-40, 40), rng=None):
        self.camera, self.mesh, self.device = camera, mesh, device
        self.rng = rng or random.Random(0)
        self.V, self.S, self.grid_range = value_size, layout_size, grid_range
        self.idx3, self.xyz = camera.positions()
        self.node_index = {tuple(t): n for n, t in enumerate(self.idx3.tolist())}
        self.pos_dev = torch.from_numpy(self.xyz).to(device)
        edges = []
        for n, (i, j, k) in enumerate(self.idx3.tolist()):
            for nb in ((i + 1, j, k), (i - 1, j, k), (i, j, k + 1), (i, j, k - 1)):
                if nb in self.node_index:
                    edges.append((n, self.node_index[nb]))
        self.edges = edges
        self.edge_id = {e: q for q, e in enumerate(edges)}
        self.nbrs = [[] for _ in range(len(self.idx3))]          # per node: (neighbour id, edge id) in +x,-x,+z,-z order
        for q, (a, b) in enumerate(edges):
            self.nbrs[a].append((b, q))
        self.edges_dev = torch.tensor(edges, dtype=torch.int32, device=device)
        segs = np.concatenate([self.xyz[[a for a, _ in edges]], self.xyz[[b for _, b in edges]]], 1).astype(np.float32)
        self.mesh_hit = hipops.segments_hit_mesh(mesh.verts, mesh.faces, torch.from_numpy(segs).to(device)).cpu().numpy()
        # host search in C++ (nbp_plan_search_host): the graph as flat int32 arrays, the two edge lists of the rollout as
        # masks that follow the (append-only) Python lists incrementally
        P, E = len(self.idx3), len(edges)
        self.idx3_list = [tuple(t) for t in self.idx3.tolist()]
        self._idx3_i32 = np.ascontiguousarray(self.idx3, np.int32)
        self._xyz_f32 = np.ascontiguousarray(self.xyz, np.float32)
        self._edges_i32 = np.asarray(edges, np.int32).reshape(-1, 2)
        self._edge_first = np.zeros(P + 1, np.int32)
        np.cumsum(np.bincount(self._edges_i32[:, 0], minlength=P), out=self._edge_first[1:])
        self._mesh_hit_u8 = np.ascontiguousarray(self.mesh_hit, np.uint8)
        self._coll_mask, self._pass_mask = np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self._coll_src, self._coll_n, self._pass_src, self._pass_n = None, 0, None, 0
        self._skip_pin, self._skip_any = None, False
        self._hist = np.zeros((256, 5), np.int32)
        self._hist_n = 0
        self._path_nodes, self._path_heads = np.zeros(P + 1, np.int32), np.zeros(P + 1, np.int32)
        self._new_coll = np.zeros(2 * (P + 1), np.int32)        # at most one failed first edge per candidate
        self.native_search = _lib.tune("NBP_PLAN_SEARCH", "native") != "python"

    # ---- the GPU half of a replan for several rollouts at once (MultiRollout): persistent result buffers, ONE device->host copy
    def result_bytes(self):
        """Bytes of one rollout's replan results (scores f64 [P] | valid u8 [P] | blocked u8 [E]), rounded up to a multiple of 256:
        MultiRollout lays the rollouts of a group out as ROWS of one buffer, and every row must start 8-byte aligned for its
        float64 scores (E = 2 W (2 L H - L - H) is a multiple of 8 on square lattices only: pose_l = 10, pose_h = 12 gives 436)."""
        P, E = len(self.idx3), len(self.edges)
        return ((8 * P + P + 7) // 8 * 8 + E + 255) // 256 * 256

    def share_result_buffers(self, dev_row, pin_row):
        """MultiRollout: this planner's results live in a row of its GROUP's buffers, so that the group needs ONE device -> host
        copy per step instead of one per replanning rollout (each such copy is a launch: ~15 per group and step)."""
        assert dev_row.numel() == self.result_bytes() == pin_row.numel()
        self._res_dev, self._res_pin, self._res_shared = dev_row, pin_row, True
        self._res_views_dev = None

    def _batch_buffers(self):
        if getattr(self, "_res_dev", None) is None or getattr(self, "_res_views_dev", None) is None:
            P, E, dev = len(self.idx3), len(self.edges), self.device
            o_valid = 8 * P
            o_blocked = (o_valid + P + 7) // 8 * 8
            total = self.result_bytes()                  # (o_blocked + E, padded to the row pitch)
            if getattr(self, "_res_dev", None) is None:
                self._res_dev = torch.empty(total, dtype=torch.uint8, device=dev)
                self._res_pin = torch.empty(total, dtype=torch.uint8, pin_memory=True)
            cut = lambda t: (t[:8 * P].view(torch.float64), t[o_valid:o_valid + P], t[o_blocked:o_blocked + E])
            self._res_views_dev, self._res_views_pin = cut(self._res_dev), cut(self._res_pin)
            self._obst = torch.empty(self.S, self.S, dtype=torch.float32, device=dev)
            self._fullproj = torch.empty(self.S, self.S, dtype=torch.float32, device=dev)
            self._cell = torch.empty(P, 2, dtype=torch.int32, device=dev)
            self._skip_dev = torch.zeros(P, dtype=torch.uint8, device=dev)
        return self._res_views_dev

    def replan_item(self, pose, out1, out2, maps6, traj_img, collision_list):
        """Arguments of hipops.replan_batch for this rollout (the skip mask goes to the device here when it is in use)."""
        score, valid, blocked = self._batch_buffers()
        self._sync_collisions(collision_list)
        skip = None
        if self._skip_any:
            self._skip_dev.copy_(self._skip_pin, non_blocking=True)
            skip = self._skip_dev
        return (out2, maps6, traj_img, self._obst, self._fullproj, self.pos_dev, (float(pose[0]), float(pose[2])), out1, skip, valid,
                self._cell, score, self.edges_dev, blocked)

    def replan_copy_back(self, pose, out1_pinned):
        """After the batched launch: this rollout's (score, valid, blocked) in ONE copy; out1_pinned = its [8,V,V] slice of the
        group's pinned copy of the value maps."""
        if not getattr(self, "_res_shared", False):          # (shared rows: the group copies all of them at once)
            self._res_pin.copy_(self._res_dev, non_blocking=True)
        score, valid, blocked = self._res_views_pin
        self._pending = (pose, [valid, score, blocked, out1_pinned])

    def _staging(self, *tensors):
        if getattr(self, "_stg", None) is None:
            self._stg = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
        return self._stg

    def _edge_ok(self, blocked_h, collision_list, passable_list):
        """Vectorised edge predicate of generate_Dijkstra_path.get_neighbors (ref :350-360):
        ok = [a,b] in passable_list or (not blocked and [a,b] not in collision_list)."""
        ok = ~blocked_h
        for a, b in (e for e in collision_list if len(e) == 2):
            q = self.edge_id.get((self.node_index.get(tuple(a), -1), self.node_index.get(tuple(b), -1)))
            if q is not None:
                ok[q] = False
        for a, b in passable_list:
            q = self.edge_id.get((self.node_index.get(tuple(a), -1), self.node_index.get(tuple(b), -1)))
            if q is not None:
                ok[q] = True
        return ok

    def _edge_q(self, a, b):
        return self.edge_id.get((self.node_index.get(tuple(a), -1), self.node_index.get(tuple(b), -1)))

    def _sync_collisions(self, collision_list):
        """Brings the collision edge mask and the skipped-goal mask up to date with the rollout's list (entries of two
        positions are edges, entries of three numbers are goal positions that led into a collision: nbp_planning.py:147-149)."""
        if self._skip_pin is None:
            self._skip_pin = torch.zeros(len(self.idx3), dtype=torch.uint8).pin_memory()
            self._skip_np = self._skip_pin.numpy()
        if self._coll_src is not collision_list or self._coll_n > len(collision_list):
            self._coll_mask[:] = 0
            self._skip_np[:] = 0
            self._coll_src, self._coll_n, self._skip_any = collision_list, 0, False
        for e in collision_list[self._coll_n:]:
            if len(e) == 2:
                q = self._edge_q(e[0], e[1])
                if q is not None:
                    self._coll_mask[q] = 1
            elif len(e) == 3:
                n = self.node_index.get(tuple(e))
                self._skip_any = True              # the reference's set is non-empty even for a position off the lattice
                if n is not None:
                    self._skip_np[n] = 1
        self._coll_n = len(collision_list)

    def _sync_passable(self, passable_list):
        if self._pass_src is not passable_list or self._pass_n > len(passable_list):
            self._pass_mask[:] = 0
            self._pass_src, self._pass_n = passable_list, 0
        for a, b in passable_list[self._pass_n:]:
            q = self._edge_q(a, b)
            if q is not None:
                self._pass_mask[q] = 1
        self._pass_n = len(passable_list)

    def _history(self):
        """cam_idx_history as an int32 [n,5] array, converted incrementally (the list only grows)."""
        h = self.camera.cam_idx_history
        n = len(h)
        if n < self._hist_n:
            self._hist_n = 0
        if n > len(self._hist):
            grown = np.zeros((2 * n, 5), np.int32)
            grown[:self._hist_n] = self._hist[:self._hist_n]
            self._hist = grown
        if n > self._hist_n:
            self._hist[self._hist_n:n] = np.asarray(h[self._hist_n:n], np.int32).reshape(-1, 5)
            self._hist_n = n
        return self._hist, n

    # ---- replanning in two halves so that several rollouts can share ONE stream synchronisation
    def replan_enqueue(self, pose, out1, out2, maps6, traj_img, collision_list):
        """Launches obstacle fusion, candidate scoring and the all-edges mask, then starts the pinned
        device->host copies.  Nothing here blocks the host."""
        obst, fullproj = hipops.fuse_obstacle(out2.reshape(self.S, self.S), maps6, traj_img.reshape(self.S, self.S))
        self._sync_collisions(collision_list)
        # goals that led into a collision are not proposed again; the pinned mask is only written between a step's stream
        # synchronisation and the next enqueue, never while a copy is in flight
        skip = self._skip_pin.to(self.device, non_blocking=True) if self._skip_any else None
        o1 = out1.reshape(8, self.V, self.V)
        valid, cell, score = hipops.score_candidates(self.pos_dev, pose, o1, fullproj, skip, self.grid_range)
        blocked = hipops.edges_blocked(obst, pose, self.pos_dev, self.edges_dev, self.grid_range)
        stg = self._staging(valid, score, blocked, o1)
        for dst, src in zip(stg, (valid, score, blocked, o1)):
            dst.copy_(src, non_blocking=True)
        self._pending = (pose, stg)

    def replan_finish(self, collision_list, passable_list, check_first_edge=True):
        """Host half (after a stream synchronisation): stable sort, search, heading choice, first-edge test."""
        pose, stg = self._pending
        valid_h, score_h = stg[0].numpy(), stg[1].numpy()
        cand = np.nonzero(valid_h)[0]
        cand = cand[np.argsort(-score_h[cand], kind="stable")]               # stable, descending (ref :233)
        self.last_candidates, self.last_goal = cand.tolist(), None           # introspection for the parity tests
        if self.native_search:
            path = self._search_native(pose, stg, cand, collision_list, passable_list, check_first_edge)
            if path is not NotImplemented:
                return path
        return self._search_python(pose, stg, self.last_candidates, collision_list, passable_list, check_first_edge)

    def _search_native(self, pose, stg, cand, collision_list, passable_list, check_first_edge):
        """The candidate loop in C++ (csrc/nbp_plan_host.cpp); NotImplemented when a path node lies outside the value map
        (random heading from the rollout's Python stream: the Python form below handles that replan from the start)."""
        self._sync_collisions(collision_list)
        self._sync_passable(passable_list)
        hist, n_hist = self._history()
        cand32 = np.ascontiguousarray(cand, np.int32)
        lo, hi = self.grid_range
        plen, goal, n_new = C.c_int(0), C.c_int(0), C.c_int(0)
        P, E = len(self.idx3_list), len(self._edges_i32)
        rc = _lib.lib().nbp_plan_search_host(
            P, self._idx3_i32.ctypes.data, self._xyz_f32.ctypes.data, E, self._edges_i32.ctypes.data,
            self._edge_first.ctypes.data, self._mesh_hit_u8.ctypes.data, stg[2].data_ptr(), self._coll_mask.ctypes.data,
            self._pass_mask.ctypes.data, cand32.ctypes.data, len(cand32), self.node_index[tuple(self.camera.cam_idx[:3])],
            float(np.float32(pose[0])), float(np.float32(pose[2])), stg[3].data_ptr(), self.V, float(np.float32(lo)),
            float(np.float32(self.V / (hi - lo))), hist.ctypes.data, n_hist, int(bool(check_first_edge)), P,
            self._path_nodes.ctypes.data, self._path_heads.ctypes.data, C.byref(plen), C.byref(goal),
            self._new_coll.ctypes.data, len(self._new_coll) // 2, C.byref(n_new))
        _lib.check(rc, "nbp_plan_search_host")
        if plen.value == -2:
            return NotImplemented
        for a, b in self._new_coll[:2 * n_new.value].reshape(-1, 2).tolist():
            A, B = list(self.idx3_list[a]), list(self.idx3_list[b])
            collision_list.append([A, B])
            collision_list.append([B, A])
        self.last_goal = goal.value if goal.value >= 0 else None
        if plen.value < 0:
            return None
        L = plen.value
        return [[*self.idx3_list[n], 2, h] for n, h in zip(self._path_nodes[:L].tolist(), self._path_heads[:L].tolist())]

    def _search_python(self, pose, stg, cand, collision_list, passable_list, check_first_edge):
        cam = self.camera
        blocked_h = stg[2].numpy().astype(bool)
        out1_h = stg[3].numpy().copy()
        start_id = self.node_index[tuple(cam.cam_idx[:3])]
        hist = np.asarray(cam.cam_idx_history, np.int64).reshape(-1, 5)
        tree, tree_version = None, -1
        path = None
        for ci in cand:
            if tree is None or tree_version != len(collision_list):
                tree = planner_host.level_order_tree(self.nbrs, self._edge_ok(blocked_h, collision_list, passable_list),
                                                     start_id)
                tree_version = len(collision_list)
            if ci not in tree:
                path = None
                continue
            ids, cur = [], ci
            while cur >= 0:
                ids.append(cur)
                cur = tree[cur]
            nodes = [tuple(self.idx3[n].tolist()) for n in ids[::-1]]
            full = planner_host.choose_headings(nodes, self.xyz, self.node_index, pose, out1_h, hist, self.V,
                                                self.grid_range, rng=self.rng)
            path = full[1:]
            if len(path) > 0:
                if not check_first_edge or not self.edge_hits_mesh(cam.cam_idx[:3], path[0][:3]):
                    self.last_goal = ci
                    break
                collision_list.append([list(cam.cam_idx[:3]), path[0][:3]])
                collision_list.append([path[0][:3], list(cam.cam_idx[:3])])
        return path

    def edge_hits_mesh(self, a, b):
        """line_segment_mesh_intersection (macarons_utils.py:120-151) between two ADJACENT lattice positions:
        the mesh is static, so the predicate is tabulated once per scene for every lattice edge (one launch)
        instead of one GPU query + sync per step (nbp_planning.py:142,245)."""
        a, b = tuple(int(v) for v in a), tuple(int(v) for v in b)
        if a == b:
            return False                                   # zero-length segment (turn in place)
        return bool(self.mesh_hit[self.edge_id[(self.node_index[a], self.node_index[b])]])

    def replan(self, pose, out1, out2, maps6, traj_img, collision_list, passable_list, check_first_edge=True):
        """Returns the path as a list of [i,j,k,2,h] (first node dropped, like ref :416) or None."""
        self.replan_enqueue(pose, out1, out2, maps6, traj_img, collision_list)
        torch.cuda.current_stream().synchronize()
        return self.replan_finish(collision_list, passable_list, check_first_edge)
