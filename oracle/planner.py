"""numpy / pure-Python restatement of the planner glue.  PINNED by tests/golden/planner.npz
(outputs of the reference functions themselves).

  * bresenham_line, line_across_image_pixel: next_best_path/utility/long_term_utils.py:277-331
  * check_pixel_values: macarons/utility/macarons_utils.py:86-100
  * obstacle fusion + candidate scoring: next_best_path/testers/nbp_planning.py:166-233
  * generate_Dijkstra_path: long_term_utils.py:334-418 (heapq on (cost, tuple), uniform cost)
  * calculate_coverage_percentage / compute_auc: long_term_utils.py:437-468, 488-490
"""
import heapq

import numpy as np

from . import maps as omaps
from . import sampling

f32 = np.float32


def bresenham_line(x0, y0, x1, y1):
    pts = []
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx = 1 if x0 < x1 else -1
    sy = 1 if y0 < y1 else -1
    err = dx - dy
    while True:
        pts.append((x0, y0))
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 > -dy:
            err -= dy
            x0 += sx
        if e2 < dx:
            err += dx
            y0 += sy
    return pts


def _cell(p3, pose, size, grid_range):
    t = omaps.transform_points_to_n_pieces(np.asarray(p3, f32).reshape(1, 3), pose)
    return omaps.get_point_position_in_the_img(t[0, 0], (size, size), grid_range)


def edge_blocked(p1, p2, pose, layout, grid_range=(-40, 40)):
    """line_across_image_pixel: layout [S,S] of {0,1}."""
    S = layout.shape[0]
    a, b = _cell(p1, pose, S, grid_range), _cell(p2, pose, S, grid_range)
    if not (0 <= a[0] < S and 0 <= a[1] < S and 0 <= b[0] < S and 0 <= b[1] < S):
        return True
    n = sum(1 for (x, y) in bresenham_line(int(a[0]), int(a[1]), int(b[0]), int(b[1])) if layout[x, y] == 1)
    return n >= 2


def check_pixel_values(proj, cell, size=10):
    H, W = proj.shape
    x, y = int(cell[0]), int(cell[1])
    region = proj[max(x - size, 0):min(x + size + 1, H), max(y - size, 0):min(y + size + 1, W)]
    return bool((region == 1).any())


def fuse_obstacle(out2, maps6, traj, threshold=0.13):
    """nbp_planning.py:166-191.  Returns (obst [S,S] float 0/1, fullproj [S,S] clipped to 1)."""
    obst = (out2 >= f32(threshold)).astype(f32)
    full = (((maps6[0] + maps6[1]) + maps6[2]) + maps6[3]) + maps6[4]
    band = (maps6[5] > 0).astype(f32)
    obst = np.where(full > 0, band, obst)
    obst = np.where(traj > 0, f32(0), obst)
    return obst.astype(f32), np.minimum(full, f32(1)).astype(f32)


def score_candidates(positions, pose, out1, fullproj, skip=None, V=64, grid_range=(-40, 40)):
    """nbp_planning.py:194-231.  positions [P,3]; out1 [8,V,V]; returns (valid, cells, scores float64)."""
    S = fullproj.shape[0]
    P = len(positions)
    valid = np.zeros(P, bool)
    cells = np.zeros((P, 2), np.int64)
    scores = np.zeros(P, np.float64)
    max_gain = out1.max(0)
    for i in range(P):
        if skip is not None and skip[i]:
            continue
        g = _cell(positions[i], pose, V, grid_range)
        if not (0 <= g[0] < V and 0 <= g[1] < V):
            continue
        s = _cell(positions[i], pose, S, grid_range)
        dens = fullproj[int(s[0]), int(s[1])]            # python/torch negative index wraps
        if not check_pixel_values(fullproj, s):
            continue
        valid[i] = True
        cells[i] = g
        scores[i] = float(max_gain[g[0], g[1]]) - 10 * float(dens)
    return valid, cells, scores


def dijkstra_tree(nodes, start, passable):
    """The search of generate_Dijkstra_path run to exhaustion: uniform cost, heap of (cost, tuple),
    neighbours in the order +x, -x, +z, -z; came_from is fixed at first discovery.
    nodes: set of (i,j,k) tuples; passable(a, b) -> bool."""
    frontier = [(0, start)]
    came_from = {start: None}
    cost = {start: 0}
    while frontier:
        _, cur = heapq.heappop(frontier)
        x, y, z = cur
        for nb in ((x + 1, y, z), (x - 1, y, z), (x, y, z + 1), (x, y, z - 1)):
            if nb in nodes and passable(cur, nb):
                nc = cost[cur] + 1
                if nb not in cost or nc < cost[nb]:
                    cost[nb] = nc
                    heapq.heappush(frontier, (nc, nb))
                    came_from[nb] = cur
    return came_from


def path_from_tree(came_from, goal):
    if goal not in came_from:
        return None
    path, cur = [], goal
    while cur:
        path.append(cur)
        cur = came_from[cur]
    return path[::-1]


def coverage(gt, pc, threshold=1.0, weight=2, seed=0):
    """calculate_coverage_percentage with the seeded bijection instead of torch.randperm and
    direct-difference distances (the reference's cdist may use the matmul formulation)."""
    gt, pc = np.asarray(gt, f32), np.asarray(pc, f32)
    if len(pc) == 0:
        return 0.0, 0
    k = int(len(gt) * weight)
    if len(pc) > k:
        pc = pc[sampling.perm_index(np.arange(k), len(pc), seed)]
    cnt = 0
    for i in range(0, len(gt), 512):
        g = gt[i:i + 512]
        e = g[:, None, :] - pc[None, :, :]
        d2 = (e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]
        cnt += int((np.sqrt(d2.min(1)) < f32(threshold)).sum())
    return float(f32(cnt) / f32(len(gt))), cnt


def compute_auc(y, dx=1 / 40):
    y = np.asarray(y, np.float64)
    trap = getattr(np, "trapezoid", None) or np.trapz
    return float(trap(y, dx=dx) + y[0] * dx / 2.0)
