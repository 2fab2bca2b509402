"""Host-side planner logic that stays in Python (tiny graphs, tie-breaking semantics of the
reference): the uniform-cost search and the per-node heading choice of
generate_Dijkstra_path (next_best_path/utility/long_term_utils.py:334-418), consuming the
edge mask computed in one launch by nbp_edges_blocked_u8."""
from __future__ import annotations

import heapq
import random

import numpy as np

f32 = np.float32


def value_cell(p3, pose, V=64, grid_range=(-40, 40)):
    """V-grid cell of a world position (utils.py:160-196 arithmetic in fp32)."""
    lo, hi = grid_range
    sc = f32(V / (hi - lo))
    v0 = -(f32(p3[2]) - f32(pose[2]))
    v1 = -(f32(p3[0]) - f32(pose[0]))
    return int(np.rint((v0 - f32(lo)) * sc)), int(np.rint((v1 - f32(lo)) * sc))


def dijkstra_tree(nodes, start, passable):
    """Search of long_term_utils.py:366-383 run to exhaustion (the tree does not depend on the goal:
    uniform cost, came_from fixed at first discovery, heap ordered by (cost, tuple))."""
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


def level_order_tree(nbrs, ok, start_id):
    """Same tree as dijkstra_tree, on integer node ids and a precomputed edge mask.
    Node ids must be assigned in lexicographic (i,j,k) order: then the reference's heap order
    (cost, tuple) == expanding each BFS level in increasing id, and came_from == first discoverer.
    nbrs[u] = [(v, edge_id), ...] in the reference's neighbour order (+x, -x, +z, -z)."""
    parent = {start_id: -1}
    level = [start_id]
    while level:
        nxt = []
        for u in sorted(level):
            for v, q in nbrs[u]:
                if ok[q] and v not in parent:
                    parent[v] = u
                    nxt.append(v)
        level = nxt
    return parent


def path_from_tree(came_from, goal):
    if goal not in came_from:
        return None
    path, cur = [], goal
    while cur:
        path.append(cur)
        cur = came_from[cur]
    return path[::-1]


def choose_headings(path, positions, node_index, pose, out1, cam_idx_history, V=64, grid_range=(-40, 40), rng=None):
    """long_term_utils.py:390-413: for every node of the path pick the best-valued heading that the
    camera has not used at that node yet (elevation index fixed to 2); random heading off-map, drawn from
    `rng` (the rollout's own random.Random, so that concurrent rollouts do not share a stream).
    out1: host array [8,V,V]; cam_idx_history: host int array [n,5].  Returns [[i,j,k,2,h], ...]."""
    rng = rng or random
    hist = {tuple(int(v) for v in row) for row in np.asarray(cam_idx_history).tolist()}
    out = []
    for step in path:
        g0, g1 = value_cell(positions[node_index[tuple(step)]], pose, V, grid_range)
        if 0 <= g0 < V and 0 <= g1 < V:
            order = np.argsort(-out1[:, g0, g1], kind="stable")
            h = int(order[-1])
            for cand in order.tolist():
                h = int(cand)
                if (step[0], step[1], step[2], 2, h) not in hist:
                    break
        else:
            for _ in range(64):            # the reference loops forever when all 8 headings were used at this node
                h = rng.randint(0, 7)
                if (step[0], step[1], step[2], 2, h) not in hist:
                    break
        out.append([int(step[0]), int(step[1]), int(step[2]), 2, h])
    return out
