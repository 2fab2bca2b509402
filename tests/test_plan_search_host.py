"""nbp_plan_search_host (csrc/nbp_plan_host.cpp, a HOST function of the C ABI: no GPU needed) against the Python
statement of the same loop (utility/planner_host.py: level_order_tree + choose_headings, themselves pinned to the
reference's heapq search and heading choice by tests/test_oracle_planner_sim.py), on random lattices with holes, random
blocked / collision / passable masks, mesh hits on first edges, used headings, and on the replan golden fixture."""
import ctypes as C
import os

import numpy as np

from nextbestpath_amd import _lib
from nextbestpath_amd.utility import planner_host as ph


def _graph(idx):
    nodes = {t: n for n, t in enumerate(idx)}
    edges, nbrs = [], [[] for _ in idx]
    for n, (i, j, k) in enumerate(idx):
        for nb in ((i + 1, j, k), (i - 1, j, k), (i, j, k + 1), (i, j, k - 1)):
            if nb in nodes:
                nbrs[n].append((nodes[nb], len(edges)))
                edges.append((n, nodes[nb]))
    first = np.zeros(len(idx) + 1, np.int32)
    if edges:
        np.cumsum(np.bincount(np.asarray(edges)[:, 0], minlength=len(idx)), out=first[1:])
    return nodes, edges, nbrs, first


def _python_search(cand, idx, pos, nodes, edges, nbrs, mesh_hit, blocked, coll, pas, start_id, pose, out1, hist, V, gr,
                   check_first_edge, rng):
    """LatticePlanner._search_python on plain arrays.  Returns (path, goal, new collision pairs) or None when a random
    heading would be drawn."""
    coll = coll.copy()
    eid = {e: q for q, e in enumerate(edges)}
    tree, dirty, path, goal, new = None, True, None, -1, []

    class Drawn(Exception):
        pass

    class Rng:
        def randint(self, a, b):
            raise Drawn

    for ci in cand:
        if dirty:
            ok = pas.astype(bool) | (~blocked.astype(bool) & ~coll.astype(bool))
            tree, dirty = ph.level_order_tree(nbrs, ok, start_id), False
        if ci not in tree:
            path = None
            continue
        ids, cur = [], ci
        while cur >= 0:
            ids.append(cur)
            cur = tree[cur]
        try:
            full = ph.choose_headings([idx[n] for n in ids[::-1]], pos, nodes, pose, out1, hist, V, gr, rng=Rng())
        except Drawn:
            return None
        path = full[1:]
        if path:
            a, b = ids[-1], ids[-2]
            if not check_first_edge or not mesh_hit[eid[(a, b)]]:
                goal = ci
                break
            new.append((a, b))
            coll[eid[(a, b)]] = 1
            coll[eid[(b, a)]] = 1
            dirty = True
    return path, goal, new


def _native_search(cand, idx, pos, edges, first, mesh_hit, blocked, coll, pas, start_id, pose, out1, hist, V, gr,
                   check_first_edge):
    L = _lib.lib()
    P, E = len(idx), len(edges)
    idx3 = np.ascontiguousarray(idx, np.int32).reshape(-1, 3)
    pos3 = np.ascontiguousarray(pos, np.float32)
    e2 = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
    cand32 = np.ascontiguousarray(cand, np.int32)
    h5 = np.ascontiguousarray(hist, np.int32).reshape(-1, 5)
    o1 = np.ascontiguousarray(out1, np.float32)
    nodes_out, heads_out, new = np.zeros(P + 1, np.int32), np.zeros(P + 1, np.int32), np.zeros(2 * (len(cand) + 1), np.int32)
    plen, goal, n_new = C.c_int(0), C.c_int(0), C.c_int(0)
    lo, hi = gr
    rc = L.nbp_plan_search_host(P, idx3.ctypes.data, pos3.ctypes.data, E, e2.ctypes.data, first.ctypes.data,
                                mesh_hit.ctypes.data, blocked.ctypes.data, coll.ctypes.data, pas.ctypes.data,
                                cand32.ctypes.data, len(cand32), start_id, float(np.float32(pose[0])),
                                float(np.float32(pose[2])), o1.ctypes.data, V, float(np.float32(lo)),
                                float(np.float32(V / (hi - lo))), h5.ctypes.data, len(h5), int(check_first_edge), P,
                                nodes_out.ctypes.data, heads_out.ctypes.data, C.byref(plen), C.byref(goal),
                                new.ctypes.data, len(cand) + 1, C.byref(n_new))
    assert rc == 0
    if plen.value == -2:
        return None
    path = None if plen.value < 0 else [[*idx[n], 2, int(h)] for n, h in zip(nodes_out[:plen.value], heads_out[:plen.value])]
    return path, goal.value, [tuple(p) for p in new[:2 * n_new.value].reshape(-1, 2).tolist()]


def test_native_search_matches_python_form_on_random_lattices():
    rng = np.random.default_rng(11)
    drawn = found = none = retried = 0
    for trial in range(300):
        Lx, Lz = int(rng.integers(2, 10)), int(rng.integers(2, 10))
        idx = [(i, 1, k) for i in range(Lx) for k in range(Lz) if rng.random() > 0.1]
        if len(idx) < 2:
            continue
        nodes, edges, nbrs, first = _graph(idx)
        spacing = float(rng.choice([3.0, 6.0, 11.0]))
        pos = np.array([[i * spacing + 0.37, 0.5, k * spacing - 1.21] for i, _, k in idx], np.float32)
        E = len(edges)
        blocked = (rng.random(E) < 0.25).astype(np.uint8)
        coll = (rng.random(E) < 0.1).astype(np.uint8)
        pas = (rng.random(E) < 0.1).astype(np.uint8)
        mesh_hit = (rng.random(E) < 0.3).astype(np.uint8)
        start_id = int(rng.integers(len(idx)))
        pose = np.array([*pos[start_id], 0.0, 0.0], np.float32)
        if rng.random() < 0.3:                                  # a map centre away from the camera: off-map nodes
            pose[:3] += rng.normal(0, 20, 3).astype(np.float32)
        V = 64
        out1 = rng.standard_normal((8, V, V)).astype(np.float32)
        out1[:, ::3, ::2] = np.round(out1[:, ::3, ::2])          # ties between headings
        hist = [(*idx[int(rng.integers(len(idx)))], 2, int(rng.integers(8))) for _ in range(int(rng.integers(0, 120)))]
        hist += [(*idx[start_id], int(rng.integers(5)), int(rng.integers(8)))]
        hist = np.asarray(hist, np.int64).reshape(-1, 5)
        cand = rng.permutation(len(idx))[:int(rng.integers(0, len(idx) + 1))].tolist()
        check = bool(rng.random() < 0.8)
        args = (start_id, pose, out1, hist, V, (-40, 40), check)
        want = _python_search(cand, idx, pos, nodes, edges, nbrs, mesh_hit, blocked, coll, pas, *args, rng=None)
        got = _native_search(cand, idx, pos, edges, first, mesh_hit, blocked, coll, pas, *args)
        assert got == want, (trial, got, want)
        if want is None:
            drawn += 1
        else:
            found += want[1] >= 0
            none += want[0] is None
            retried += len(want[2]) > 0
    assert drawn > 5 and found > 50 and none > 5 and retried > 20, (drawn, found, none, retried)   # every branch was taken


def test_native_search_on_the_replan_fixture(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "replan.npz"), allow_pickle=False))
    idx = [tuple(r) for r in g["idx"].tolist()]
    nodes, edges, nbrs, first = _graph(idx)
    eid = {e: q for q, e in enumerate(edges)}
    import oracle.planner as opl
    layout = g["obst"].astype(np.float32)
    blocked = np.array([opl.edge_blocked(g["pos"][a], g["pos"][b], g["pose"], layout) for a, b in edges], np.uint8)
    coll, pas = np.zeros(len(edges), np.uint8), np.zeros(len(edges), np.uint8)
    for a, b in g["collision"].tolist():
        q = eid.get((nodes.get(tuple(a), -1), nodes.get(tuple(b), -1)))
        if q is not None:
            coll[q] = 1
    for a, b in g["passable"].tolist():
        q = eid.get((nodes.get(tuple(a), -1), nodes.get(tuple(b), -1)))
        if q is not None:
            pas[q] = 1
    mesh_hit = np.zeros(len(edges), np.uint8)
    start_id = nodes[tuple(g["start"].tolist())]
    off = 0
    for gi, n in zip(g["goals"].tolist(), g["path_lens"].tolist()):
        got = _native_search([gi], idx, g["pos"], edges, first, mesh_hit, blocked, coll, pas, start_id, g["pose"],
                             g["out1"][0], g["cam_hist"], 64, (-40, 40), False)
        if n < 0:
            assert got[0] is None
            continue
        want = g["paths"][off:off + n]
        off += n
        assert np.array_equal(np.array(got[0]).reshape(-1, 5), want)       # the reference's own paths and headings


def test_native_search_rejects_bad_arguments():
    L = _lib.lib()
    z = np.zeros(8, np.int32)
    o = C.c_int(0)
    rc = L.nbp_plan_search_host(0, z.ctypes.data, z.ctypes.data, 0, z.ctypes.data, z.ctypes.data, z.ctypes.data,
                                z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 0, 0, 0.0, 0.0, z.ctypes.data,
                                64, -40.0, 0.8, z.ctypes.data, 0, 1, 4, z.ctypes.data, z.ctypes.data, C.byref(o),
                                C.byref(o), z.ctypes.data, 4, C.byref(o))
    assert rc == -1
