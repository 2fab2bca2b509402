"""CPU: planner / simulator oracles vs the reference's golden vectors and analytic known answers."""
import os

import numpy as np

from oracle import camera as ocam
from oracle import mesh_rays
from oracle import planner as opl
from oracle import raster as orast
from oracle import sampling


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ------------------------------------------------------------------ pinned by the reference
def test_bresenham_vs_reference(golden_dir):
    g = _g(golden_dir, "planner.npz")
    off = 0
    for (x0, y0, x1, y1), n in zip(g["ends"].tolist(), g["line_lens"].tolist()):
        want = [tuple(p) for p in g["line_pts"][off:off + n].tolist()]
        assert opl.bresenham_line(x0, y0, x1, y1) == want
        off += n


def test_edge_blocked_and_window_check_vs_reference(golden_dir):
    g = _g(golden_dir, "planner.npz")
    layout = g["layout"][0, 0].astype(np.float32)
    got = [opl.edge_blocked(a, b, g["edge_pose"], layout) for a, b in zip(g["edge_p1"], g["edge_p2"])]
    assert got == g["edge_blocked"].tolist()
    assert 0 < sum(got) < len(got)
    proj = g["proj"][0, 0].astype(np.float32)
    assert [opl.check_pixel_values(proj, c) for c in g["cpv_cells"]] == g["cpv"].tolist()


def test_coverage_vs_reference(golden_dir):
    g = _g(golden_dir, "planner.npz")
    gt, pc = g["cov_gt"], g["cov_pc"]
    # same subset as the reference drew (its randperm is recorded in the fixture)
    sub = pc[g["cov_perm"]]
    frac, cnt = opl.coverage(gt, sub, seed=0)          # len(sub) == 2G -> no resampling
    assert abs(frac - float(g["cov"])) <= 2.0 / len(gt)     # cdist matmul vs direct difference at d ~ 1.0
    frac_small, _ = opl.coverage(gt, pc[:1000])
    assert abs(frac_small - float(g["cov_small"])) <= 2.0 / len(gt)
    assert opl.coverage(gt, pc[:0])[0] == float(g["cov_empty"]) == 0.0
    assert abs(opl.compute_auc(np.linspace(0, 0.8, 101)) - float(g["auc"])) < 1e-12


def test_fusion_scoring_vs_reference(golden_dir):
    g = _g(golden_dir, "replan.npz")
    maps6 = np.zeros((6, 256, 256), np.float32)
    maps6[0] = g["full"]
    maps6[5] = g["band"]
    obst, fullproj = opl.fuse_obstacle(g["out2"], maps6, g["traj"].astype(np.float32))
    assert np.array_equal(obst, g["obst"].astype(np.float32))
    assert np.array_equal(fullproj, g["fullproj"].astype(np.float32))
    valid, cells, scores = opl.score_candidates(g["pos"], g["pose"], g["out1"][0], fullproj, skip=g["skip"])
    ids = np.nonzero(valid)[0]
    assert np.array_equal(ids, g["cand"][:, 0])
    assert np.array_equal(cells[ids], g["cand"][:, 1:3])
    assert np.array_equal(scores[ids], g["cand_score"])                 # float64, exact
    order = sorted(range(len(ids)), key=lambda r: scores[ids[r]], reverse=True)
    assert order == g["cand_order"].tolist()


def test_dijkstra_vs_reference(golden_dir):
    from nextbestpath_amd.utility import planner_host as ph
    g = _g(golden_dir, "replan.npz")
    idx = [tuple(r) for r in g["idx"].tolist()]
    nodes = {t: n for n, t in enumerate(idx)}
    layout = g["obst"].astype(np.float32)
    coll = [[list(a), list(b)] for a, b in g["collision"].tolist()]
    pas = [[list(a), list(b)] for a, b in g["passable"].tolist()]

    def passable(a, b):
        if [list(a), list(b)] in pas:
            return True
        return (not opl.edge_blocked(g["pos"][nodes[a]], g["pos"][nodes[b]], g["pose"], layout)) and \
            [list(a), list(b)] not in coll

    tree = opl.dijkstra_tree(set(nodes), tuple(g["start"].tolist()), passable)
    # the product's integer level-order search builds the same tree as the heapq search
    edges, nbrs = [], [[] for _ in idx]
    for n, (i, j, k) in enumerate(idx):
        for nb in ((i + 1, j, k), (i - 1, j, k), (i, j, k + 1), (i, j, k - 1)):
            if nb in nodes:
                nbrs[n].append((nodes[nb], len(edges)))
                edges.append((n, nodes[nb]))
    ok = [passable(idx[a], idx[b]) for a, b in edges]
    par = ph.level_order_tree(nbrs, ok, nodes[tuple(g["start"].tolist())])
    assert {idx[v]: (idx[p] if p >= 0 else None) for v, p in par.items()} == tree
    assert ph.dijkstra_tree(set(nodes), tuple(g["start"].tolist()), passable) == tree
    off = 0
    for gi, n in zip(g["goals"].tolist(), g["path_lens"].tolist()):
        path = opl.path_from_tree(tree, idx[gi])
        if n < 0:
            assert path is None
            continue
        want = g["paths"][off:off + n]
        off += n
        assert [list(p) for p in path[1:]] == want[:, :3].tolist()
        # heading choice (long_term_utils.py:395-404) through the product's host helper
        got = ph.choose_headings(path, g["pos"], nodes, g["pose"], g["out1"][0], g["cam_hist"].astype(np.int64))
        assert np.array_equal(np.array(got)[1:], want)


# ------------------------------------------------------------------ analytic known answers (unpinned libraries)
def test_sampling_bijection():
    for n in (1, 2, 3, 5, 64, 1000, 116_736):
        p = sampling.perm_index(np.arange(n), n, 1234)
        assert np.array_equal(np.sort(p), np.arange(n))
    a = sampling.perm_index(np.arange(1000), 116_736, 7)
    assert len(set(a.tolist())) == 1000 and a.max() < 116_736
    assert not np.array_equal(a, sampling.perm_index(np.arange(1000), 116_736, 8))


def test_look_at_and_unproject_plane():
    # camera at origin looking down +z (elev 0, azim 0 in the reference's convention)
    R, T = ocam.camera_RT([0.0, 0.0, 0.0], [0.0, 0.0])
    view = np.array([[0.0, 0.0, 5.0]]) @ R + T
    assert abs(abs(view[0, 2]) - 5.0) < 1e-5 and abs(view[0, 0]) < 1e-5
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
    # constant depth d: every un-projected point lies on the plane z_view = d, and re-projects to its pixel
    H, W, d = 16, 28, 7.5
    X = np.array([3.0, 1.0, -2.0])
    R, T = ocam.camera_RT(X, [10.0, 135.0])
    pts = ocam.unproject(np.full((H, W), d, np.float32), R, T)
    v = pts.astype(np.float64) @ R + T
    assert np.allclose(v[:, 2], d, atol=1e-4)
    ndc_x, ndc_y = ocam.ndc_tables(H, W)
    assert np.allclose(v[:, 0] / (v[:, 2] * ocam.TAN_HALF_FOV), ndc_x.reshape(-1), atol=1e-4)
    assert np.allclose(v[:, 1] / (v[:, 2] * ocam.TAN_HALF_FOV), ndc_y.reshape(-1), atol=1e-4)
    # the camera centre un-projects nothing: depth 0 -> the camera position itself
    c = ocam.unproject(np.zeros((2, 2), np.float32), R, T)
    assert np.allclose(c, X, atol=1e-5)


def test_partial_point_cloud_counts():
    H, W = 32, 57
    rng = np.random.default_rng(0)
    depth = rng.uniform(1, 100, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.2] = -1
    R, T = ocam.camera_RT([0, 0, 0], [0, 0])
    pts, nv = ocam.partial_point_cloud(depth, None, R, T, 0.05, 70.0, seed=3)
    assert nv == int(((depth > -1) & (depth < 70)).sum())
    assert len(pts) == int(nv * 0.05)
    allp = ocam.unproject(depth, R, T)
    # every kept point is one of the valid pixels, no pixel twice
    keys = {tuple(p) for p in allp[((depth > -1) & (depth < 70)).reshape(-1)].tolist()}
    got = [tuple(p) for p in pts.tolist()]
    assert set(got) <= keys and len(set(got)) == len(got)


def _box_room(h=4.0):
    """Axis-aligned room [-h,h]^3 seen from inside: 12 triangles."""
    v = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], np.int32)
    return v, f


def test_raster_plane_and_box_closed_form():
    H, W = 24, 40
    t = ocam.TAN_HALF_FOV
    R, T = ocam.camera_RT([0, 0, 0], [0.0, 0.0])
    v, f = _box_room(4.0)
    z = orast.raster_zbuf(v, f, R, T, H, W, t)
    assert (z > 0).all()                       # closed room: no background pixel, no cracks on shared edges
    s = min(H, W)
    col, row = np.meshgrid(np.arange(W), np.arange(H))
    dx = (W - (2 * col + 1)) / s * t
    dy = (H - (2 * row + 1)) / s * t
    want = 4.0 / np.maximum(np.maximum(np.abs(dx), np.abs(dy)), 1.0)      # first wall hit along the pixel ray
    assert np.allclose(z, want, rtol=1e-5)
    # a wall crossing the near plane is clipped, not dropped: camera 0.3 from the x = +4 wall, looking along it
    R2, T2 = ocam.camera_RT([3.7, 0, 0], [0.0, 0.0])
    z2 = orast.raster_zbuf(v, f, R2, T2, H, W, t)
    vis = z2[z2 > 0]
    assert (vis > 0.5).all() and vis.min() < 1.0           # the wall is visible right up to the clip plane
    assert (z2 < 0).any() and (z2 < 0).sum() < 0.5 * z2.size   # only the sliver nearer than z_clip is lost


def test_mesh_ray_queries_on_cube():
    v, f = _box_room(1.0)
    assert mesh_rays.point_in_mesh([0.1, 0.2, -0.3], v, f)
    assert not mesh_rays.point_in_mesh([1.5, 0.2, -0.3], v, f)
    assert mesh_rays.axis_ray_counts([-3.0, 0.2, 0.1], v, f) == [0, 2, 0]
    assert mesh_rays.segment_hits_mesh([0.1, 0.1, 0.1], [2.0, 0.1, 0.1], v, f)
    assert not mesh_rays.segment_hits_mesh([0.1, 0.1, 0.1], [0.8, 0.1, 0.1], v, f)      # stops before the wall
    assert not mesh_rays.segment_hits_mesh([0.1, 0.1, 0.1], [0.1, 0.1, 0.1], v, f)      # zero length


def test_pose_lattice_order_and_values():
    idx, poses = ocam.pose_lattice([-10.0, 0.0, -20.0], 3, 1, 4, 5, 8)
    assert len(idx) == 3 * 1 * 4 * 5 * 8
    assert idx[1].tolist() == [0, 0, 0, 0, 1] and idx[8].tolist() == [0, 0, 0, 1, 0]      # i-major ... azimuth fastest
    k = np.nonzero((idx == [2, 0, 3, 2, 4]).all(1))[0][0]
    assert np.allclose(poses[k], [-10 + 6, 3.3, -20 + 9, 0.0, 180.0])


def test_carving_oracle_bilinear_matches_torch_grid_sample_and_plane_kat():
    """A20: the oracle's bilinear lookup == torch.nn.functional.grid_sample (the reference's own op,
    macarons_utils.py:2939-2944); plane known answer: points in front of the depth map are carved."""
    import torch
    H, W, zfar = 24, 40, 750.0
    rng = np.random.default_rng(2)
    depth = rng.uniform(5, 40, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.1] = -1
    R, T = ocam.camera_RT([1.0, 2.0, -3.0], [15.0, 70.0])
    X = np.array([1.0, 2.0, -3.0])
    pts = (X + rng.normal(0, 25, (4000, 3))).astype(np.float32)
    st = [np.zeros(4000, np.float32), np.zeros(4000, np.float32), np.ones(4000, np.float32), np.ones(4000, np.float32)]
    inf, sd = ocam.carve_update(pts, depth, None, R, T, zfar, 60.0, 10.0, 0.95, *st)
    assert 100 < inf.sum() < 3000
    # reference formulation with torch ops on the in-frustum points
    v = pts.astype(np.float64) @ R.astype(np.float64) + T.astype(np.float64)
    t = float(ocam.TAN_HALF_FOV)
    nx, ny = v[:, 0] / (v[:, 2] * t), v[:, 1] / (v[:, 2] * t)
    s = min(H, W)
    grid = torch.tensor(np.stack([-s / W * nx[inf], -s / H * ny[inf]], 1), dtype=torch.float32).view(1, -1, 1, 2)
    dd = torch.from_numpy(np.where(depth > -1, depth, np.float32(1.1 * zfar)).astype(np.float32)).view(1, 1, H, W)
    samp = torch.nn.functional.grid_sample(dd, grid, mode="bilinear", padding_mode="border", align_corners=False)
    want = v[inf, 2] - samp.view(-1).numpy()
    assert np.allclose(sd[inf], want, rtol=1e-4, atol=2e-3)
    # state update rules
    assert np.all(st[0][inf] == 1) and np.all(st[0][~inf] == 0) and np.all(st[3][inf] == 0) and np.all(st[3][~inf] == 1)
    assert np.array_equal(st[2][inf], (sd[inf] >= -10.0).astype(np.float32))
    # plane: constant depth 20 -> points at z_view < 20 - tol are carved (occ 0), behind stay occupied
    plane = np.full((H, W), 20.0, np.float32)
    zs = np.array([3.0, 15.0, 19.5, 25.0, 45.0], np.float32)
    view_pts = np.stack([np.zeros(5), np.zeros(5), zs], 1)
    world = ((view_pts - T) @ R.T).astype(np.float32)
    st2 = [np.zeros(5, np.float32), np.zeros(5, np.float32), np.ones(5, np.float32), np.ones(5, np.float32)]
    inf2, sd2 = ocam.carve_update(world, plane, None, R, T, zfar, 60.0, 1.0, 0.95, *st2)
    assert inf2.all() and np.allclose(sd2, zs - 20.0, atol=1e-3)
    assert st2[2].tolist() == [0.0, 0.0, 1.0, 1.0, 1.0]


def _carve_golden_sequence(golden_dir, step):
    """Runs `step(k, R, T, depth, mask, fov_range, state)` over the three views of tests/golden/carve.npz and checks the state
    after each against what the REFERENCE's own code left (make_golden.py::gen_carve)."""
    g = np.load(os.path.join(golden_dir, "carve.npz"))
    H, W = (int(v) for v in g["HW"])
    P = len(g["pts"])
    state = [g["n_inside_init"].reshape(P).copy(), g["n_behind_init"].reshape(P).copy(), np.ones(P, np.float32), np.ones(P, np.float32)]
    for k in range(3):
        inf, sd, state = step(k, g, H, W, state)
        want_mask = g[f"fov_mask{k}"]
        assert np.array_equal(inf, want_mask), (k, int((inf != want_mask).sum()))
        # signed distances: torch's grid_sample against the restated bilinear blend.  Where a "no hit" pixel (1.1 zfar = 1100) meets
        # a surface pixel the blend has a slope of ~1080 per pixel, and one fp32 ulp of the sample coordinate (4e-6 px at x ~ 60)
        # moves it by 4e-3: the bound is that, not a looser arithmetic (median deviation 0, 99th percentile 6e-5)
        dev = np.abs(sd[want_mask] - g[f"sd{k}"])
        assert dev.max() <= 8e-3 and np.percentile(dev, 99) <= 5e-4 and np.median(dev) <= 4e-6, (float(dev.max()), float(np.percentile(dev, 99)))
        for got, key in zip(state, ("n_inside", "n_behind", "occ", "oof")):
            assert np.array_equal(got, g[f"{key}{k}"].reshape(P)), (k, key)
    return g


def test_carving_oracle_vs_reference_golden(golden_dir):
    """oracle/camera.py::carve_update against Camera.get_points_in_fov + get_signed_distance_to_depth_maps + Scene.update_proxy_*
    of the reference run on the same points / depth maps (macarons_utils.py:2849-2949, 3329-3363): identical field-of-view masks,
    signed distances to 2e-3 (of depths up to 1100), identical counters and occupancy after each of three views.  The PyTorch3D
    projection feeding those functions was the restatement's own (the fixture's one stub): that part stays parity-unpinned."""
    def step(k, g, H, W, state):
        inf, sd = ocam.carve_update(g["pts"], g[f"depth{k}"], g[f"mask{k}"], g[f"R{k}"], g[f"T{k}"], float(g["zfar"]),
                                    float(g[f"fov_range{k}"]), float(g["tol"]), float(g["score_threshold"]), *state)
        return inf, sd, state
    g = _carve_golden_sequence(golden_dir, step)
    # the NDC window the reference's Camera.__init__ derived for this image size is the restatement's
    H, W = (int(v) for v in g["HW"])
    s = min(H, W)
    max_x, max_y = np.float32(W / s), np.float32(H / s)
    want = [max_x - (np.float32(W - 1) / np.float32(s - 1)) * np.float32(2), max_x, max_y - (np.float32(H - 1) / np.float32(s - 1)) * np.float32(2), max_y]
    assert np.allclose(g["ndc_minmax"], want, rtol=0, atol=1e-6)


def test_level_order_search_equals_heap_search_on_random_lattices():
    """Host logic of the product (integer level-order search on a precomputed edge mask) against the heapq restatement
    of generate_Dijkstra_path (long_term_utils.py:366-383) on 40 random lattices with random DIRECTED blocked edges:
    same came_from tree, hence the same paths and the reference's tie-breaking."""
    from nextbestpath_amd.utility import planner_host as ph
    rng = np.random.default_rng(5)
    for trial in range(40):
        L, Hh = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        idx = [(i, 0, k) for i in range(L) for k in range(Hh) if rng.random() > 0.1]      # lexicographic ids, holes
        if not idx:
            continue
        nodes = {t: n for n, t in enumerate(idx)}
        edges, nbrs = [], [[] for _ in idx]
        for n, (i, j, k) in enumerate(idx):
            for nb in ((i + 1, j, k), (i - 1, j, k), (i, j, k + 1), (i, j, k - 1)):
                if nb in nodes:
                    nbrs[n].append((nodes[nb], len(edges)))
                    edges.append((n, nodes[nb]))
        ok = rng.random(len(edges)) > 0.3
        okset = {(idx[a], idx[b]) for (a, b), o in zip(edges, ok) if o}
        start = idx[int(rng.integers(len(idx)))]
        tree = opl.dijkstra_tree(set(nodes), start, lambda a, b: (tuple(a), tuple(b)) in okset)
        par = ph.level_order_tree(nbrs, ok, nodes[start])
        assert {idx[v]: (idx[p] if p >= 0 else None) for v, p in par.items()} == tree, trial
        for goal in idx:
            want = opl.path_from_tree(tree, goal)
            if goal not in tree:
                assert want is None and nodes[goal] not in par
                continue
            got, cur = [], nodes[goal]
            while cur >= 0:
                got.append(idx[cur]); cur = par[cur]
            assert got[::-1] == [tuple(p) for p in want]


def test_obj_loader_formats(tmp_path):
    """Host logic: OBJ records as real scenes carry them (v/vt/vn triples, quads and n-gons, negative indices,
    comments, mtllib / usemtl / vt / vn lines)."""
    from nextbestpath_amd.simulator.mesh import load_obj, save_obj
    lines = ["# comment", "mtllib x.mtl", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0.5 0.5 1", "vt 0 0", "vn 0 0 1",
             "usemtl wall", "f 1/1/1 2/1/1 3/1/1 4/1/1", "f -1 -5 -4", "f 1//1 2//1 5//1 3//1 4//1"]
    p = tmp_path / "m.obj"
    p.write_text(chr(10).join(lines) + chr(10))
    v, f = load_obj(str(p))
    assert v.shape == (5, 3) and v.dtype == np.float32 and f.dtype == np.int32
    assert f.tolist() == [[0, 1, 2], [0, 2, 3], [4, 0, 1], [0, 1, 4], [0, 4, 2], [0, 2, 3]]
    q = tmp_path / "r.obj"
    save_obj(str(q), v, f)
    v2, f2 = load_obj(str(q))
    assert np.array_equal(v, v2) and np.array_equal(f, f2)


def test_threaded_c_twins_equal_the_scalar_restatements():
    """bench.py's cpu_baseline runs the raster and the map accumulation on the host's threads: (frame, row band) tasks and
    per-thread count planes must give the single-threaded / numpy results bit for bit."""
    from nextbestpath_amd.simulator.mesh import make_maze_mesh
    from nextbestpath_amd.utility.synthetic import make_point_cloud
    from oracle import camera as ocam
    from oracle import csim
    from oracle import maps as omaps
    verts, faces = make_maze_mesh(seed=2, cells=4, size=24.0, height=12.0, tess=3.0)[:2]
    poses = [([3.0, 3.3, -6.0], [0.0, 100.0]), ([0.0, 40.0, 0.0], [-89.0, 10.0]), ([1.0, 3.3, 2.0], [0.0, 200.0])]
    RT = [ocam.camera_RT(x, v) for x, v in poses]
    Rs, Ts = np.stack([r for r, _ in RT]), np.stack([t for _, t in RT])
    one = [csim.raster_zbuf(verts, faces, R, T, 64, 114, ocam.TAN_HALF_FOV) for R, T in RT]
    for omp, band in ((False, 64), (True, 16), (True, 7)):
        fr = csim.raster_zbuf_frames(verts, faces, Rs, Ts, 64, 114, ocam.TAN_HALF_FOV, band_rows=band, omp=omp)
        assert all(np.array_equal(fr[i], one[i]) for i in range(3)), (omp, band)
    assert (one[0] > 0).sum() > 100
    pc = make_point_cloud(60_000, seed=3).numpy()
    pose = np.array([1.0, 13.3, -2.0, 0, 0], np.float32)
    for ybins in (np.arange(0.5, 29.5 + 7.25, 7.25, dtype=np.float32), np.array([0.5, 5, 10, 15, 20, 25], np.float32)):
        want = omaps.accumulate_step_maps(pc, pose, ybins, S=128)
        for omp, nt in ((False, 0), (True, 0), (True, 3)):
            assert np.array_equal(csim.accumulate_step_maps(pc, pose, ybins, S=128, omp=omp, max_threads=nt), want), (omp, nt)
    assert want.sum() > 1000
