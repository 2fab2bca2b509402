"""GPU parity: simulator + planner kernels vs oracle/ (bit-exact for integer / index results and
for fp32 results whose op order is shared) and vs the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from nextbestpath_amd.utility import hipops as ho
from nextbestpath_amd import _lib
from oracle import camera as ocam
from oracle import csim
from oracle import mesh_rays
from oracle import planner as opl
from oracle import raster as orast
from oracle import sampling

pytestmark = pytest.mark.gpu
D = "cuda"


def _maze(seed=0, n=12, size=60.0, h=30.0):
    from nextbestpath_amd.simulator.mesh import make_maze_mesh
    return make_maze_mesh(seed=seed, cells=n, size=size, height=h)


def test_perm_index_host_matches_oracle(hip):
    for n, seed in ((1, 0), (7, 3), (1000, 99), (116_736, 12345)):
        js = np.arange(min(n, 300))
        want = sampling.perm_index(js, n, seed)
        got = np.array([hip.nbp_perm_index_host(int(j), n, seed) for j in js])
        assert np.array_equal(got, want)


@pytest.mark.parametrize("H,W", [(32, 57), (256, 456)])
def test_unproject_append_vs_oracle(hip, H, W):
    rng = np.random.default_rng(5)
    F_ = 5
    depth = rng.uniform(0.6, 120, (F_, H, W)).astype(np.float32)
    depth[rng.random((F_, H, W)) < 0.25] = -1
    depth[4] = -1                                                   # a frame that sees nothing
    poses = [([1.0, 3.3, -2.0], [0.0, 45.0]), ([4.0, 3.3, -2.0], [30.0, 90.0]), ([4.0, 3.3, 1.0], [-30.0, 200.0]),
             ([7.0, 3.3, 1.0], [0.0, 315.0]), ([7.0, 3.3, 4.0], [60.0, 0.0])]
    RT = [ocam.camera_RT(x, v) for x, v in poses]
    cams = ho.cams12(np.stack([r for r, _ in RT]), np.stack([t for _, t in RT]), D)
    cap = 100_000
    cloud = torch.zeros(cap, 3, device=D)
    cnt = torch.tensor([17], dtype=torch.int64, device=D)           # non-zero base: appended after 17 points
    counts = ho.unproject_append(torch.from_numpy(depth).to(D), None, cams, cloud, cnt, 0.05, 70.0, seed=11)
    want, nvs = [], []
    for f in range(F_):
        pts, nv = ocam.partial_point_cloud(depth[f], None, RT[f][0], RT[f][1], 0.05, 70.0, seed=11, frame_index=f)
        want.append(pts); nvs.append(nv)
    counts = counts.cpu().numpy()
    assert counts[:, 0].tolist() == nvs and counts[:, 1].tolist() == [len(w) for w in want]
    total = sum(len(w) for w in want)
    assert int(cnt.item()) == 17 + total
    got = cloud[17:17 + total].cpu().numpy()
    assert np.array_equal(got, np.concatenate(want, 0))             # bit-exact fp32 (shared op order)
    assert float(cloud[:17].abs().sum()) == 0.0
    # explicit mask input == derived mask
    cloud2 = torch.zeros(cap, 3, device=D)
    cnt2 = torch.zeros(1, dtype=torch.int64, device=D)
    m = torch.from_numpy((depth > -1).astype(np.uint8)).to(D)
    ho.unproject_append(torch.from_numpy(depth).to(D), m, cams, cloud2, cnt2, 0.05, 70.0, seed=11)
    assert torch.equal(cloud2[:total], cloud[17:17 + total])
    # capacity clamp
    small = torch.zeros(100, 3, device=D)
    c3 = torch.tensor([90], dtype=torch.int64, device=D)
    ho.unproject_append(torch.from_numpy(depth).to(D), None, cams, small, c3, 0.05, 70.0, seed=11)
    assert int(c3.item()) == 100


def test_raster_vs_oracle_maze(hip):
    verts, faces = _maze(seed=3)
    H, W = 64, 114
    poses = [([-12.0, 3.3, -9.0], [0.0, 30.0]), ([6.0, 3.3, 3.0], [20.0, 170.0]), ([15.0, 10.0, -21.0], [-45.0, 260.0])]
    RT = [ocam.camera_RT(x, v) for x, v in poses]
    cams = ho.cams12(np.stack([r for r, _ in RT]), np.stack([t for _, t in RT]), D)
    z, ov = ho.raster_zbuf(torch.from_numpy(verts).to(D), torch.from_numpy(faces).to(D), cams, H, W, bin_cap=4096)
    assert int(ov.item()) == 0
    z = z.cpu().numpy()
    for i, (R, T) in enumerate(RT):
        want = orast.raster_zbuf(verts, faces, R, T, H, W, ocam.TAN_HALF_FOV)
        assert np.array_equal(z[i], want), (i, (z[i] != want).mean())       # identical algebra, identical op order
        assert np.array_equal(want, csim.raster_zbuf(verts, faces, R, T, H, W, ocam.TAN_HALF_FOV))   # C twin == numpy
        assert (z[i] > 0).mean() > 0.9


def test_raster_closed_form_box_full_res(hip):
    """256x456 (the reference's image size): closed room, camera at the centre -> closed-form depth, no cracks."""
    h = 4.0
    v = np.array([[x, y, zz] for x in (-h, h) for y in (-h, h) for zz in (-h, h)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], np.int32)
    H, W = 256, 456
    R, T = ocam.camera_RT([0, 0, 0], [0.0, 0.0])
    z, ov = ho.raster_zbuf(torch.from_numpy(v).to(D), torch.from_numpy(f).to(D), ho.cams12(R[None], T[None], D), H, W)
    z = z[0].cpu().numpy()
    t = float(ocam.TAN_HALF_FOV)
    col, row = np.meshgrid(np.arange(W), np.arange(H))
    dx, dy = (W - (2 * col + 1)) / 256 * t, (H - (2 * row + 1)) / 256 * t
    want = h / np.maximum(np.maximum(np.abs(dx), np.abs(dy)), 1.0)
    assert (z > 0).all() and np.allclose(z, want, rtol=2e-5)
    # render -> un-project round trip: every un-projected pixel lies on the box surface
    depth = torch.from_numpy(z[None]).to(D)
    cloud = torch.zeros(H * W, 3, device=D)
    cnt = torch.zeros(1, dtype=torch.int64, device=D)
    ho.unproject_append(depth, None, ho.cams12(R[None], T[None], D), cloud, cnt, 1.0, 70.0, seed=1)
    assert int(cnt.item()) == H * W
    p = cloud.cpu().numpy()
    # NOTE the reference un-projects with NDC tables (mu:2270-2279) whose pixel pitch differs slightly from
    # the rasteriser's pixel centres, so points land on the walls only to within that skew
    assert np.abs(np.abs(p).max(1) - h).max() < 0.05


def test_mesh_ray_queries_vs_oracle(hip):
    verts, faces = _maze(seed=4)
    rng = np.random.default_rng(6)
    vd, fd = torch.from_numpy(verts).to(D), torch.from_numpy(faces).to(D)
    p0 = rng.uniform(-28, 28, (64, 3)).astype(np.float32); p0[:, 1] = rng.uniform(1, 20, 64)
    p1 = p0 + rng.normal(0, 4, (64, 3)).astype(np.float32)
    p1[0] = p0[0]                                                       # zero-length segment
    segs = np.concatenate([p0, p1], 1)
    got = ho.segments_hit_mesh(vd, fd, torch.from_numpy(segs).to(D)).cpu().numpy().astype(bool)
    want = np.array([mesh_rays.segment_hits_mesh(a, b, verts, faces) for a, b in zip(p0, p1)])
    assert np.array_equal(got, want) and 0 < want.sum() < 64
    cnt = ho.axis_ray_counts(vd, fd, torch.from_numpy(p0).to(D)).cpu().numpy()
    assert np.array_equal(cnt, np.array([mesh_rays.axis_ray_counts(p, verts, faces) for p in p0]))


def test_fusion_scoring_edges_vs_reference_golden(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "replan.npz"))
    S = 256
    maps6 = torch.zeros(6, S, S, device=D)
    maps6[0] = torch.from_numpy(g["full"].astype(np.float32))
    maps6[5] = torch.from_numpy(g["band"].astype(np.float32))
    out2 = torch.from_numpy(g["out2"]).to(D).contiguous()
    traj = torch.from_numpy(g["traj"].astype(np.float32)).to(D)
    obst, fullproj = ho.fuse_obstacle(out2, maps6, traj)
    assert np.array_equal(obst.cpu().numpy(), g["obst"].astype(np.float32))
    assert np.array_equal(fullproj.cpu().numpy(), g["fullproj"].astype(np.float32))
    pos = torch.from_numpy(g["pos"]).to(D)
    out1 = torch.from_numpy(g["out1"][0]).to(D).contiguous()
    skip = torch.from_numpy(g["skip"].astype(np.uint8)).to(D)
    valid, cell, score = ho.score_candidates(pos, g["pose"], out1, fullproj, skip)
    ids = np.nonzero(valid.cpu().numpy())[0]
    assert np.array_equal(ids, g["cand"][:, 0])
    assert np.array_equal(cell.cpu().numpy()[ids], g["cand"][:, 1:3])
    assert np.array_equal(score.cpu().numpy()[ids], g["cand_score"])        # float64, exact
    # every lattice edge in one launch == the reference's per-edge Python loop (via the pinned oracle)
    idx = g["idx"]
    nodes = {tuple(r): n for n, r in enumerate(idx.tolist())}
    edges = [(n, nodes[nb]) for n, (i, j, k) in enumerate(idx.tolist())
             for nb in ((i + 1, j, k), (i - 1, j, k), (i, j, k + 1), (i, j, k - 1)) if nb in nodes]
    ed = torch.tensor(edges, dtype=torch.int32, device=D)
    got = ho.edges_blocked(obst, g["pose"], pos, ed).cpu().numpy().astype(bool)
    layout = g["obst"].astype(np.float32)
    want = np.array([opl.edge_blocked(g["pos"][a], g["pos"][b], g["pose"], layout) for a, b in edges])
    assert np.array_equal(got, want) and 0 < want.sum() < len(want)


def test_edges_blocked_vs_reference_golden(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "planner.npz"))
    p = np.concatenate([g["edge_p1"], g["edge_p2"]], 0)
    n = len(g["edge_p1"])
    ed = torch.tensor([[i, n + i] for i in range(n)], dtype=torch.int32, device=D)
    obst = torch.from_numpy(g["layout"][0, 0].astype(np.float32)).to(D)
    got = ho.edges_blocked(obst, g["edge_pose"], torch.from_numpy(p).to(D), ed).cpu().numpy().astype(bool)
    assert got.tolist() == g["edge_blocked"].tolist()


def test_coverage_vs_oracle_and_reference(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "planner.npz"))
    gt, pc = g["cov_gt"], g["cov_pc"]
    gtd, pcd = torch.from_numpy(gt).to(D), torch.from_numpy(pc).to(D)
    out = ho.coverage_count(gtd, pcd, seed=21).cpu().numpy()
    frac, cnt = opl.coverage(gt, pc, seed=21)
    assert out[1] == 2 * len(gt) and out[0] == cnt                          # same subset, same count
    # the reference's own subset (its randperm is in the fixture): no resampling when len == 2G
    sub = torch.from_numpy(pc[g["cov_perm"]]).to(D)
    o2 = ho.coverage_count(gtd, sub, seed=0).cpu().numpy()
    assert abs(o2[0] / len(gt) - float(g["cov"])) <= 2.0 / len(gt)
    # device-side N, small cloud (no subsampling), empty cloud
    nd = torch.tensor([1000], dtype=torch.int64, device=D)
    o3 = ho.coverage_count(gtd, pcd, n_dev=nd).cpu().numpy()
    assert o3[1] == 1000 and o3[0] == opl.coverage(gt, pc[:1000])[1]
    nd.zero_()
    o4 = ho.coverage_count(gtd, pcd, n_dev=nd).cpu().numpy()
    assert o4[0] == 0 and o4[1] == 0


def test_coverage_full_size_properties(hip):
    """G = 50k vs a 3 M cloud (BASELINE sizes): monotone in the cloud, 100 % on itself, deterministic."""
    rng = np.random.default_rng(8)
    gt = torch.from_numpy(rng.uniform(-30, 30, (50_000, 3)).astype(np.float32)).to(D)
    pc = torch.from_numpy(rng.uniform(-30, 30, (3_000_000, 3)).astype(np.float32)).to(D)
    a = ho.coverage_count(gt, pc, seed=1).cpu().numpy()
    b = ho.coverage_count(gt, pc, seed=1).cpu().numpy()
    assert np.array_equal(a, b) and a[1] == 100_000
    self_cov = ho.coverage_count(gt, gt, seed=1).cpu().numpy()
    assert self_cov[0] == 50_000
    half = ho.coverage_count(gt, pc[:50_000], seed=1).cpu().numpy()
    assert half[0] <= a[0] <= 50_000


def test_carving_vs_oracle(hip):
    """A20: fused frustum test + bilinear depth lookup + carving counters, two consecutive frames."""
    H, W, zfar = 64, 114, 750.0
    rng = np.random.default_rng(9)
    P = 20000
    pts = rng.uniform(-40, 40, (P, 3)).astype(np.float32)
    ptd = torch.from_numpy(pts).to(D)
    st_o = [np.zeros(P, np.float32), np.zeros(P, np.float32), np.ones(P, np.float32), np.ones(P, np.float32)]
    st_d = [torch.from_numpy(a.copy()).to(D) for a in st_o]
    n_in = 0
    for k, (x, v) in enumerate([([2.0, 3.3, -5.0], [0.0, 30.0]), ([5.0, 3.3, -5.0], [-20.0, 120.0])]):
        depth = rng.uniform(4, 50, (H, W)).astype(np.float32)
        depth[rng.random((H, W)) < 0.15] = -1
        R, T = ocam.camera_RT(x, v)
        inf, _ = ocam.carve_update(pts, depth, None, R, T, zfar, 70.0, 10.0, 0.95, *st_o)
        n_in += int(inf.sum())
        ho.carve_update(ptd, torch.from_numpy(depth).to(D), None, np.concatenate([R.reshape(-1), T]), zfar, 70.0, 10.0, 0.95,
                        *st_d)
    assert n_in > 500
    for a, b in zip(st_o, st_d):
        assert np.array_equal(a, b.cpu().numpy())


def test_carving_vs_reference_golden(hip, golden_dir):
    """nbp_carve_update_f32 against the state the REFERENCE's own get_points_in_fov / get_signed_distance_to_depth_maps /
    update_proxy_* left after three views (tests/golden/carve.npz; the PyTorch3D projection is the fixture's one stub)."""
    from test_oracle_planner_sim import _carve_golden_sequence
    dev = {}

    def step(k, g, H, W, state):
        if not dev:
            dev["pts"] = torch.from_numpy(g["pts"]).to(D)
            dev["st"] = [torch.from_numpy(a.copy()).to(D) for a in state]
        ho.carve_update(dev["pts"], torch.from_numpy(g[f"depth{k}"]).to(D), torch.from_numpy(g[f"mask{k}"].astype(np.uint8)).to(D),
                        np.concatenate([g[f"R{k}"].reshape(-1), g[f"T{k}"]]), float(g["zfar"]), float(g[f"fov_range{k}"]),
                        float(g["tol"]), float(g["score_threshold"]), *dev["st"])
        st = [a.cpu().numpy() for a in dev["st"]]
        # the kernel keeps no signed distances: the oracle's (bit-exact twin of the kernel, test_carving_vs_oracle) stand in for
        # the distance check, the counters / occupancy below are the kernel's own
        inf, sd = ocam.carve_update(g["pts"], g[f"depth{k}"], g[f"mask{k}"], g[f"R{k}"], g[f"T{k}"], float(g["zfar"]),
                                    float(g[f"fov_range{k}"]), float(g["tol"]), float(g["score_threshold"]),
                                    *[a.copy() for a in state])
        return inf, sd, st
    _carve_golden_sequence(golden_dir, step)


def test_edge_arguments_never_crash(hip):
    """Degenerate / hostile arguments to the simulator, map and planner entry points: an error code or a sane result,
    never a crash (each call is followed by a device synchronisation so that a faulting kernel would surface here)."""
    import numpy as np
    from nextbestpath_amd import _lib
    from nextbestpath_amd.utility import hipops
    from nextbestpath_amd.utility import utils as hu
    dev = "cuda"
    sync = torch.cuda.synchronize
    # one degenerate triangle lying in the slicing plane, one zero-area triangle
    verts = torch.tensor([[0, 1, 0], [1, 1, 0], [0, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2]], dtype=torch.float32, device=dev)
    faces = torch.tensor([[0, 1, 2], [3, 4, 5]], dtype=torch.int32, device=dev)
    lab = hipops.slice_obstacle(verts, faces, 1.0, 0.0, 0.0); sync()
    assert float(lab.sum()) == 0.0                      # vertices ON the plane count as the positive side: no crossing
    cams = np.concatenate([np.eye(3, dtype=np.float32).reshape(1, 9), np.zeros((1, 3), np.float32)], 1)
    z1, _ = hipops.raster_zbuf(verts, faces, cams, 16, 24, bin_cap=1); sync()      # the legacy capacity argument is ignored
    z, ovf = hipops.raster_zbuf(verts, faces, cams, 16, 24, bin_cap=64); sync()
    assert torch.equal(z, z1)
    assert z.shape == (1, 16, 24) and torch.isfinite(z).all() and int(ovf) == 0
    seg = torch.tensor([[0, 0, 0, 0, 0, 0], [0, 0, 0, 5, 5, 5]], dtype=torch.float32, device=dev)   # zero-length segment
    hit = hipops.segments_hit_mesh(verts, faces, seg); sync()
    assert hit.shape == (2,)
    cnt = hipops.axis_ray_counts(verts, faces, seg[:, :3].contiguous()); sync()
    assert cnt.shape == (2, 3)
    # coverage against an EMPTY cloud and a cloud far outside the GT bounding box
    gt = torch.rand(100, 3, device=dev)
    cloud = torch.zeros(64, 3, device=dev)
    n0 = torch.zeros(1, dtype=torch.int64, device=dev)
    out = hipops.coverage_count(gt, cloud, n_dev=n0, n=64); sync()
    assert int(out[0]) == 0
    far = torch.full((64, 3), 1e6, device=dev)
    out = hipops.coverage_count(gt, far, n_dev=torch.tensor([64], device=dev), n=64); sync()
    assert int(out[0]) == 0
    # maps: S = 1, points at +-inf / NaN are dropped, not scattered out of bounds
    bad = torch.tensor([[float("nan"), 0, 0], [float("inf"), 1, 2], [-float("inf"), 1, 2], [0.1, 1.0, 0.1]], device=dev)
    m = hu.accumulate_step_maps(bad, torch.zeros(5), torch.tensor([0.0, 0.5, 1.5, 2.5, 3.5]), 8, (-1, 1)); sync()
    assert torch.isfinite(m).all() and float(m[:5].sum()) == 1.0
    one = hu.map_points_to_n_imgs(torch.zeros(1, 3, 2, device=dev), (1, 1), (-1, 1)); sync()
    assert one.shape == (1, 1, 1)
    # forward: unsupported size and undersized workspace are refused
    L = _lib.lib()
    x = torch.zeros(1, 5, 24, 24, device=dev)
    o1 = torch.zeros(1, 8, 6, 6, device=dev); o2 = torch.zeros(1, 1, 24, 24, device=dev)
    ws = torch.zeros(1024, dtype=torch.uint8, device=dev)
    assert L.nbp_forward_workspace_bytes(1, 24) == 0
    from nextbestpath_amd.networks import packing
    from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
    pk = packing.pack_state_dict(make_nbp_state_dict(9), torch.device(dev))
    assert L.nbp_forward_f32(pk.handle, x.data_ptr(), 1, 24, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(), None) == -3
    x = torch.zeros(1, 5, 32, 32, device=dev)
    assert L.nbp_forward_f32(pk.handle, x.data_ptr(), 1, 32, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(), None) == -2
    assert L.nbp_forward_f32(pk.handle, x.data_ptr(), 0, 32, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(), None) == -1
    sync()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_raster_random_triangle_soup_vs_oracle(hip, seed):
    """Random triangle soups around random cameras: faces crossing the clip plane, behind the camera, degenerate
    (zero area), needle-thin and screen-filling ones.  Same criterion as the maze test (silhouette pixels may differ
    through rounding order)."""
    rng = np.random.default_rng(seed)
    n = 300
    c = rng.uniform(-6, 6, (n, 1, 3))
    size = rng.choice([0.05, 0.5, 3.0, 30.0], (n, 1, 1), p=[0.2, 0.4, 0.3, 0.1])
    tri = (c + rng.normal(0, 1, (n, 3, 3)) * size).astype(np.float32)
    tri[:5, 2] = tri[:5, 1]                                   # degenerate faces
    tri[5:10, :, 1] = tri[5:10, :1, 1]                        # faces in a plane through ... (flat in y)
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    H, W = 48, 80
    poses = [(rng.uniform(-3, 3, 3).tolist(), [float(rng.uniform(-60, 60)), float(rng.uniform(0, 360))]) for _ in range(4)]
    RT = [ocam.camera_RT(x, v) for x, v in poses]
    cams = ho.cams12(np.stack([r for r, _ in RT]), np.stack([t for _, t in RT]), D)
    z, ov = ho.raster_zbuf(torch.from_numpy(verts).to(D), torch.from_numpy(faces).to(D), cams, H, W, bin_cap=4096)
    assert int(ov.item()) == 0
    z = z.cpu().numpy()
    assert np.isfinite(z).all()
    for i, (R, T) in enumerate(RT):
        want = orast.raster_zbuf(verts, faces, R, T, H, W, ocam.TAN_HALF_FOV)
        same = np.isclose(z[i], want, rtol=1e-5, atol=1e-5)
        assert same.mean() > 0.995, (seed, i, same.mean())
        assert ((z[i] > 0) == (want > 0)).mean() > 0.995


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15])
def test_coverage_random_clouds_vs_oracle(hip, seed):
    """Clustered / sparse / lattice-aligned clouds (distances exactly equal to the threshold, points on cell borders and
    outside the GT bounding box): the grid query counts exactly what the brute-force restatement counts."""
    rng = np.random.default_rng(seed)
    G = int(rng.integers(200, 3000))
    kind = seed % 3
    if kind == 0:        # integer lattice: many distances are exactly 1.0 (not covered: d < 1 is strict)
        gt = rng.integers(-8, 8, (G, 3)).astype(np.float32)
        pc = rng.integers(-9, 9, (int(rng.integers(50, 2 * G)), 3)).astype(np.float32)
    elif kind == 1:      # clusters
        centres = rng.uniform(-20, 20, (12, 3))
        gt = (centres[rng.integers(0, 12, G)] + rng.normal(0, 1.5, (G, 3))).astype(np.float32)
        pc = (centres[rng.integers(0, 6, 2 * G - 7)] + rng.normal(0, 1.0, (2 * G - 7, 3))).astype(np.float32)
    else:                # sparse cloud partly outside the GT box
        gt = rng.uniform(-5, 5, (G, 3)).astype(np.float32)
        pc = rng.uniform(-15, 15, (int(rng.integers(10, G)), 3)).astype(np.float32)
    assert len(pc) <= 2 * G                                           # no subsampling: exact comparison
    out = ho.coverage_count(torch.from_numpy(gt).to(D), torch.from_numpy(pc).to(D), seed=seed).cpu().numpy()
    _, cnt = opl.coverage(gt, pc, seed=seed)
    assert out[1] == len(pc) and out[0] == cnt, (out, cnt)
    # the rollout's variant (GT sorted once, one mark kernel per call, epoch stamps): same count on repeated calls,
    # on a growing prefix of the cloud (device-side size) and on a sub-sampled one
    plan = ho.CoveragePlan(torch.from_numpy(gt).to(D), 1.0, 2)
    pcd = torch.from_numpy(pc).to(D)
    res = torch.zeros(2, dtype=torch.int32, device=D)
    for _ in range(2):
        assert plan.count(pcd, res, seed=seed).cpu().tolist() == [cnt, len(pc)]
    half = len(pc) // 2
    nd = torch.tensor([half], dtype=torch.int64, device=D)
    assert plan.count(pcd, res, n_dev=nd, seed=seed).cpu().tolist() == [opl.coverage(gt, pc[:half], seed=seed)[1], half]
    big = np.concatenate([pc, pc + np.float32(0.25), pc - np.float32(0.5)], 0)[:2 * G + 50]
    if len(big) > 2 * G:
        want = opl.coverage(gt, big, seed=seed + 1)[1]
        assert plan.count(torch.from_numpy(big).to(D), res, seed=seed + 1).cpu().tolist() == [want, 2 * G]
        assert ho.coverage_count(torch.from_numpy(gt).to(D), torch.from_numpy(big).to(D), seed=seed + 1).cpu().tolist() == [want, 2 * G]


@pytest.mark.parametrize("thr", [1.0, 0.7, 1.37])
def test_coverage_threshold_boundary_is_the_square_root_one(hip, thr):
    """The planned kernel tests d^2 <= (largest float whose square root is below the threshold) instead of sqrt(d^2) < thr:
    GT points whose distance from a cloud point straddles the threshold ulp by ulp, along an axis and along a diagonal, must be
    counted exactly as the square-root form of the oracle (and of the one-shot kernel) counts them."""
    t = np.float32(thr)
    steps = np.arange(-40, 41)
    d = np.array([t], np.float32).view(np.int32)[0] + steps
    d = d.astype(np.int32).view(np.float32)                                     # thr +- 40 ulps
    gt_axis = np.stack([d + np.float32(3.0), np.full_like(d, 2.0), np.full_like(d, -1.0)], 1)
    diag = (d / np.float32(np.sqrt(3.0))).astype(np.float32)
    gt_diag = np.stack([diag + np.float32(30.0), diag + np.float32(2.0), diag - np.float32(1.0)], 1)
    gt = np.concatenate([gt_axis, gt_diag]).astype(np.float32)
    pc = np.array([[3.0, 2.0, -1.0], [30.0, 2.0, -1.0]], np.float32)
    _, want = opl.coverage(gt, pc, threshold=thr)
    assert 0 < want < len(gt)
    gtd, pcd = torch.from_numpy(gt).to(D), torch.from_numpy(pc).to(D)
    plan = ho.CoveragePlan(gtd, thr, 2)
    res = torch.zeros(2, dtype=torch.int32, device=D)
    assert plan.count(pcd, res, seed=0).cpu().tolist() == [want, 2]
    assert ho.coverage_count(gtd, pcd, seed=0, threshold=thr).cpu().tolist() == [want, 2]


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_planner_kernels_random_vs_oracle(hip, seed):
    """Random maps, random lattice positions partly outside the window, random skip flags: candidate scoring (one wave
    per candidate, 21 x 21 window by ballot) and the all-edges Bresenham mask against the pinned restatements."""
    rng = np.random.default_rng(seed)
    S, V = 256, 64
    full = (rng.random((S, S)) < rng.choice([0.002, 0.02, 0.3])).astype(np.float32) * rng.integers(1, 4, (S, S))
    fullproj = np.minimum(full, 1).astype(np.float32)
    obst = (rng.random((S, S)) < 0.05).astype(np.float32)
    out1 = rng.normal(0, 1, (8, V, V)).astype(np.float32)
    pose = np.array([rng.uniform(-5, 5), 3.3, rng.uniform(-5, 5), 0, 0], np.float32)
    P = 400
    pos = np.stack([rng.uniform(-48, 48, P), np.full(P, 3.3), rng.uniform(-48, 48, P)], 1).astype(np.float32)
    skip = (rng.random(P) < 0.1)
    valid, cell, score = ho.score_candidates(torch.from_numpy(pos).to(D), pose, torch.from_numpy(out1).to(D),
                                             torch.from_numpy(fullproj).to(D), torch.from_numpy(skip.astype(np.uint8)).to(D))
    v_o, c_o, s_o = opl.score_candidates(pos, pose, out1, fullproj, skip)
    v_g = valid.cpu().numpy().astype(bool)
    assert np.array_equal(v_g, v_o) and 0 < v_o.sum() < P
    assert np.array_equal(cell.cpu().numpy()[v_o], c_o[v_o]) and np.array_equal(score.cpu().numpy()[v_o], s_o[v_o])
    E = 600
    ed = rng.integers(0, P, (E, 2)).astype(np.int32)
    got = ho.edges_blocked(torch.from_numpy(obst).to(D), pose, torch.from_numpy(pos).to(D),
                           torch.from_numpy(ed).to(D)).cpu().numpy().astype(bool)
    want = np.array([opl.edge_blocked(pos[a], pos[b], pose, obst) for a, b in ed])
    assert np.array_equal(got, want) and 0 < want.sum() < E
