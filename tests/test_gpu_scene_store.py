"""GPU parity of the device Scene / Cell store (csrc/nbp_scene.hip) against the reference-generated fixture
tests/golden/scene.npz and the numpy restatement oracle/scene_store.py (row A18 + the store the MACARONS baseline
drivers fill every step)."""
import os

import numpy as np
import pytest
import torch

from oracle import scene_store as oss

pytestmark = pytest.mark.gpu
D = "cuda"


def _rows(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])]


def _scene(g, capacity, resolution, seed=0):
    from nextbestpath_amd.simulator.scene import Scene
    return Scene(g["x_min"], g["x_max"], 3, 1, 3, capacity, resolution, 900, torch.device(D), seed=seed)


def test_fill_cells_vs_reference_fixture(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "scene.npz"))
    sc = _scene(g, 2000, 0.5)
    assert sc.cell_keys() == [tuple(k) for k in g["cell_keys"].tolist()]
    gt = torch.from_numpy(g["gt"]).to(D)
    sc.fill_cells(gt)
    off = 0
    for key, n in zip(sc.cell_keys(), g["first_n"].tolist()):          # first fill: no thinning, every point kept
        assert np.array_equal(_rows(sc.cell_points(key).cpu().numpy()), _rows(g["first_pts"][off:off + n]))
        off += n
    entire = sc.return_entire_pt_cloud().cpu().numpy()
    assert np.array_equal(_rows(entire), _rows(g["entire_first"]))
    sc.fill_cells(torch.from_numpy(g["extra"]).to(D))                  # second fill: fp64 thinning at the resolution
    off = 0
    for key, n in zip(sc.cell_keys(), g["second_n"].tolist()):
        assert np.array_equal(_rows(sc.cell_points(key).cpu().numpy()), _rows(g["second_pts"][off:off + n]))
        off += n
    # an (n_dev-bounded) third fill of the same points adds nothing: each is at distance 0 from itself
    before = sc.cell_count.clone()
    sc.fill_cells(gt, n_dev=torch.tensor([1000], dtype=torch.int64, device=D))
    assert torch.equal(before, sc.cell_count)


def test_capacity_cap_vs_oracle_and_reference_counts(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "scene.npz"))
    sc = _scene(g, 150, 0.5, seed=3)
    sc.fill_cells(torch.from_numpy(g["gt"]).to(D))
    assert sc.cell_count.cpu().tolist() == g["cap_n"].tolist()          # min(capacity, points of the cell)
    ora = oss.Scene(g["x_min"], g["x_max"], 3, 1, 3, 150, 0.5)
    ora.fill_cells(g["gt"], seed=3 + 7919)                               # the product's per-fill seed
    for key, cell in zip(sc.cell_keys(), ora.cells.values()):
        assert np.array_equal(_rows(sc.cell_points(key).cpu().numpy()), _rows(cell.pts))
    # a second fill on top of full cells (thin, then cap over [stored | new]) still matches the restatement
    extra = torch.from_numpy(g["extra"]).to(D)
    sc.fill_cells(extra)
    ora.fill_cells(g["extra"], seed=3 + 2 * 7919)
    for key, cell in zip(sc.cell_keys(), ora.cells.values()):
        assert np.array_equal(_rows(sc.cell_points(key).cpu().numpy()), _rows(cell.pts))


def test_scene_coverage_vs_oracle(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "scene.npz"))
    rng = np.random.default_rng(4)
    gt_s, rec_s = _scene(g, 2000, 0.5), _scene(g, 2000, None)
    o_gt, o_rec = oss.Scene(g["x_min"], g["x_max"], 3, 1, 3, 2000, 0.5), oss.Scene(g["x_min"], g["x_max"], 3, 1, 3, 2000, None)
    assert abs(rec_s.cell_resolution - o_rec.cells[(0, 0, 0)].resolution) < 1e-12
    gt_s.fill_cells(torch.from_numpy(g["gt"]).to(D))
    o_gt.fill_cells(g["gt"])
    assert gt_s.scene_coverage(rec_s, 1.0) == (0.0, int(gt_s.cell_count.sum()))      # empty reconstruction
    for k in range(3):                                                   # a growing partial reconstruction
        part = (g["gt"][rng.choice(len(g["gt"]), 400, replace=False)] + rng.normal(0, 0.2, (400, 3))).astype(np.float32)
        rec_s.fill_cells(torch.from_numpy(part).to(D))
        o_rec.fill_cells(part)
        assert rec_s.cell_count.cpu().tolist() == [len(c.pts) for c in o_rec.cells.values()]
        for eps in (0.3, 1.0):
            covered, n_gt = oss.scene_coverage(o_gt, o_rec, eps)
            cov, n = gt_s.scene_coverage(rec_s, eps)
            assert n == n_gt and abs(cov - covered / n_gt) < 1e-15, (k, eps, cov, covered / n_gt)
    assert 0.5 < cov <= 1.0


def test_gt_surface_pipeline_keeps_every_sample(hip, tmp_path):
    """Row A18 semantics: the GT cloud of a rollout = ALL sampled surface points strictly inside a cell (no thinning on
    the first fill), capped per cell only by surface_cell_capacity."""
    from nextbestpath_amd.simulator import scene as sc
    from nextbestpath_amd.simulator.mesh import make_maze_scene
    from nextbestpath_amd.testers import nbp_planning as tp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    make_maze_scene(str(tmp_path / "m"), seed=0, cells=8, size=4.8, height=1.2, tess=0.3)
    params = tp.load_params(os.path.join(root, "configs/macarons/macarons_default_training_config.json"))
    ds = sc.SceneDataset(str(tmp_path), ["m"])
    settings = sc.Settings(ds[0]["settings"], params.scene_scale_factor)
    mesh = sc.load_scene(os.path.join(str(tmp_path), "m", ds[0]["obj_name"]), params.scene_scale_factor, torch.device(D))
    gt_scene, gt = sc.setup_gt_scene(params, settings, mesh, torch.device(D), 0.05, seed=2)
    assert gt_scene.cell_capacity == params.surface_cell_capacity and abs(gt_scene.cell_resolution - 0.5) < 1e-12
    pts = sc.sample_gt_surface(mesh.verts_host, mesh.faces_host, params.n_gt_surface_points, gt_scene.x_min,
                               gt_scene.x_max, seed=2)
    ora = oss.Scene(gt_scene.x_min, gt_scene.x_max, gt_scene.grid_l, gt_scene.grid_w, gt_scene.grid_h,
                    params.surface_cell_capacity, 0.5)
    ora.fill_cells(pts, seed=2 + 7919)
    assert np.array_equal(_rows(gt.cpu().numpy()), _rows(ora.return_entire_pt_cloud()))
    assert len(gt) > 0.95 * params.n_gt_surface_points                   # ~50 k, not a 0.5-voxel thinned ~35 k


def test_view_state_kernel_vs_reference_golden(hip, golden_dir):
    """nbp_view_state_update_f32 against the reference's own compute_view_state / update_proxy_view_states outputs
    (tests/golden/viewstate.npz).  The bins come from asin / acos of this platform; a ray within 1e-5 rad of a bin's rounding
    boundary may land on the other side than under torch's libm -- such rays are excluded by name, everything else is exact."""
    import os
    from nextbestpath_amd.utility import hipops as ho
    from oracle import view_state as ovs
    g = np.load(os.path.join(golden_dir, "viewstate.npz"))
    for key, views, ne, na, pts in (("vs", g["views"], 7, 14, g["pts"]), ("vs1", g["views"][:1], 7, 14, g["pts"]),
                                    ("vs_small", g["views"][:2], 4, 6, g["pts"][:500])):
        vs = torch.zeros(len(pts), ne * na, device=D)
        ho.view_state_update(torch.from_numpy(pts).to(D), views, ne, na, vs)
        got = vs.cpu().numpy().astype(np.uint8)
        safe = ovs.boundary_distance(pts, views, ne, na).min(1) > 1e-5
        assert safe.mean() > 0.98, key               # (the fixture's eight degenerate rays sit ON boundaries by construction)
        assert np.array_equal(got[safe], g[key][safe]), key
        assert (got[~safe].sum(1) >= 1).all()
    vs = torch.zeros(len(g["proxy"]), 98, device=D)
    proxy = torch.from_numpy(g["proxy"]).to(D)
    for k in range(2):
        ho.view_state_update(proxy, g["upd_cam"][k], 7, 14, vs, mask=torch.from_numpy(g["upd_mask"][k].astype(np.uint8)).to(D),
                             sd=torch.from_numpy(g["upd_sd"][k]).to(D), distance_to_surface=3 * float(g["dist_between"]))
        safe = ovs.boundary_distance(g["proxy"], g["upd_cam"][:k + 1], 7, 14).min(1) > 1e-5
        assert np.array_equal(vs.cpu().numpy().astype(np.uint8)[safe], g["upd_state"][k][safe]), k


def test_update_proxy_view_states_takes_the_reference_masked_distances(hip, golden_dir):
    """Scene.update_proxy_view_states with signed distances as the REFERENCE passes them (one per point inside the mask,
    macarons_utils.py:3299-3302) gives what the full-size form gives; any other size raises (it used to be read out
    of bounds)."""
    import os
    import types
    from nextbestpath_amd.simulator.scene import Scene
    g = np.load(os.path.join(golden_dir, "viewstate.npz"))
    proxy = torch.from_numpy(g["proxy"]).to(D)
    P = len(proxy)

    def scene():
        return types.SimpleNamespace(proxy_points=proxy, view_states=torch.zeros(P, 98, device=D), view_state_n_elev=7,
                                     view_state_n_azim=14, distance_between_proxy_points=float(g["dist_between"]))
    mask = torch.from_numpy(g["upd_mask"][0].astype(bool)).to(D)
    sd_full = torch.from_numpy(g["upd_sd"][0]).to(D)
    cam = types.SimpleNamespace(X_cam=torch.from_numpy(g["upd_cam"][0]))
    a, b = scene(), scene()
    Scene.update_proxy_view_states(a, cam, mask, sd_full)
    Scene.update_proxy_view_states(b, cam, mask, sd_full[mask])                  # the reference's calling convention
    assert torch.equal(a.view_states, b.view_states) and float(a.view_states.sum()) > 0
    with pytest.raises(ValueError):
        Scene.update_proxy_view_states(scene(), cam, mask, sd_full[:7])
    with pytest.raises(ValueError):
        Scene.update_proxy_view_states(scene(), cam, mask[:-1], sd_full)
