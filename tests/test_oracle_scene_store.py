"""CPU: the scene-store restatement against tests/golden/scene.npz (outputs of the reference's own Scene / Cell /
surface-sampling functions) and the product's setup-time sampler against the same fixture."""
import os

import numpy as np

from oracle import camera as ocam
from oracle import scene_store as oss


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "scene.npz"))


def _rows(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])]


def test_face_areas_and_surface_sampling_vs_reference(golden_dir):
    g = _g(golden_dir)
    assert np.allclose(oss.face_areas(g["verts"], g["faces"]), g["areas"], rtol=2e-6, atol=0)
    inside, fin = oss.faces_inside(g["verts"], g["faces"], g["x_min"], g["x_max"])
    assert np.array_equal(inside, g["inside"])
    pts, _ = oss.sample_surface(g["verts"], fin, g["u_face"], g["u_alpha"], g["u_beta"])
    # the recorded uniform stream replayed: same faces, same barycentric points (fp32 cumsum order may move a pick
    # across a face boundary when u lands within 1 ulp of the cumulative value -- none of the 3000 draws does)
    assert np.allclose(pts, g["gt"], rtol=0, atol=2e-6), np.abs(pts - g["gt"]).max()


def test_product_surface_sampler_vs_reference(golden_dir):
    """simulator/scene.py::sample_gt_surface (setup-time host code of the product) replaying the reference's stream:
    bit-identical points."""
    from nextbestpath_amd.simulator import scene as sc
    g = _g(golden_dir)
    pts = sc.sample_gt_surface(g["verts"], g["faces"], 3000, g["x_min"], g["x_max"],
                               uniforms=(g["u_face"], g["u_alpha"], g["u_beta"]))
    assert np.array_equal(pts, g["gt"])
    assert np.allclose(sc.face_areas(g["verts"], g["faces"]), g["areas"], rtol=2e-6, atol=0)
    again = sc.sample_gt_surface(g["verts"], g["faces"], 3000, g["x_min"], g["x_max"], seed=5)
    assert np.array_equal(again, sc.sample_gt_surface(g["verts"], g["faces"], 3000, g["x_min"], g["x_max"], seed=5))


def test_cell_assignment_and_fill_vs_reference(golden_dir):
    g = _g(golden_dir)
    sc = oss.Scene(g["x_min"], g["x_max"], 3, 1, 3, 2000, 0.5)
    assert np.array_equal(sc.cells_for_each_pt(g["gt"]), g["cells_of"])
    assert [list(k) for k in sc.cells] == g["cell_keys"].tolist()          # same cell iteration order
    # first fill: empty cells keep EVERY point strictly inside them (no thinning, below capacity)
    sc.fill_cells(g["gt"])
    off = 0
    for cell, n in zip(sc.cells.values(), g["first_n"].tolist()):
        assert len(cell.pts) == n
        assert np.array_equal(_rows(cell.pts), _rows(g["first_pts"][off:off + n]))
        off += n
    assert np.array_equal(_rows(sc.return_entire_pt_cloud()), _rows(g["entire_first"]))
    # second fill: new points further than the resolution (fp64 distances) from the cell's points are added
    sc.fill_cells(g["extra"])
    off = 0
    for cell, n in zip(sc.cells.values(), g["second_n"].tolist()):
        assert np.array_equal(_rows(cell.pts), _rows(g["second_pts"][off:off + n]))
        off += n


def test_capacity_cap_and_derived_parameters_vs_reference(golden_dir):
    g = _g(golden_dir)
    sc = oss.Scene(g["x_min"], g["x_max"], 3, 1, 3, 150, 0.5)
    sc.fill_cells(g["gt"], seed=3)
    full = oss.Scene(g["x_min"], g["x_max"], 3, 1, 3, 10**6, 0.5)
    full.fill_cells(g["gt"])
    for cell, ref_n, everything in zip(sc.cells.values(), g["cap_n"].tolist(), full.cells.values()):
        assert len(cell.pts) == ref_n                                       # min(capacity, points in the cell)
        have = {tuple(r) for r in everything.pts.tolist()}
        assert all(tuple(r) in have for r in cell.pts.tolist())             # a subset, no duplicates
        assert len({tuple(r) for r in cell.pts.tolist()}) == len(cell.pts)
    c = oss.Cell([0.0, 3.0, 0.0], np.float32(6.0), np.float32(6.4), np.float32(6.0), None, 0.5)
    assert c.capacity == int(g["auto_capacity"])
    c2 = oss.Cell([0.0, 3.0, 0.0], np.float32(6.0), np.float32(6.4), np.float32(6.0), 500, None)
    assert abs(c2.resolution - float(g["auto_resolution"])) < 1e-6


def test_cartesian_coords_vs_reference(golden_dir):
    """get_cartesian_coords (macarons/utility/CustomGeometry.py:5-24) as get_camera_RT feeds it (mu:949-952): the
    reference evaluates in fp32, the restatement (and the product's camera_RT) in fp64."""
    from nextbestpath_amd.simulator.camera import _cartesian
    g = _g(golden_dir)
    for e, a, want in zip(g["cart_elev"], g["cart_azim"], g["cart_rays"]):
        assert np.allclose(-ocam.cartesian(1.0, -float(e), 180.0 + float(a)), want, atol=3e-7)
        assert np.allclose(-_cartesian(-float(e), 180.0 + float(a)), want, atol=3e-7)


def test_view_state_restatement_vs_reference_golden(golden_dir):
    """compute_view_state (scone_utils.py:799-862) and Scene.update_proxy_view_states (macarons_utils.py:3268-3327): the numpy
    restatement reproduces the reference's own outputs -- 4000 points x 6 cameras incl. the degenerate directions, two grid
    sizes, two accumulating updates with signed-distance selection."""
    import os
    from oracle import view_state as ovs
    g = np.load(os.path.join(golden_dir, "viewstate.npz"))
    for key, views, ne, na, pts in (("vs", g["views"], 7, 14, g["pts"]), ("vs1", g["views"][:1], 7, 14, g["pts"]),
                                    ("vs_small", g["views"][:2], 4, 6, g["pts"][:500])):
        got = ovs.compute_view_state(pts, views, ne, na).astype(np.uint8)
        assert np.array_equal(got, g[key]), key
        assert got.sum(1).min() >= 1 and got.sum(1).max() <= len(views)
    vs = np.zeros((len(g["proxy"]), 98), np.float32)
    for k in range(2):
        upd = ovs.update_proxy_view_states(vs, g["proxy"], g["upd_mask"][k], g["upd_sd"][k], g["upd_cam"][k], 7, 14,
                                           3 * float(g["dist_between"]))
        assert np.array_equal(vs.astype(np.uint8), g["upd_state"][k]), k
        assert 0 < upd.sum() < g["upd_mask"][k].sum()                  # the signed-distance test removed some points
    assert (g["upd_state"][1].sum(1) == 2).sum() > 100                  # vectors accumulate over the two cameras
