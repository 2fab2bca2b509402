"""CPU: the oracle restatements reproduce the golden vectors captured from the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import maps as omaps
from oracle import nbp_net


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_state_dict_matches_reference_layout(nbp_weights, golden_dir):
    from nextbestpath_amd.networks.nbp_model import NBP
    with torch.device("meta"):
        net = NBP()
    assert len(net.state_dict()) == 327
    assert sum(p.numel() for p in net.parameters()) == 49_964_669
    g = _load(golden_dir, "nbp_fwd_S32.npz")
    for k, s in zip(g["probe_keys"], g["probe_sums"]):
        assert abs(float(nbp_weights[str(k)].double().sum()) - float(s)) < 1e-9, "synthetic weight stream drifted"


def test_network_oracle_vs_reference(nbp_weights, golden_dir):
    for tag in ("S32", "S64B2"):
        g = _load(golden_dir, f"nbp_fwd_{tag}.npz")
        with torch.no_grad():
            o1, o2 = nbp_net.nbp_forward(nbp_weights, torch.from_numpy(g["x"]))
        assert o1.shape == g["out1"].shape and o2.shape == g["out2"].shape
        # same ATen ops, same weights: only thread-count summation order may differ
        assert np.abs(o1.numpy() - g["out1"]).max() < 2e-5
        assert np.abs(o2.numpy() - g["out2"]).max() < 2e-6


def test_config1_cpu_plumbing_128(nbp_weights, golden_dir):
    """BASELINE.json configs[0]: one NBP forward on a 128x128 map on CPU (oracle only)."""
    g = _load(golden_dir, "nbp_fwd_S128.npz")
    with torch.no_grad():
        o1, o2 = nbp_net.nbp_forward(nbp_weights, torch.from_numpy(g["x"]))
    assert tuple(o1.shape) == (1, 8, 32, 32) and tuple(o2.shape) == (1, 1, 128, 128)
    assert np.abs(o1.numpy() - g["out1"]).max() < 2e-5
    assert np.array_equal(o1.numpy().reshape(8, -1).argmax(1), g["out1"].reshape(8, -1).argmax(1))


def test_maps_oracle_vs_reference(golden_dir):
    g = _load(golden_dir, "maps.npz")
    pts, pose = g["points"], g["pose"]
    t2d = omaps.transform_points_to_n_pieces(pts, pose)
    assert np.array_equal(t2d, g["t2d"])                       # bit-exact fp32
    img = omaps.map_points_to_n_imgs(t2d, (256, 256), (-40, 40))
    assert np.array_equal(img, g["img"].astype(np.float32))
    assert np.array_equal(omaps.get_point_position_in_the_img(t2d[0, :64], (256, 256), (-40, 40)), g["pos256"])
    assert np.array_equal(omaps.get_point_position_in_the_img(t2d[0, :64], (64, 64), (-40, 40)), g["pos64"])
    assert np.array_equal(omaps.get_point_position_in_the_img(t2d[0, 5], (256, 256), (-40, 40)), g["pos1"])
    t2 = np.stack([t2d[0, :5000], t2d[0, 5000:10000]])
    assert np.array_equal(omaps.map_points_to_n_imgs(t2, (128, 128), (-40, 40)), g["img2"].astype(np.float32))


def test_fused_accumulate_oracle_vs_reference(golden_dir):
    g = _load(golden_dir, "maps.npz")
    pts, pose = g["points"], g["pose"]
    for tag in ("nominal", "six_bins"):
        out = omaps.accumulate_step_maps(pts, pose, g[f"ybins_{tag}"], S=256)
        assert np.array_equal(out[:4], g[f"slabs_{tag}"].astype(np.float32)), tag
        assert np.array_equal(out[:5].sum(0), g["img"][0].astype(np.float32)), tag
        assert np.array_equal(out[5], g["band"][0].astype(np.float32)), tag
    assert len(g["ybins_six_bins"]) in (5, 6)


def test_maps_edge_cases():
    pose = np.array([0, 0, 0, 0, 0], np.float32)
    empty = omaps.accumulate_step_maps(np.zeros((0, 3), np.float32), pose, [0, 1, 2, 3, 4], S=16)
    assert empty.shape == (6, 16, 16) and empty.sum() == 0
    # +40 maps to cell S (dropped), -40 maps to cell 0 (kept): asymmetric half-width border cells
    p = np.array([[40.0, 0.5, 0.0], [-40.0, 0.5, 0.0]], np.float32)
    out = omaps.accumulate_step_maps(p, pose, [0, 1, 2, 3, 4], S=16)
    assert out[:5].sum() == 1


@pytest.mark.parametrize("tag", ["S32B2", "S128B4"])
def test_training_oracle_vs_reference(nbp_weights, golden_dir, tag):
    """Train-mode forward, NBP.loss and parameter gradients of the REFERENCE module (tests/golden/make_golden.py::
    gen_training) reproduced by the oracle's functional restatement + autograd (small ill-conditioned and
    well-conditioned fixture)."""
    g = _load(golden_dir, f"nbp_train_{tag}.npz")
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
          for k, v in nbp_weights.items()}
    x, coords = torch.from_numpy(g["x"]), torch.from_numpy(g["coords"])
    o1, o2 = nbp_net.nbp_forward(sd, x, train=True)
    assert np.abs(o1.detach().numpy() - g["out1"]).max() < 5e-5 * max(1.0, float(np.abs(g["out1"]).max()))
    assert np.abs(o2.detach().numpy() - g["out2"]).max() < 5e-6
    pred = o1[coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3]]
    loss = nbp_net.nbp_loss(sd["log_vars"], pred, torch.from_numpy(g["gains"]), o2, torch.from_numpy(g["gt"]))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    for k in g["grad_keys"]:
        k = str(k)
        kk = k.replace(".", "__")
        stride, gsum, gabs = g[kk + "__stats"]
        got = sd[k].grad.double().flatten()
        # same ops, same weights; the thread-count summation order and one ReLU / BatchNorm conditioning effect apart
        assert abs(float(got.abs().sum()) - gabs) < 2e-3 * gabs + 1e-9, k
        ref = g[kk]
        err = np.abs(got[::int(stride)].numpy() - ref).max()
        assert err < 2e-3 * max(np.abs(ref).max(), 1e-6), (k, err)


def test_camera_tables_and_sampling_logic_vs_reference(golden_dir):
    """VERDICT r04 missing 2: the pure-torch pieces of rows A13 / A14 as the reference itself computes them
    (tests/golden/camera.npz, generated by make_golden.py::gen_camera from the imported reference): Camera.__init__'s NDC tables
    and pose lattice (macarons_utils.py:2270-2279, 2283-2327), compute_partial_point_cloud's mask and keep count (:2811-2838, the
    PyTorch3D un-projection replaced by the identity), obtain_depth's depth / mask outputs and numpy draws
    (long_term_utils.py:50-155)."""
    from oracle import camera as ocam
    g = _load(golden_dir, "camera.npz")
    H, W = g["ndc_x"].shape
    nx, ny = ocam.ndc_tables(H, W)
    assert nx.dtype == np.float32 and np.array_equal(nx, g["ndc_x"]) and np.array_equal(ny, g["ndc_y"])      # bit for bit
    assert np.array_equal(np.array([nx[-1, -1], nx[0, 0], ny[-1, -1], ny[0, 0]], np.float32), g["ndc_minmax"])
    # the pose lattice: same order as the reference's dict (string keys of the index rows), same fp32 values
    L, Wd, Hd, E, A = (int(v) for v in g["dims"])
    idx, poses = ocam.pose_lattice(g["x_min_in"], L, Wd, Hd, E, A)
    assert [str(list(r)) for r in idx] == [str(k) for k in g["pose_keys"]]
    assert np.array_equal(poses, g["poses"])
    assert np.array_equal(g["cam_x_min"], g["x_min_in"] + 3)            # (self.x_min is shifted; the poses are offset from the argument)
    # compute_partial_point_cloud: which pixels are candidates, how many are kept
    depth, mask = g["depth"], g["mask"]
    for tag in "abc":
        gf, fr = float(g[f"ppc_{tag}_args"][0]), float(g[f"ppc_{tag}_args"][1])
        fov_range = np.inf if fr < 0 else fr
        ref = g[f"ppc_{tag}"]                                          # rows (ndc_x, ndc_y, depth) of the kept pixels
        pts, n_valid = ocam.partial_point_cloud(depth, mask, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), gf, fov_range, seed=1)
        assert len(pts) == len(ref) == int(n_valid * gf)
        valid = (mask != 0) & (depth < np.float32(fov_range))
        assert int(valid.sum()) == n_valid
        rows = np.stack([nx[valid], ny[valid], depth[valid]], 1)
        have = {r.tobytes() for r in rows}
        got = [r.tobytes() for r in np.ascontiguousarray(ref)]
        assert len(set(got)) == len(got) and all(r in have for r in got)      # a duplicate-free subset of exactly the candidates
    # obtain_depth with perfect depth: depth = clamp(zbuf, znear, zfar), the mask is the input mask (zbuf > -1), and the two
    # numpy draws are made only under data augmentation (the rollout's config has none): the product derives the mask in its
    # un-projection kernel and draws nothing
    zb = g["od_zbuf"]
    zn, zf = g["od_znear_zfar"]
    for tag, n in (("", 0), ("_aug", 2)):
        assert int(g[f"od_draws{tag}"]) == n
        assert np.array_equal(g[f"od_mask{tag}"], zb > -1)
        assert np.array_equal(g[f"od_depth{tag}"], np.clip(zb, zn, zf))
    m_none = ocam.partial_point_cloud(zb[0, :, :, 0], None, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), 1.0, np.inf, seed=1)[1]
    assert m_none == int((zb[0] > -1).sum())
