"""GPU parity (bit-exact): HIP map accumulation vs the numpy oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from nextbestpath_amd.utility import utils as hu
from nextbestpath_amd.utility.synthetic import make_point_cloud
from oracle import maps as omaps

pytestmark = pytest.mark.gpu


def test_reference_api_functions_vs_golden(hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "maps.npz"))
    pts = torch.from_numpy(g["points"]).cuda()
    pose = torch.from_numpy(g["pose"]).cuda()
    t2d = hu.transform_points_to_n_pieces(pts, pose, "cuda")
    assert tuple(t2d.shape) == (1, pts.shape[0], 2)
    assert np.array_equal(t2d.cpu().numpy(), g["t2d"])
    img = hu.map_points_to_n_imgs(t2d, (256, 256), (-40, 40), "cuda")
    assert np.array_equal(img.cpu().numpy(), g["img"].astype(np.float32))
    assert np.array_equal(hu.get_point_position_in_the_img(t2d.squeeze(0)[:64], (256, 256), (-40, 40)).cpu().numpy(),
                          g["pos256"])
    assert np.array_equal(hu.get_point_position_in_the_img(t2d.squeeze(0)[:64], (64, 64), (-40, 40)).cpu().numpy(),
                          g["pos64"])
    p1 = hu.get_point_position_in_the_img(t2d.squeeze(0)[5:6].squeeze(0), (256, 256), (-40, 40))
    assert tuple(p1.shape) == (2,) and np.array_equal(p1.cpu().numpy(), g["pos1"])
    t2 = torch.stack([t2d[0, :5000], t2d[0, 5000:10000]])
    assert np.array_equal(hu.map_points_to_n_imgs(t2, (128, 128), (-40, 40), "cuda").cpu().numpy(),
                          g["img2"].astype(np.float32))


@pytest.mark.parametrize("tag", ["nominal", "six_bins"])
def test_fused_accumulate_vs_golden(hip, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "maps.npz"))
    out = hu.accumulate_step_maps(torch.from_numpy(g["points"]).cuda(), torch.from_numpy(g["pose"]).cuda(),
                                  torch.from_numpy(g[f"ybins_{tag}"]), 256).cpu().numpy()
    assert np.array_equal(out[:4], g[f"slabs_{tag}"].astype(np.float32))
    assert np.array_equal(out[:5].sum(0), g["img"][0].astype(np.float32))
    assert np.array_equal(out[5], g["band"][0].astype(np.float32))


@pytest.mark.parametrize("N", [0, 1, 3, 5, 1000, 250_003])
def test_fused_accumulate_vs_oracle_ragged(hip, N):
    pc = make_point_cloud(N, seed=N + 1)
    pose = np.array([2.5, 13.3, -4.0, 0, 0], np.float32)
    ybins = np.arange(0.5, 29.5 + 7.25, 7.25, dtype=np.float32)
    want = omaps.accumulate_step_maps(pc.numpy(), pose, ybins, S=256)
    got = hu.accumulate_step_maps(pc.cuda(), torch.from_numpy(pose), torch.from_numpy(ybins), 256).cpu().numpy()
    assert np.array_equal(got, want)


def test_full_size_properties(hip):
    """3 M points, 512^2 grid (config 5 scale): conservation + linearity (size independent)."""
    N = 3_000_000
    pc = make_point_cloud(N, seed=7, extent=70.0).cuda()
    pose = torch.tensor([0.0, 13.3, 0.0, 0, 0])
    ybins = torch.arange(0.5, 29.5 + 7.25, 7.25)
    full = hu.accumulate_step_maps(pc, pose, ybins, 512, (-80, 80))
    # every point lands in exactly one of channels 0..4 or outside the window
    t2d = hu.transform_points_to_n_pieces(pc, pose)
    one = hu.map_points_to_n_imgs(t2d, (512, 512), (-80, 80))
    assert torch.equal(full[:5].sum(0), one[0])
    assert full[:5].sum().item() <= N and full[:5].sum().item() > 0.5 * N
    # linearity: maps of two halves add up to the map of the whole
    a = hu.accumulate_step_maps(pc[: N // 2], pose, ybins, 512, (-80, 80))
    b = hu.accumulate_step_maps(pc[N // 2:], pose, ybins, 512, (-80, 80))
    assert torch.equal(a + b, full)
    # idempotence / determinism
    assert torch.equal(hu.accumulate_step_maps(pc, pose, ybins, 512, (-80, 80)), full)
