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


@pytest.mark.parametrize("N,n_old,n_fresh,S", [(0, 0, 1, 256), (1000, 0, 8, 256), (250_003, 37, 4, 256), (250_003, 300, 0, 256),
                                               (50_000, 2000, 5, 512)])
def test_step_maps_is_the_separate_calls(hip, golden_dir, N, n_old, n_fresh, S):
    """nbp_step_maps_f32 (six maps + trajectory channel + network input in one call) against the reference-API functions it
    replaces in the step loop, themselves pinned to the golden vectors above; the fresh trajectory points are appended to the
    device history by the same launch; a device-resident cloud size is honoured; nothing outside the outputs is touched."""
    gr = (-40 * S // 256, 40 * S // 256)
    pc = make_point_cloud(max(N, 1), seed=N + 3, extent=35.0 * S / 256).cuda()
    pose = np.array([2.5, 13.3, -4.0, 0, 0], np.float32)
    ybins = torch.arange(0.5, 29.5 + 7.25, 7.25)
    rng = np.random.default_rng(N + n_old)
    traj = (rng.standard_normal((n_old + n_fresh, 3)) * 25 * S / 256).astype(np.float32)
    traj[:, 1] = 1.5
    if n_old + n_fresh > 3:
        traj[-1] = traj[0]                                   # a revisited position counts twice
        traj[1] = [1e4, 0, 0]                                # off the map: dropped
    cap = n_old + n_fresh + 3
    traj_dev = torch.full((cap, 3), 7.0, device="cuda")
    traj_dev[:n_old] = torch.from_numpy(traj[:n_old]).cuda()
    n_dev = torch.tensor([N], dtype=torch.int64, device="cuda")
    guard = torch.full((2, 5 * S * S + 2), -3.0, device="cuda")
    net_in = guard[0, 1:-1].view(5, S, S)
    net_in.fill_(9.0)
    out6 = torch.full((6, S, S), 5.0, device="cuda")
    hu.step_maps(pc, pose, ybins, S, gr, traj_dev, n_old, traj[n_old:], out6, net_in, n_dev=n_dev)
    want6 = hu.accumulate_step_maps(pc[:N] if N else pc[:0], torch.from_numpy(pose), ybins, S, gr)
    assert torch.equal(out6, want6)
    assert torch.equal(net_in[:4], want6[:4])
    if n_old + n_fresh:
        t2d = hu.transform_points_to_n_pieces(torch.from_numpy(traj).cuda(), pose)
        want_t = hu.map_points_to_n_imgs(t2d, (S, S), gr)[0]
    else:
        want_t = torch.zeros(S, S, device="cuda")
    assert torch.equal(net_in[4], want_t)
    assert torch.equal(traj_dev[:n_old + n_fresh].cpu(), torch.from_numpy(traj))
    assert bool((traj_dev[n_old + n_fresh:] == 7.0).all())
    assert bool((guard[0, 0] == -3.0) & (guard[0, -1] == -3.0) & (guard[1] == -3.0).all())
    with pytest.raises(ValueError):
        hu.step_maps(pc, pose, ybins, S, gr, traj_dev[:1], 1, traj[:1], out6, net_in)
    with pytest.raises(Exception):
        hu.step_maps(pc, pose, ybins, S, gr, traj_dev, 0, np.zeros((9, 3), np.float32), out6, net_in)


# ---- the tile-binned shadow copy of the cloud (round 4): same maps, bit for bit, whatever the append pattern
def _wall_cloud(n, seed, extent=30.0, y_range=(0.0, 12.0)):
    """Points on a few vertical walls (what a rollout's cloud looks like: thousands of points per top-down cell)."""
    rng = np.random.default_rng(seed)
    k = 12
    a = rng.uniform(-extent, extent, (k, 2)).astype(np.float32)
    d = rng.uniform(-1, 1, (k, 2)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    w = rng.integers(0, k, n)
    s = rng.uniform(0, 25, n).astype(np.float32)
    xz = a[w] + d[w] * s[:, None] + rng.normal(0, 0.02, (n, 2)).astype(np.float32)
    y = rng.uniform(*y_range, n).astype(np.float32)
    return torch.from_numpy(np.stack([xz[:, 0], y, xz[:, 1]], 1).astype(np.float32))


@pytest.mark.parametrize("S", [256, 512])
def test_binned_maps_equal_the_append_order_kernel(hip, S):
    """A cloud that grows in ragged chunks (as a rollout's does), maps built after every chunk from a moving pose: the binned build
    (CloudBins: pages per 2.5-unit tile, dense LDS histograms) equals map_accumulate_kernel bit for bit; every point is filed once."""
    gr = (-40 * S // 256, 40 * S // 256)
    cap = 400_000
    pts = _wall_cloud(cap, seed=S, extent=35.0)
    cloud = torch.zeros(cap, 3, device="cuda")
    n_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    bins = hu.CloudBins((-62.0, -62.0), (62.0, 62.0), cap, "cuda")
    ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
    n = 0
    rng = np.random.default_rng(5)
    for step, add in enumerate([1, 63, 64, 0, 5836, 29_180, 130_000, 17, 120_000, 60_000]):
        cloud[n:n + add] = pts[n:n + add].cuda()
        n += add
        n_dev.fill_(n)
        pose = np.array([rng.uniform(-20, 20), 6.0 + 0.01 * step, rng.uniform(-20, 20), 0, 0], np.float32)
        want = hu.accumulate_step_maps(cloud, torch.from_numpy(pose), ybins, S, gr, n_dev=n_dev)
        got = hu.accumulate_step_maps(cloud, torch.from_numpy(pose), ybins, S, gr, n_dev=n_dev, bins=bins)
        assert torch.equal(got, want), (step, n)
        h = bins.header()
        assert h["n_binned"] == n and h["error"] == 0 and h["n_overflow"] == 0, (step, h)
    # a second build from the same state files nothing and gives the same maps (idempotent)
    again = hu.accumulate_step_maps(cloud, torch.from_numpy(pose), ybins, S, gr, n_dev=n_dev, bins=bins)
    assert torch.equal(again, want) and bins.header()["n_binned"] == n
    # after reset() the store follows a new cloud from zero points
    bins.reset()
    n_dev.fill_(1000)
    got = hu.accumulate_step_maps(cloud, torch.from_numpy(pose), ybins, S, gr, n_dev=n_dev, bins=bins)
    assert torch.equal(got, hu.accumulate_step_maps(cloud[:1000], torch.from_numpy(pose), ybins, S, gr))
    assert bins.header()["n_binned"] == 1000


def test_binned_maps_points_that_cannot_be_filed_are_still_counted(hip):
    """Points outside the tile grid and NaN points go to the side list and are counted with direct atomics; a tile may own any
    number of pages (a 150 k-point blob on one spot: 70+ pages in a tile or two); a side list that overflows breaks the store,
    which then counts the whole cloud directly: the maps equal the append-order kernel's in every case."""
    S, gr = 256, (-40, 40)
    rng = np.random.default_rng(11)
    inside = _wall_cloud(60_000, seed=2, extent=10.0)
    blob = torch.from_numpy(np.concatenate([rng.normal(3.0, 0.3, (150_000, 1)), rng.uniform(0, 12, (150_000, 1)),
                                            rng.normal(-2.0, 0.3, (150_000, 1))], 1).astype(np.float32))
    far = torch.from_numpy(rng.uniform(-39, 39, (5_000, 3)).astype(np.float32) * np.array([1, 0.1, 1], np.float32) + np.array([0, 5, 0], np.float32))
    nan = torch.full((7, 3), float("nan"))
    pc = torch.cat([inside, far, blob, nan, inside[:100]]).cuda().contiguous()
    bins = hu.CloudBins((-12.0, -12.0), (12.0, 12.0), pc.shape[0], "cuda")          # `far` mostly lies outside this grid
    ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
    pose = torch.tensor([1.0, 5.05, -1.0, 0, 0])
    want = hu.accumulate_step_maps(pc, pose, ybins, S, gr)
    got = hu.accumulate_step_maps(pc, pose, ybins, S, gr, bins=bins)
    assert torch.equal(got, want)
    h = bins.header()
    assert h["error"] == 0 and h["n_binned"] == pc.shape[0]
    assert 1000 < h["n_overflow"] < 65536 and h["n_pages"] >= 150_000 // 2048     # the outside points only; the blob is paged
    # ... and again from another pose (the side list is walked on every build)
    pose2 = torch.tensor([-8.0, 5.05, 9.0, 0, 0])
    assert torch.equal(hu.accumulate_step_maps(pc, pose2, ybins, S, gr, bins=bins), hu.accumulate_step_maps(pc, pose2, ybins, S, gr))
    # a grid that misses the scene: 215 k points on a 65536-entry side list -> the store breaks, the maps stay right
    tiny = hu.CloudBins((500.0, 500.0), (501.0, 501.0), pc.shape[0], "cuda")
    n_dev = torch.tensor([pc.shape[0] - 5000], dtype=torch.int64, device="cuda")
    for k, pz in enumerate((pose, pose2, pose)):
        if k == 2:
            n_dev.fill_(pc.shape[0])                    # new points arrive after the break
        got = hu.accumulate_step_maps(pc, pz, ybins, S, gr, n_dev=n_dev, bins=tiny)
        assert torch.equal(got, hu.accumulate_step_maps(pc, pz, ybins, S, gr, n_dev=n_dev)), k
    assert tiny.header()["error"] == 1


def test_binned_step_maps_and_batch_equal_the_unbinned_calls(hip):
    """nbp_step_maps_binned_f32 / _batch_f32 (trajectory channel, network input, group form with a low page bound) == nbp_step_maps_f32."""
    S, gr = 256, (-40, 40)
    ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
    R = 5
    clouds = [_wall_cloud(40_000 + 7_000 * r, seed=20 + r, extent=25.0).cuda().contiguous() for r in range(R)]
    n_devs = [torch.tensor([c.shape[0] - 11 * r], dtype=torch.int64, device="cuda") for r, c in enumerate(clouds)]
    binss = [hu.CloudBins((-52.0, -52.0), (52.0, 52.0), c.shape[0], "cuda") for c in clouds]
    rng = np.random.default_rng(3)
    for rnd in range(2):                       # second round: everything already filed
        poses = [np.array([rng.uniform(-10, 10), 5.0, rng.uniform(-10, 10), 0, 0], np.float32) for _ in range(R)]
        trajs = [(rng.standard_normal((3 + r, 3)) * 10).astype(np.float32) for r in range(R)]
        items_a, items_b = [], []
        for r in range(R):
            ta, tb = torch.zeros(64, 3, device="cuda"), torch.zeros(64, 3, device="cuda")
            ta[:2] = tb[:2] = torch.from_numpy(trajs[r][:2]).cuda()
            items_a.append((clouds[r], clouds[r].shape[0], n_devs[r], poses[r], ybins, ta, 2, trajs[r][2:]))
            items_b.append((clouds[r], 100, n_devs[r], poses[r], ybins, tb, 2, trajs[r][2:], binss[r]))      # n_upper far too low on purpose
        o6a, nia = torch.zeros(R, 6, S, S, device="cuda"), torch.zeros(R, 5, S, S, device="cuda")
        o6b, nib = torch.full((R, 6, S, S), 3.0, device="cuda"), torch.full((R, 5, S, S), 4.0, device="cuda")
        hu.step_maps_batch(items_a, S, gr, o6a, nia)
        hu.step_maps_batch(items_b, S, gr, o6b, nib)
        assert torch.equal(o6a, o6b) and torch.equal(nia, nib), rnd
        for r in range(R):
            assert torch.equal(items_a[r][5], items_b[r][5])                    # trajectory history appended alike
            assert binss[r].header()["n_binned"] == int(n_devs[r].item())
        # single-call form
        o6c, nic = torch.zeros(6, S, S, device="cuda"), torch.zeros(5, S, S, device="cuda")
        tc = torch.zeros(64, 3, device="cuda")
        tc[:2] = torch.from_numpy(trajs[0][:2]).cuda()
        hu.step_maps(clouds[0], poses[0], ybins, S, gr, tc, 2, trajs[0][2:], o6c, nic, n_dev=n_devs[0], bins=binss[0])
        assert torch.equal(o6c, o6a[0]) and torch.equal(nic, nia[0])


# ---- round 6: the un-projection launch files the points it appends (and clears the maps), the build is the page launch alone
def _frames(rng, F_, H, W):
    from oracle import camera as ocam
    from nextbestpath_amd.utility import hipops as ho
    depth = rng.uniform(0.6, 60, (F_, H, W)).astype(np.float32)
    depth[rng.random((F_, H, W)) < 0.2] = -1
    poses = [([float(rng.uniform(-8, 8)), 3.3, float(rng.uniform(-8, 8))], [float(rng.uniform(-30, 30)), float(rng.uniform(0, 360))])
             for _ in range(F_)]
    RT = [ocam.camera_RT(x, v) for x, v in poses]
    return torch.from_numpy(depth).cuda(), ho.cams12(np.stack([r for r, _ in RT]), np.stack([t for _, t in RT]), "cuda")


@pytest.mark.parametrize("colours", [False, True])
def test_filing_unprojection_and_one_launch_build_equal_the_two_launch_build(hip, colours):
    """Frames appended through unproject_append(bins=, clear=): the cloud (and its colours) are those of the plain call bit for bit,
    the store stays in step with the cloud (n_binned == cloud size, nothing on the side list), and the ONE-launch build
    (prefiled) gives the append-order kernel's maps; stale maps / trajectory channel are cleared by the filing launch."""
    from nextbestpath_amd.utility import hipops as ho
    S, gr, H, W = 256, (-40, 40), 128, 228
    rng = np.random.default_rng(3)
    cap = 300_000
    cloud, ref = torch.zeros(cap, 3, device="cuda"), torch.zeros(cap, 3, device="cuda")
    rgbc, rgbr = torch.zeros(cap, 3, device="cuda"), torch.zeros(cap, 3, device="cuda")
    cnt, cnt_ref = torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    bins = hu.CloudBins((-160.0, -160.0), (160.0, 160.0), cap, "cuda")
    ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
    maps6 = torch.full((6, S, S), 7.0, device="cuda")           # stale contents: the filing launch clears them
    traj = torch.full((S, S), 5.0, device="cuda")
    for step, F_ in enumerate([1, 4, 1, 4, 4, 1]):
        depth, cams = _frames(rng, F_, H, W)
        rgb = torch.rand(F_, H, W, 3, device="cuda") if colours else None
        kw = dict(rgb=rgb, cloud_rgb=rgbc) if colours else {}
        kw_ref = dict(rgb=rgb, cloud_rgb=rgbr) if colours else {}
        assert ho.unproject_files(depth, None)
        clear = (maps6, traj) if F_ == 1 else None             # (as Rollout.pre: the frame in front of the build clears)
        ho.unproject_append(depth, None, cams, cloud, cnt, 0.2, 70.0, seed=100 + step, bins=bins, clear=clear, **kw)
        ho.unproject_append(depth, None, cams, ref, cnt_ref, 0.2, 70.0, seed=100 + step, **kw_ref)
        n = int(cnt.item())
        assert n == int(cnt_ref.item()) and torch.equal(cloud[:n], ref[:n])
        if colours:
            assert torch.equal(rgbc[:n], rgbr[:n])
        h = bins.header()
        assert h["n_binned"] == n and h["error"] == 0 and h["n_overflow"] == 0, (step, h)
        if clear is None:
            continue
        assert float(traj.abs().sum()) == 0.0
        traj.fill_(5.0)
        pose = torch.tensor([float(rng.uniform(-10, 10)), 3.3, float(rng.uniform(-10, 10)), 0.0, 0.0])
        got = hu.accumulate_step_maps(cloud, pose, ybins, S, gr, n_dev=cnt, out=maps6, bins=bins, prefiled=True)
        want = hu.accumulate_step_maps(ref, pose, ybins, S, gr, n_dev=cnt_ref)
        assert torch.equal(got, want), step
        assert float(want.sum()) > 0
        maps6.fill_(7.0)
    assert n > 50_000


def test_one_launch_build_counts_points_no_launch_filed(hip):
    """A frame appended by the plain call puts the store out of step: the filing call behind it refuses to file (n_binned stays),
    and the one-launch build still counts every point (the unfiled tail directly); a two-launch build catches the store up and
    filing resumes."""
    from nextbestpath_amd.utility import hipops as ho
    S, gr, H, W = 256, (-40, 40), 64, 116
    rng = np.random.default_rng(9)
    cap = 100_000
    cloud = torch.zeros(cap, 3, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    bins = hu.CloudBins((-160.0, -160.0), (160.0, 160.0), cap, "cuda")
    ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
    maps6 = torch.zeros(6, S, S, device="cuda")
    pose = torch.tensor([1.0, 3.3, -2.0, 0.0, 0.0])
    d, c = _frames(rng, 4, H, W)
    ho.unproject_append(d, None, c, cloud, cnt, 0.3, 70.0, seed=1, bins=bins)
    n1 = int(cnt.item())
    assert bins.header()["n_binned"] == n1
    d, c = _frames(rng, 4, H, W)
    ho.unproject_append(d, None, c, cloud, cnt, 0.3, 70.0, seed=2)                    # not filed
    d, c = _frames(rng, 1, H, W)
    ho.unproject_append(d, None, c, cloud, cnt, 0.3, 70.0, seed=3, bins=bins, clear=(maps6, None))   # out of step: files nothing
    n3 = int(cnt.item())
    assert n3 > n1 and bins.header()["n_binned"] == n1
    got = hu.accumulate_step_maps(cloud, pose, ybins, S, gr, n_dev=cnt, out=maps6, bins=bins, prefiled=True)
    want = hu.accumulate_step_maps(cloud, pose, ybins, S, gr, n_dev=cnt)
    assert torch.equal(got, want)
    got = hu.accumulate_step_maps(cloud, pose, ybins, S, gr, n_dev=cnt, out=maps6, bins=bins)                     # catches up
    assert torch.equal(got, want) and bins.header()["n_binned"] == n3
    d, c = _frames(rng, 1, H, W)
    ho.unproject_append(d, None, c, cloud, cnt, 0.3, 70.0, seed=4, bins=bins, clear=(maps6, None))
    n4 = int(cnt.item())
    assert bins.header()["n_binned"] == n4 > n3
    got = hu.accumulate_step_maps(cloud, pose, ybins, S, gr, n_dev=cnt, out=maps6, bins=bins, prefiled=True)
    assert torch.equal(got, hu.accumulate_step_maps(cloud, pose, ybins, S, gr, n_dev=cnt))
    # a frame the three-launch form does not take (H W % 4 != 0) cannot file: the wrapper says so and the library refuses
    odd = torch.full((1, 63, 115), 2.0, device="cuda")
    assert not ho.unproject_files(odd, None)
    with pytest.raises(Exception):
        ho.unproject_append(odd, None, c, cloud, cnt, 0.3, 70.0, seed=5, bins=bins)
