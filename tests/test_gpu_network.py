"""GPU parity: the HIP network path vs the torch-fp32 oracle and the reference's golden vectors.
Tolerance 1e-4 (BASELINE.json north_star); argmax goal cells must be identical."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_helpers import conv_igemm, nchw, nhwc, pack_conv, stream
from nextbestpath_amd import _lib
from oracle import nbp_net

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


# (B, H, W, C0, C1, N, ksize, ups, split_k, tile)
CONV_CASES = [
    (1, 16, 16, 64, 0, 128, 3, 0, 1, 1),     # 128x128 tile
    (1, 16, 16, 64, 0, 128, 3, 0, 4, 1),     # split-K
    (1, 16, 16, 64, 0, 128, 3, 0, 0, 0),     # auto plan
    (2, 12, 20, 32, 0, 64, 3, 0, 1, 2),      # 256x64 tile, ragged M (480), W != H
    (1, 24, 24, 64, 0, 32, 1, 0, 1, 3),      # 256x32 tile, 1x1
    (1, 16, 16, 96, 0, 64, 3, 0, 1, 4),      # 128x64 tile, C not a power of two
    (3, 4, 4, 128, 0, 256, 3, 0, 1, 5),      # 64x128 tile, tiny M (48)
    (1, 8, 8, 64, 0, 128, 3, 1, 1, 0),       # fused x2 nearest upsample -> 16x16
    (1, 16, 16, 64, 64, 128, 3, 0, 2, 0),    # fused concat + split-K
    (2, 8, 8, 128, 128, 64, 1, 0, 1, 0),     # attention-style 1x1 over [g|x]
    (1, 2, 2, 1024, 0, 1024, 3, 0, 0, 0),    # bottleneck shape at S=32 (M=4, K=9216)
    (1, 1, 1, 512, 0, 1024, 3, 0, 0, 0),     # S=16 bottleneck: 1x1 image, all taps but centre OOB
    (1, 16, 32, 64, 0, 128, 3, 0, 0, 6),     # halo-tile kernel, BN = 128: image borders on every side
    (2, 8, 64, 96, 0, 64, 3, 0, 0, 7),       # halo-tile kernel, BN = 64, three chunks, two images
    (1, 8, 16, 64, 0, 128, 3, 1, 0, 6),      # halo + fused x2 upsample (16 x 32 output)
    (1, 16, 32, 32, 64, 256, 3, 0, 0, 6),    # halo + fused concat, two n blocks
    (3, 24, 96, 32, 0, 64, 3, 0, 0, 7),      # halo, 3 x 3 tiles per image: an interior tile without padding
    (1, 8, 32, 256, 0, 128, 3, 0, 4, 6),     # halo + split-K over whole chunks (8 chunks / 4)
    (1, 16, 32, 96, 64, 64, 3, 0, 2, 7),     # halo + split-K with a ragged split (5 chunks / 2) across the concat seam
    (1, 12, 32, 64, 0, 128, 3, 0, 0, 8),     # 4-row halo tiles (H % 4 == 0 only), BN = 128
    (2, 4, 64, 96, 32, 64, 3, 0, 2, 9),      # 4-row halo tiles, BN = 64, concat + split-K, image = one tile row
    (1, 4, 16, 64, 0, 128, 3, 1, 0, 8),      # 4-row halo + fused upsample (8 x 32 output)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_vs_torch(hip, case):
    B, H, W, C0, C1, N, k, ups, split_k, tile = case
    dev = "cuda"
    x0 = _rand(B, C0, H, W, seed=1)
    x1 = _rand(B, C1, H, W, seed=2) if C1 else None
    w = _rand(N, C0 + C1, k, k, seed=3, scale=(6.0 / ((C0 + C1) * k * k)) ** 0.5)
    scale = _rand(N, seed=4) * 0.2 + 1.0
    shift = _rand(N, seed=5) * 0.1
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    if ups:
        xin = F.interpolate(xin, scale_factor=2)
    ref = F.relu(F.conv2d(xin, w, None, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    wd = w.to(dev).contiguous()
    wpk = pack_conv(wd)
    x0d, x1d = nhwc(x0).to(dev), None if x1 is None else nhwc(x1).to(dev)
    scd, shd = scale.to(dev), shift.to(dev)
    out = conv_igemm(x0d, x1d, ups, wpk, N, k, scd, shd, True, split_k, tile)
    got = nchw(out).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL


def test_small_layers_vs_torch(hip):
    dev = "cuda"
    L = hip
    d = lambda t: t.to(dev).contiguous()          # keep device copies alive until the sync below
    # Conv1.conv.0: NCHW in, NHWC out
    B, H, W = 2, 20, 12
    x = torch.floor(_rand(B, 5, H, W, seed=1).abs() * 4)
    w = _rand(64, 5, 3, 3, seed=2, scale=0.3)
    sc, sh = _rand(64, seed=3) * 0.2 + 1, _rand(64, seed=4) * 0.1
    ref = F.relu(F.conv2d(x, w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    out = torch.empty(B, H, W, 64, device=dev)
    xd, wd, scd, shd = d(x), d(w), d(sc), d(sh)
    _lib.check(L.nbp_conv_first_f32(_lib.ptr(xd), B, H, W, _lib.ptr(wd), _lib.ptr(scd), _lib.ptr(shd),
                                    _lib.ptr(out), stream()), "first")
    assert (nchw(out).cpu() - ref).abs().max() < TOL
    # maxpool
    a = _rand(2, 64, 8, 12, seed=5)
    ad = d(nhwc(a))
    o = torch.empty(2, 4, 6, 64, device=dev)
    _lib.check(L.nbp_maxpool2_nhwc_f32(_lib.ptr(ad), 2, 8, 12, 64, _lib.ptr(o), stream()), "pool")
    assert torch.equal(nchw(o).cpu(), F.max_pool2d(a, 2, 2))
    # psi gate
    M, Fq, C = 37, 32, 64
    q, xs, wp = _rand(M, Fq, seed=6).abs(), _rand(M, C, seed=7), _rand(Fq, seed=8)
    st = torch.tensor([0.9, -0.2])
    ref = xs * torch.sigmoid((q @ wp) * st[0] + st[1]).unsqueeze(1)
    o = torch.empty(M, C, device=dev)
    qd, wpd, std, xsd = d(q), d(wp), d(st), d(xs)
    _lib.check(L.nbp_psi_gate_f32(_lib.ptr(qd), Fq, _lib.ptr(wpd), _lib.ptr(std), _lib.ptr(xsd), C, M, _lib.ptr(o),
                                  stream()), "gate")
    assert (o.cpu() - ref).abs().max() < 1e-5
    # final 1x1 (8 outputs linear; 1 output sigmoid)
    a = _rand(2, 256, 6, 5, seed=9)
    ad = d(nhwc(a))
    for n_out, sig in ((8, 0), (1, 1)):
        w = _rand(n_out, 256, 1, 1, seed=10, scale=0.1)
        b = _rand(n_out, seed=11)
        ref = F.conv2d(a, w, b)
        ref = torch.sigmoid(ref) if sig else ref
        o = torch.empty(2, n_out, 6, 5, device=dev)
        ones, wd, bd = torch.ones(n_out, device=dev), d(w), d(b)
        _lib.check(L.nbp_final_1x1_f32(_lib.ptr(ad), 2, 6, 5, 256, _lib.ptr(wd), n_out, _lib.ptr(ones), _lib.ptr(bd),
                                       sig, _lib.ptr(o), stream()), "final")
        assert (o.cpu() - ref).abs().max() < 1e-5
    # layout helpers round trip
    t = d(_rand(2, 7, 5, 3, seed=12))
    o = torch.empty(2, 5, 3, 7, device=dev)
    _lib.check(L.nbp_nchw_to_nhwc_f32(_lib.ptr(t), 2, 7, 5, 3, _lib.ptr(o), stream()), "to_nhwc")
    assert torch.equal(o, t.permute(0, 2, 3, 1).contiguous())
    o2 = torch.empty_like(t)
    _lib.check(L.nbp_nhwc_to_nchw_f32(_lib.ptr(o), 2, 7, 5, 3, _lib.ptr(o2), stream()), "to_nchw")
    assert torch.equal(o2, t)


def _module(nbp_weights):
    from nextbestpath_amd.networks.nbp_model import NBP
    net = NBP()
    net.load_state_dict(nbp_weights, strict=True)
    net.conv_precision = "fp32"        # this file: the fp32 MFMA pipe (tests/test_gpu_split.py: the default split path)
    return net.cuda().eval()


@pytest.fixture(scope="module")
def net(nbp_weights):
    return _module(nbp_weights)


@pytest.mark.parametrize("tag", ["S32", "S64B2", "S128"])
def test_forward_vs_reference_golden(hip, net, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"nbp_fwd_{tag}.npz"))
    with torch.no_grad():
        o1, o2 = net(torch.from_numpy(g["x"]).cuda())
    o1, o2 = o1.cpu().numpy(), o2.cpu().numpy()
    assert o1.shape == g["out1"].shape and o2.shape == g["out2"].shape
    assert np.abs(o1 - g["out1"]).max() < TOL
    assert np.abs(o2 - g["out2"]).max() < TOL
    B = o1.shape[0]
    # goal cells: argmax over the max-over-headings map (nbp_planning.py:194) must be identical
    assert np.array_equal(o1.max(1).reshape(B, -1).argmax(1), g["out1"].max(1).reshape(B, -1).argmax(1))
    assert np.array_equal(o1.reshape(B, 8, -1).argmax(2), g["out1"].reshape(B, 8, -1).argmax(2))
    assert np.array_equal(o2 >= 0.13, g["out2"] >= 0.13)      # obstacle threshold (nbp_planning.py:168)


def test_forward_256_vs_oracle(hip, net, nbp_weights):
    """BASELINE config 2 size: 256x256, B=1, against the torch-fp32 CPU oracle."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(1, 256, seed=3)
    with torch.no_grad():
        r1, r2 = nbp_net.nbp_forward(nbp_weights, x)
        o1, o2 = net(x.cuda())
    assert tuple(o1.shape) == (1, 8, 64, 64) and tuple(o2.shape) == (1, 1, 256, 256)
    assert (o1.cpu() - r1).abs().max() < TOL
    assert (o2.cpu() - r2).abs().max() < TOL
    assert torch.equal(o1.cpu().amax(1).flatten().argmax(), r1.amax(1).flatten().argmax())


def test_forward_512_vs_oracle(hip, net, nbp_weights):
    """BASELINE configs[4] grid in fp32: 512x512 against the torch-fp32 CPU oracle (every level is >= 32 pixels wide,
    so every 3x3 layer takes the halo-tile kernel)."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(1, 512, seed=17)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        r1, r2 = nbp_net.nbp_forward(nbp_weights, x)
        o1, o2 = net(x.cuda())
    assert tuple(o1.shape) == (1, 8, 128, 128) and tuple(o2.shape) == (1, 1, 512, 512)
    assert (o1.cpu() - r1).abs().max() < TOL
    assert (o2.cpu() - r2).abs().max() < TOL
    assert torch.equal(o1.cpu().amax(1).flatten().argmax(), r1.amax(1).flatten().argmax())


@pytest.mark.parametrize("B,S", [(1, 16), (3, 48), (2, 96), (1, 160), (5, 32)])
def test_forward_odd_sizes_vs_oracle(hip, net, nbp_weights, B, S):
    """Sizes whose pyramid mixes the halo-tile kernel (levels that are multiples of 8 x 32) with the implicit GEMM
    (48, 24, 12 ... wide levels), odd batch sizes, and the smallest legal map (16 -> a 1 x 1 bottleneck)."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(B, S, seed=100 + S + B)
    with torch.no_grad():
        r1, r2 = nbp_net.nbp_forward(nbp_weights, x)
        o1, o2 = net(x.cuda())
    assert (o1.cpu() - r1).abs().max() < TOL
    assert (o2.cpu() - r2).abs().max() < TOL


def test_forward_batch_consistency_and_determinism(hip, net):
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(3, 64, seed=5).cuda()
    with torch.no_grad():
        a1, a2 = net(x)
        b1, b2 = net(x)
        c1, c2 = net(x[1:2])
    assert torch.equal(a1, b1) and torch.equal(a2, b2)            # run-to-run bit identical
    assert (a1[1:2] - c1).abs().max() < 1e-5 and (a2[1:2] - c2).abs().max() < 1e-5


def test_repack_after_weight_update(hip, nbp_weights):
    net = _module(nbp_weights)
    x = torch.ones(1, 5, 32, 32, device="cuda")
    with torch.no_grad():
        a1, _ = net(x)
        net.Final1.bias.add_(1.0)
        b1, _ = net(x)
    assert (b1 - a1 - 1.0).abs().max() < 1e-5


@pytest.mark.parametrize("precision", ["fp32_split", "fp32", "bf16"])
def test_forward_graph_is_bit_identical(hip, nbp_weights, precision):
    """packing.ForwardGraph / NBP.forward_static: the eval forward on fixed buffers captured once into a hipGraph and replayed -- the
    same kernels with the same arguments in the same order, so the outputs are the eager call's bit for bit, replay after replay
    and when the input tensor's CONTENT changes; a weight update re-packs and re-captures."""
    from nextbestpath_amd.networks import packing
    from nextbestpath_amd.networks.nbp_model import NBP
    from nextbestpath_amd.utility.synthetic import make_count_maps
    m = NBP()
    m.load_state_dict(nbp_weights, strict=True)
    m = m.cuda().eval()
    m.conv_precision = precision
    for B, S in ((1, 256), (3, 64)):
        x = make_count_maps(B, S, seed=B).cuda()
        with torch.no_grad():
            e1, e2 = m(x)
            g1, g2 = m.forward_static(x)
            assert torch.equal(e1, g1) and torch.equal(e2, g2), (precision, B, S)
            x.copy_(make_count_maps(B, S, seed=B + 10).cuda())           # new content, same tensor: the replay reads it
            e1, e2 = m(x)
            for _ in range(3):
                g1, g2 = m.forward_static(x)
            assert torch.equal(e1, g1) and torch.equal(e2, g2), (precision, B, S)
    n_graphs = len(m._graphs)
    assert n_graphs == 2
    with torch.no_grad():
        m.Final1.weight.mul_(1.5)                                        # a weight update: new pack, the captured graphs are dropped
        e1, _ = m(x)
        g1, _ = m.forward_static(x)
    assert torch.equal(e1, g1) and len(m._graphs) == 1
    assert isinstance(next(iter(m._graphs.values())), packing.ForwardGraph)


def test_halo_tile_on_1x1_is_a_shape_error(hip):
    """An explicit halo tile id on a 1x1 convolution (or an image that is not a multiple of the tile) is refused with
    NBP_E_SHAPE -- regression: the plan used to divide by zero."""
    dev = "cuda"
    x = torch.zeros(1, 8, 32, 64, device=dev)
    one = torch.ones(64, device=dev)
    wpk = torch.zeros(64 * 64, device=dev)
    for tile in (6, 7, 8, 9):
        with pytest.raises(_lib.NbpHipError):
            conv_igemm(x, None, False, wpk, 64, 1, one, one, True, 0, tile)
    x2 = torch.zeros(1, 6, 32, 64, device=dev)        # H = 6: neither 8- nor 4-row tiles fit
    wpk3 = torch.zeros(9 * 64 * 64, device=dev)
    for tile in (7, 9):
        with pytest.raises(_lib.NbpHipError):
            conv_igemm(x2, None, False, wpk3, 64, 3, one, one, True, 0, tile)


def test_argument_errors(hip):
    L = hip
    assert L.nbp_forward_f32(None, None, 1, 256, None, None, None, 0, None) == -1
    z = torch.zeros(16, device="cuda")
    assert L.nbp_maxpool2_nhwc_f32(_lib.ptr(z), 1, 3, 3, 4, _lib.ptr(z), None) == -3


def test_build_then_smoke_in_one_process(hip):
    """Loading libnbp_hip.so before PyTorch has initialised HIP must still bind to PyTorch's runtime
    (regression: two libamdhip64 instances in one process -> hipErrorNoDevice on the first call)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from nextbestpath_amd import _lib\n_lib.lib()\n"
            "import __graft_entry__ as g\ng.build(); g.smoke(); print('OK')\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-800:] + out.stderr[-1500:]


def test_conv_entry_point_fuzz(hip):
    """120 seeded random (shape, tile, split) requests, valid and invalid: the library either refuses (NbpHipError) or
    returns the right convolution -- it never crashes the process and never returns garbage."""
    rng = np.random.default_rng(123)
    dev = "cuda"
    ok = refused = 0
    for trial in range(120):
        B = int(rng.integers(1, 3))
        H = int(rng.choice([1, 2, 4, 6, 8, 12, 16]))
        W = int(rng.choice([1, 4, 8, 16, 32, 64]))
        C0 = int(rng.choice([32, 64, 96]))
        C1 = int(rng.choice([0, 0, 32, 64]))
        N = int(rng.choice([32, 64, 128]))
        k = int(rng.choice([1, 3]))
        ups = int(rng.integers(0, 2)) if (H % 2 == 0 and W % 2 == 0) else 0
        tile = int(rng.integers(0, 12))          # 10, 11 do not exist
        split = int(rng.choice([0, 0, 1, 2, 3, 5]))
        Hs, Ws = (H // 2, W // 2) if ups else (H, W)
        x0 = _rand(B, C0, Hs, Ws, seed=trial)
        x1 = _rand(B, C1, Hs, Ws, seed=trial + 1000) if C1 else None
        w = _rand(N, C0 + C1, k, k, seed=trial + 2000, scale=0.05)
        sc = (_rand(N, seed=trial + 3000) * 0.2 + 1.0)
        sh = _rand(N, seed=trial + 4000) * 0.1
        wpk = pack_conv(w.to(dev).contiguous())
        try:
            out = conv_igemm(nhwc(x0).to(dev), None if x1 is None else nhwc(x1).to(dev), ups, wpk, N, k, sc.to(dev),
                             sh.to(dev), True, split, tile)
        except _lib.NbpHipError:
            refused += 1
            continue
        xin = x0 if x1 is None else torch.cat((x0, x1), 1)
        if ups:
            xin = F.interpolate(xin, scale_factor=2)
        ref = F.relu(F.conv2d(xin, w, None, padding=k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        assert (nchw(out).cpu() - ref).abs().max().item() < TOL, (trial, B, H, W, C0, C1, N, k, ups, tile, split)
        ok += 1
    assert ok >= 30 and refused >= 10, (ok, refused)
