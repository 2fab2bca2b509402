"""GPU parity of the split path: fp32 convolutions whose products run on the fp16 matrix pipe as three exact fp16 MFMAs per
product, operands scaled by per-tensor powers of two and cut into two pieces (nbp_split.hip).  It is an fp32 path -- same tensors, same 1e-4 bar against the torch-fp32 oracle and the reference's
golden vectors as nbp_forward_f32 -- and its error against fp64 must not exceed the fp32 MFMA path's (both are measured)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_helpers import conv3x3_split, conv_igemm, nchw, nhwc, pack_conv, pack_conv_split, pack_upconv_split, upconv3x3_split
from nextbestpath_amd import _lib
from oracle import nbp_net

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def test_weight_planes_sum_to_the_scaled_weight(hip):
    """hi + lo == w * 2^(14 - floor(log2 max|w|)) to 2^-23 relative (two fp16 pieces by round-to-nearest), max |w| is reported
    exactly, layout [chunk of 16 channels][tap][plane][k half][N][8]."""
    w = (_rand(64, 96, 3, 3, seed=1) * torch.logspace(-4, 0, 96).view(1, -1, 1, 1)).cuda().contiguous()
    planes, wamax = pack_conv_split(w)
    assert wamax.view(torch.float32).item() == w.abs().max().item()
    s = 2.0 ** (14 - int(np.floor(np.log2(w.abs().max().item()))))
    f = planes.view(torch.float16).view(6, 9, 2, 2, 64, 8).double()          # chunk16, tap, plane, k half, n, c
    total = f[:, :, 0] + f[:, :, 1]
    want = (w.double() * s).view(64, 6, 2, 8, 9).permute(1, 4, 2, 0, 3)       # [chunk16][tap][k half][n][c]
    assert 2 ** 14 <= float(total.abs().max()) < 2 ** 15
    err = (total - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -22 + 2.0 ** -25).all()), float((err / (want.abs() + 1e-30)).max())
    assert bool((f[:, :, 1].abs() <= f[:, :, 0].abs() * 2.0 ** -10 + 2.0 ** -24).all())        # |lo| <= half an ulp of hi


# (B, H, W, C0, C1, N, ups, split_k)
CASES = [
    (1, 16, 32, 64, 0, 128, 0, 1),      # one 16 x 32 tile: image borders on every side, two n blocks
    (2, 16, 64, 96, 0, 64, 0, 1),       # six chunks, two images, two tiles per image
    (1, 8, 16, 64, 0, 128, 1, 1),       # fused x2 nearest upsample (16 x 32 output)
    (1, 16, 32, 32, 64, 256, 0, 1),     # fused concat
    (3, 48, 96, 32, 0, 64, 0, 1),       # 3 x 3 tiles per image: an interior tile without padding
    (1, 16, 32, 256, 0, 128, 0, 4),     # split-K over whole chunks (16 chunks / 4)
    (1, 16, 32, 96, 64, 64, 0, 3),      # ragged split (10 chunks / 3) across the concat seam
    (2, 32, 32, 128, 0, 128, 0, 0),     # automatic split-K, B = 2 at 32 x 32
    (2, 16, 16, 64, 0, 128, 0, 1),      # 16 x 16 pixel tiles x 128 channels (the bottleneck level of a 256 grid)
    (1, 32, 48, 96, 32, 128, 0, 2),     # 16 x 16 tiles, 2 x 3 per image, concat + split-K
    (1, 8, 8, 512, 0, 1024, 1, 0),      # 16 x 16 tiles + fused upsample + automatic split-K over 32 chunks
]


@pytest.mark.parametrize("mag", [1.0, 300.0, 1e-3])
@pytest.mark.parametrize("case", CASES)
def test_conv3x3_split_vs_fp64_and_fp32_path(hip, case, mag):
    """mag scales the activations: the per-tensor power-of-two scale must make the result independent of magnitude."""
    B, H, W, C0, C1, N, ups, split_k = case
    if mag != 1.0 and case not in (CASES[0], CASES[3], CASES[8]):
        pytest.skip("magnitude sweep on three cases")
    dev = "cuda"
    x0 = _rand(B, C0, H, W, seed=1) * mag
    x1 = _rand(B, C1, H, W, seed=2) * mag * 0.01 if C1 else None       # the second source two decades below the first
    w = _rand(N, C0 + C1, 3, 3, seed=3, scale=(6.0 / ((C0 + C1) * 9)) ** 0.5)
    scale = _rand(N, seed=4) * 0.2 + 1.0
    shift = _rand(N, seed=5) * 0.1 * mag
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    if ups:
        xin = F.interpolate(xin, scale_factor=2)
    ref = F.relu(F.conv2d(xin.double(), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    wd = w.to(dev).contiguous()
    x0d, x1d = nhwc(x0).to(dev), None if x1 is None else nhwc(x1).to(dev)
    scd, shd = scale.to(dev), shift.to(dev)
    got = nchw(conv3x3_split(x0d, x1d, ups, pack_conv_split(wd), N, scd, shd, True, split_k)).cpu().double()
    # the fp32 pipe with ONE accumulation chain per output (its automatic split-K shortens the chains of these small shapes)
    f32 = nchw(conv_igemm(x0d, x1d, ups, pack_conv(wd), N, 3, scd, shd, True, 1, 0)).cpu().double()
    assert got.shape == ref.shape
    e_split, e_f32 = (got - ref).abs(), (f32 - ref).abs()
    assert e_split.max().item() < 4e-6 * mag and e_split.max().item() <= 3.0 * e_f32.max().item() + 1e-7 * mag, (e_split.max(), e_f32.max())
    assert e_split.pow(2).mean().sqrt().item() <= 1.2 * e_f32.pow(2).mean().sqrt().item() + 1e-9 * mag
    # max |out| reported through amax_out == the tensor's max, and feeding it back as amax_in changes nothing
    amax_out = torch.zeros(64, dtype=torch.int32, device=dev)           # 64 words per tensor; the max is the max over them
    got2 = conv3x3_split(x0d, x1d, ups, pack_conv_split(wd), N, scd, shd, True, split_k, amax_out=amax_out)
    assert amax_out.view(torch.float32).max().item() == got2.abs().max().item()
    amax_in = torch.zeros(64, dtype=torch.float32, device=dev)
    amax_in[5] = max(x0d.abs().max().item(), 0 if x1d is None else x1d.abs().max().item())
    got3 = conv3x3_split(x0d, x1d, ups, pack_conv_split(wd), N, scd, shd, True, split_k, amax_in=amax_in.view(torch.int32))
    assert torch.equal(got2, got3)


# (B, Hs, Ws, C, N, split_k): low-resolution input size
UP_CASES = [
    (1, 16, 32, 64, 64, 1),        # one low-resolution tile -> 32 x 64 output, every border
    (2, 32, 64, 96, 128, 1),       # 2 x 2 tiles, two images, two n blocks
    (1, 16, 16, 64, 128, 1),       # 16 x 16 tiles x 128 channels
    (1, 16, 32, 256, 64, 4),       # split-K
    (2, 16, 16, 1024, 512, 0),     # Up5-like: automatic split-K
]


@pytest.mark.parametrize("case", UP_CASES)
def test_upconv_parity_kernels_vs_fp64_and_fp32_path(hip, case):
    """up_conv = x2 nearest upsample + 3x3 convolution, evaluated as four 2x2 convolutions of the low-resolution input with
    pre-summed weights: against fp64 of the reference formulation (F.interpolate + conv2d), next to the fp32 pipe."""
    B, Hs, Ws, C, N, split_k = case
    dev = "cuda"
    x = _rand(B, C, Hs, Ws, seed=1)
    w = _rand(N, C, 3, 3, seed=3, scale=(6.0 / (C * 9)) ** 0.5)
    scale = _rand(N, seed=4) * 0.2 + 1.0
    shift = _rand(N, seed=5) * 0.1
    ref = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1)
                 + shift.double().view(1, -1, 1, 1))
    wd, xd, scd, shd = w.to(dev).contiguous(), nhwc(x).to(dev), scale.to(dev), shift.to(dev)
    got = nchw(upconv3x3_split(xd, pack_upconv_split(wd), N, scd, shd, True, split_k)).cpu().double()
    f32 = nchw(conv_igemm(xd, None, 1, pack_conv(wd), N, 3, scd, shd, True, 1, 0)).cpu().double()
    assert got.shape == ref.shape
    e_split, e_f32 = (got - ref).abs(), (f32 - ref).abs()
    assert e_split.max().item() < 4e-6 and e_split.max().item() <= 3.0 * e_f32.max().item() + 1e-7, (e_split.max(), e_f32.max())
    assert e_split.pow(2).mean().sqrt().item() <= 1.2 * e_f32.pow(2).mean().sqrt().item() + 1e-9


@pytest.mark.parametrize("mag", [0.0, 1e-30, 1e-12, 1e12, 1e25])
def test_split_scales_cover_the_fp32_range(hip, mag):
    """The per-tensor power-of-two scale keeps the fp16 pieces in range whatever the magnitude of the activations: all-zero
    input gives exactly relu(shift); 1e-30 .. 1e25 give the fp32 pipe's relative accuracy (fp16 alone spans 6e-8 .. 65504)."""
    dev = "cuda"
    B, H, W, C, N = 1, 16, 32, 64, 64
    x = _rand(B, C, H, W, seed=1) * mag
    w = _rand(N, C, 3, 3, seed=3, scale=0.1)
    scale = torch.ones(N)
    shift = _rand(N, seed=5) * 0.1 * mag
    ref = F.relu(F.conv2d(x.double(), w.double(), None, padding=1) + shift.double().view(1, -1, 1, 1))
    xd, wd, scd, shd = nhwc(x).to(dev), w.to(dev).contiguous(), scale.to(dev), shift.to(dev)
    got = nchw(conv3x3_split(xd, None, 0, pack_conv_split(wd), N, scd, shd, True, 1)).cpu().double()
    if mag == 0.0:
        assert torch.equal(got, ref)
        return
    f32 = nchw(conv_igemm(xd, None, 0, pack_conv(wd), N, 3, scd, shd, True, 1, 0)).cpu().double()
    assert torch.isfinite(got).all()
    e_split, e_f32 = (got - ref).abs().max().item(), (f32 - ref).abs().max().item()
    assert e_split <= 3.0 * e_f32 + 1e-7 * mag, (e_split, e_f32)


def test_split_kernel_refuses_what_it_does_not_take(hip):
    dev = "cuda"
    x = torch.zeros(1, 16, 24, 64, device=dev)                          # 24 wide: neither 32- nor 16-pixel tiles
    sc = torch.ones(64, device=dev)
    packed = (torch.zeros(4 * 9 * 4 * 64 * 8, dtype=torch.int16, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
    with pytest.raises(_lib.NbpHipError):
        conv3x3_split(x, None, 0, packed, 64, sc, sc, True)
    with pytest.raises(_lib.NbpHipError):
        conv3x3_split(torch.zeros(1, 8, 32, 64, device=dev), None, 0, packed, 64, sc, sc, True)      # 8 rows: no 16-row tile
    with pytest.raises(_lib.NbpHipError):
        conv3x3_split(torch.zeros(1, 16, 16, 64, device=dev), None, 0, packed, 64, sc, sc, True)     # 16 wide needs N % 128


def _module(nbp_weights, precision):
    from nextbestpath_amd.networks.nbp_model import NBP
    net = NBP()
    net.load_state_dict(nbp_weights, strict=True)
    net.conv_precision = precision
    return net.cuda().eval()


@pytest.fixture(scope="module")
def nets(nbp_weights):
    return _module(nbp_weights, "fp32_split"), _module(nbp_weights, "fp32")


@pytest.mark.parametrize("tag", ["S32", "S64B2", "S128"])
def test_split_forward_vs_reference_golden(hip, nets, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"nbp_fwd_{tag}.npz"))
    with torch.no_grad():
        o1, o2 = nets[0](torch.from_numpy(g["x"]).cuda())
    o1, o2 = o1.cpu().numpy(), o2.cpu().numpy()
    assert np.abs(o1 - g["out1"]).max() < TOL and np.abs(o2 - g["out2"]).max() < TOL
    B = o1.shape[0]
    assert np.array_equal(o1.max(1).reshape(B, -1).argmax(1), g["out1"].max(1).reshape(B, -1).argmax(1))
    assert np.array_equal(o1.reshape(B, 8, -1).argmax(2), g["out1"].reshape(B, 8, -1).argmax(2))
    assert np.array_equal(o2 >= 0.13, g["out2"] >= 0.13)


@pytest.mark.parametrize("B,S", [(1, 256), (4, 256), (2, 96), (3, 64), (1, 512)])
def test_split_forward_vs_oracle_and_fp32_path(hip, nets, nbp_weights, B, S):
    """Against the torch-fp32 CPU oracle at the 1e-4 bar, and next to the fp32 MFMA path against an fp64 evaluation of the
    same network: the split path's error is not larger."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(B, S, seed=40 + S + B)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        o1, o2 = nets[0](x.cuda())
        f1, f2 = nets[1](x.cuda())
        r1, r2 = nbp_net.nbp_forward(nbp_weights, x)
    assert (o1.cpu() - r1).abs().max() < TOL and (o2.cpu() - r2).abs().max() < TOL
    assert torch.equal(o1.cpu().amax(1).flatten(1).argmax(1), r1.amax(1).flatten(1).argmax(1))
    if S <= 256 and B <= 2:
        sd64 = {k: v.double() for k, v in nbp_weights.items()}
        with torch.no_grad():
            d1, d2 = nbp_net.nbp_forward(sd64, x.double())
        for o, f, d in ((o1, f1, d1), (o2, f2, d2)):
            es, ef = (o.cpu().double() - d).abs(), (f.cpu().double() - d).abs()
            assert es.max().item() <= 2.0 * ef.max().item() + 1e-7, (es.max().item(), ef.max().item())
            assert es.mean().item() <= 1.25 * ef.mean().item() + 1e-9, (es.mean().item(), ef.mean().item())


def test_split_handle_also_runs_the_fp32_forward(hip, nets, nbp_weights):
    from nextbestpath_amd.networks import packing
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(2, 64, seed=9).cuda()
    pk = packing.pack_state_dict(nbp_weights, "cuda", precision="fp32_split")
    a1, a2 = packing.forward_packed(pk, x, precision="fp32")
    with torch.no_grad():
        f1, f2 = nets[1](x)
    assert torch.equal(a1, f1) and torch.equal(a2, f2)
    pk32 = packing.pack_state_dict(nbp_weights, "cuda", precision="fp32")
    with pytest.raises(ValueError):
        packing.forward_packed(pk32, x, precision="fp32_split")


@pytest.mark.parametrize("B,S", [(1, 64), (3, 96), (2, 160), (5, 192), (1, 320), (2, 384), (1, 448), (7, 128), (16, 64)])
def test_split_forward_equals_fp32_pipe_over_sizes(hip, nets, B, S):
    """Grids whose pyramid mixes the three routes of the split path (16 x 32 tiles, 16 x 16 tiles with >= 128 channels, parity
    up-conv kernels) with the fp32-pipe fallbacks (levels that are not multiples of 16), odd batch sizes: both paths agree to
    1e-4 of the output range and pick the same goal cells."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(B, S, seed=300 + S + B).cuda()
    with torch.no_grad():
        o1, o2 = nets[0](x)
        f1, f2 = nets[1](x)
    rng = max(1.0, float(f1.abs().max()))
    assert float((o1 - f1).abs().max()) < 1e-4 * rng and float((o2 - f2).abs().max()) < 1e-4
    assert torch.equal(o1.amax(1).flatten(1).argmax(1), f1.amax(1).flatten(1).argmax(1))
    assert torch.equal(o2 >= 0.13, f2 >= 0.13)


@pytest.mark.parametrize("mag", [1e-4, 1e4])
def test_split_forward_under_rescaled_inputs(hip, nets, mag):
    """Count maps scaled by 1e-4 / 1e4 (far outside what a rollout produces): the per-tensor scales follow, the split path stays
    within 1e-4 of the output range of the fp32 pipe and picks the same goal cells."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = (make_count_maps(2, 128, seed=77) * mag).cuda()
    with torch.no_grad():
        o1, o2 = nets[0](x)
        f1, f2 = nets[1](x)
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
    rng = max(1e-30, float(f1.abs().max()))
    assert float((o1 - f1).abs().max()) <= 1e-4 * rng
    # the obstacle head is a sigmoid of logits that scale with the input: at 1e4 one fp32 ulp of a logit is already 1e-3
    assert float((o2 - f2).abs().max()) < (1e-4 if mag <= 1 else 5e-3)
    assert torch.equal(o1.amax(1).flatten(1).argmax(1), f1.amax(1).flatten(1).argmax(1))


_FUSION_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from nextbestpath_amd.networks import packing
from nextbestpath_amd.utility.synthetic import make_count_maps, make_explorer_state_dict
dev = torch.device("cuda")
packed = packing.pack_state_dict(make_explorer_state_dict(9), dev, precision="fp32_split")
for B, S in ((8, 256), (3, 128), (1, 64), (1, 256)):
    o1, o2 = packing.forward_packed(packed, make_count_maps(B, S, seed=B).to(dev))
    torch.save((o1.cpu(), o2.cpu()), f"{sys.argv[2]}_{B}_{S}.pt")
"""


def test_epilogue_fusions_against_the_separate_kernels(hip, tmp_path):
    """The encoder's max-pools and the one-channel sigmoid head ride in the producing convolution's epilogue and the attention
    gates' psi tail in the gate GEMM's (NBP_CONV_POOL / NBP_CONV_HEAD / NBP_GATE_PSI = 0 switch back to the separate kernels; NBP_SPLIT_R8_BLOCKS = 0
    keeps every launch on 16-row tiles; the switches are read once per process, hence
    the subprocesses; a single map of 256 x 256 is the case whose encoder runs split-K, where the pool rides in the reduce kernel).
    The pooled tensor is the max of the same four values: bit-identical outputs.  The fused psi sums
    q . w_psi in another order: equal to fp32 rounding of a 32..128-term dot product."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fwd.py"
    script.write_text(_FUSION_SCRIPT)
    outs = {}
    T = {"NBP_TUNING": "1"}          # the switches are honoured only under the explicit opt-in
    clean = {k: v for k, v in os.environ.items() if not k.startswith("NBP_")}
    for tag, env in (("both", {}), ("nopool", {**T, "NBP_CONV_POOL": "0"}), ("nopsi", {**T, "NBP_GATE_PSI": "0"}),
                     ("nohead", {**T, "NBP_CONV_HEAD": "0"}), ("nor8", {**T, "NBP_SPLIT_R8_BLOCKS": "0"}),
                     ("noup2", {**T, "NBP_SPLIT_UP2": "0"}), ("gatedw", {**T, "NBP_SPLIT_PSI_ON_LOAD": "0"}), ("psiall", {**T, "NBP_SPLIT_PSI_ON_LOAD": "2"}),
                     # a polluted environment WITHOUT the opt-in must change nothing
                     ("polluted", {"NBP_CONV_POOL": "0", "NBP_GATE_PSI": "0", "NBP_CONV_HEAD": "0", "NBP_SPLIT_MAX_K": "576",
                                   "NBP_SPLIT_MAX_K_SMALL": "288", "NBP_SPLIT_R8_SK": "1", "NBP_SPLIT_HALO": "0", "NBP_SPLIT_UP": "0",
                                   "NBP_SPLIT_GATE": "0", "NBP_CONV_PRECISION": "fp32", "NBP_XCD_REMAP": "0"})):
        subprocess.run([sys.executable, str(script), root, str(tmp_path / tag)], check=True, env={**clean, **env},
                       timeout=600)
        outs[tag] = {k: torch.load(tmp_path / f"{tag}_{k[0]}_{k[1]}.pt") for k in ((8, 256), (3, 128), (1, 64), (1, 256))}
    for k, (o1, o2) in outs["both"].items():
        p1, p2 = outs["nopool"][k]
        assert torch.equal(o1, p1) and torch.equal(o2, p2), k
        q1, q2 = outs["nopsi"][k]
        assert float((o1 - q1).abs().max()) <= 2e-5 * max(1.0, float(q1.abs().max())), k
        assert float((o2 - q2).abs().max()) <= 2e-5, k
        r1, r2 = outs["nohead"][k]                            # Final2 in the last convolution's epilogue: out1 untouched
        assert torch.equal(o1, r1) and float((o2 - r2).abs().max()) <= 2e-6, k
        # 8 x 32-pixel tiles (launches with fewer 16-row tiles than CUs) sum the same products in the same order per output, over the
        # same split-K slices: bit-identical
        t1, t2 = outs["nor8"][k]
        assert torch.equal(o1, t1) and torch.equal(o2, t2), k
        # up_conv layers: two column parities per workgroup on half-height tiles against one parity per workgroup -- the same
        # products in the same order into every accumulator: bit-identical
        w1, w2 = outs["noup2"][k]
        assert torch.equal(o1, w1) and torch.equal(o2, w2), k
        # round 6: x * psi formed in the consumer's halo staging (the gate writes psi [M] only) against the gated tensor written by the
        # gate and read back: the same fp32 products, scaled and split the same way: bit-identical
        for tag in ("gatedw", "psiall"):                      # (never / on every fused level; the default takes levels with >= 128 channels)
            g1, g2 = outs[tag][k]
            assert torch.equal(o1, g1) and torch.equal(o2, g2), (tag, k)
        u1, u2 = outs["polluted"][k]                          # no NBP_TUNING=1: the environment is not read
        assert torch.equal(o1, u1) and torch.equal(o2, u2), k


# ---- in-tensor dynamic range (the per-tensor scale's floor)
# The scale 2^(14 - floor(log2 max|x|)) is per TENSOR: an element 2^r below the tensor's max keeps the full two-piece precision
# (2^-23) while r <= 16 -- its lo piece is then still a normal fp16 -- and 2^(r - 39) beyond (lo falls into fp16's subnormal
# spacing 2^-24 of the scaled value): 2^-19 at a 10^6 spike, 2^-15 at 2^24.  The fp32 pipe has no such floor.
@pytest.mark.parametrize("r", [16, 20, 24])
def test_split_in_tensor_dynamic_range_floor(hip, r):
    """One tensor holding O(1) values AND a few spikes of 2^r: outputs whose 3x3 window contains a spike are dominated by fp32's
    own ulp of the spike terms on both pipes; outputs that do not see a spike carry the split scheme's floor 2^(r - 37) of
    sum |w x| -- nothing at r <= 16 (rollout tensors: counts up to 1e4 beside 1, 2^17 between a wall cell and the median cell
    of the encoder's activations), measurable beyond."""
    dev = "cuda"
    B, H, W, C, N = 1, 32, 64, 64, 64
    x = _rand(B, C, H, W, seed=1)
    spikes = [(5, 7), (16, 40), (30, 63)]
    for (yy, xx) in spikes:
        x[0, :, yy, xx] = float(2 ** r) * (1.0 + 0.25 * _rand(C, seed=yy))
    w = _rand(N, C, 3, 3, seed=3, scale=(6.0 / (C * 9)) ** 0.5)
    one, zero = torch.ones(N), torch.zeros(N)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1)          # sum |w x| per output
    near = torch.zeros(H, W, dtype=torch.bool)
    for (yy, xx) in spikes:
        near[max(0, yy - 1):yy + 2, max(0, xx - 1):xx + 2] = True
    xd, wd = nhwc(x).to(dev), w.to(dev).contiguous()
    got = nchw(conv3x3_split(xd, None, 0, pack_conv_split(wd), N, one.to(dev), zero.to(dev), False, 1)).cpu().double()
    f32 = nchw(conv_igemm(xd, None, 0, pack_conv(wd), N, 3, one.to(dev), zero.to(dev), False, 1, 0)).cpu().double()
    rel_s, rel_f = (got - ref).abs() / mag, (f32 - ref).abs() / mag
    far_s, far_f = rel_s[0, :, ~near], rel_f[0, :, ~near]
    near_s, near_f = rel_s[0, :, near], rel_f[0, :, near]
    print(f"\n[dynamic range 2^{r}] far outputs: split max {far_s.max():.2e} rms {far_s.pow(2).mean().sqrt():.2e} | fp32 pipe max "
          f"{far_f.max():.2e} rms {far_f.pow(2).mean().sqrt():.2e} | near: split {near_s.max():.2e} fp32 pipe {near_f.max():.2e}")
    # with a spike in the window both pipes sit at fp32's own rounding of the spike terms
    assert near_s.max().item() <= 3.0 * near_f.max().item() + 2.0 ** -24
    floor = 2.0 ** (r - 39) if r > 16 else 2.0 ** -23
    assert far_s.max().item() <= 2.0 * floor, (far_s.max().item(), floor)          # the documented floor holds ...
    if r <= 16:
        assert far_s.pow(2).mean().sqrt().item() <= 1.5 * far_f.pow(2).mean().sqrt().item() + 1e-9      # ... no loss in range
    else:
        assert far_s.pow(2).mean().sqrt().item() >= 2.0 ** (r - 46)                 # ... and is real (documented, not hidden)


def test_split_network_with_a_spike_in_the_input(hip, nets, nbp_weights):
    """Whole network on a count map with 10^6-count cells beside unit counts (2^20 of range inside the first activation tensors;
    rollouts reach 10^4): the split path against fp64 with the fp32 pipe and stock torch CPU fp32 beside it."""
    from nextbestpath_amd.utility.synthetic import make_count_maps
    x = make_count_maps(1, 128, seed=5)
    for (c, yy, xx) in ((0, 40, 41), (2, 64, 64), (3, 90, 30)):
        x[0, c, yy, xx] = 1.0e6
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in nbp_weights.items()}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        o1, o2 = nets[0](x.cuda())
        f1, f2 = nets[1](x.cuda())
        c1, c2 = nbp_net.nbp_forward(nbp_weights, x)
        d1, d2 = nbp_net.nbp_forward(sd64, x.double())
    rng = float(d1.abs().max())
    es, ef, ec = ((t.cpu().double() - d1).abs() for t in (o1, f1, c1))
    print(f"\n[network, 1e6 spike] out1 range {rng:.3g}: split max {es.max()/rng:.2e} mean {es.mean()/rng:.2e} | fp32 pipe max "
          f"{ef.max()/rng:.2e} mean {ef.mean()/rng:.2e} | torch fp32 max {ec.max()/rng:.2e} mean {ec.mean()/rng:.2e}; out2: split "
          f"{(o2.cpu().double()-d2).abs().max():.2e} fp32 pipe {(f2.cpu().double()-d2).abs().max():.2e} torch {(c2.double()-d2).abs().max():.2e}")
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
    assert es.max().item() <= max(1e-4 * rng, 3.0 * ef.max().item()) and es.mean().item() <= max(1e-6 * rng, 3.0 * ef.mean().item())
    assert torch.equal(o1.cpu().amax(1).flatten(1).argmax(1), d1.float().amax(1).flatten(1).argmax(1))


# (pixels as B, H, W; input channels; output channels)
CASES_1X1 = [(2, 16, 32, 64, 32), (1, 25, 40, 128, 64), (2, 8, 8, 512, 256), (3, 16, 16, 32, 64), (1, 16, 16, 256, 512), (1, 7, 9, 96, 160)]


@pytest.mark.parametrize("case", CASES_1X1)
def test_conv1x1_split_vs_fp64(hip, case):
    """nbp_conv1x1_split_f32 (training's W_g / W_x layers: the gates' kernel with one source) and its data gradient through
    nbp_pack_conv1x1_weight_split_dgrad against float64: the split scheme's error (<= a few 2^-24 of sum |terms|), whole-range
    pixel counts (M not a multiple of the 128-pixel workgroup), 1 .. 16 stages of 32 channels (K not a multiple of 128)."""
    B, H, W, C, N = case
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    x = (_rand(B, H, W, C, seed=C + N) * torch.logspace(-2, 0, C)).cuda().contiguous()
    w = (_rand(N, C, 1, 1, seed=N) / np.sqrt(C)).cuda().contiguous()
    bias = _rand(N, seed=3).cuda()
    M = B * H * W

    def run(src, Cin, planes, wamax, Nout, shift):
        slot = torch.zeros(64, dtype=torch.int32, device="cuda")
        _lib.check(L.nbp_amax_f32(src.data_ptr(), src.numel(), slot.data_ptr(), st), "amax")
        out = torch.full((B, H, W, Nout), float("nan"), device="cuda")
        one = torch.ones(Nout, device="cuda")
        _lib.check(L.nbp_conv1x1_split_f32(src.data_ptr(), Cin, M, planes.data_ptr(), wamax.data_ptr(), Nout, one.data_ptr(),
                                           shift.data_ptr(), 0, out.data_ptr(), slot.data_ptr(), st), "conv1x1_split")
        return out
    planes = torch.empty(C // 16 * 4 * N * 8, dtype=torch.int16, device="cuda")
    wamax = torch.empty(1, dtype=torch.int32, device="cuda")
    _lib.check(L.nbp_pack_conv_weight_split(w.data_ptr(), N, C, 1, None, 0, C, planes.data_ptr(), wamax.data_ptr(), st), "pack")
    y = run(x, C, planes, wamax, N, bias)
    xd, wd = x.double().view(M, C), w.double().view(N, C)
    want = xd @ wd.t() + bias.double()
    mag = xd.abs() @ wd.abs().t() + bias.double().abs()
    err = (y.double().view(M, N) - want).abs()
    assert bool(torch.isfinite(y).all()) and float((err / mag).max()) < 4e-7, float((err / mag).max())
    # data gradient: dx = dy W (w^T packed straight from the layer's [N][C] weight)
    dy = _rand(B, H, W, N, seed=7).cuda().contiguous()
    planes_t = torch.empty(N // 16 * 4 * C * 8, dtype=torch.int16, device="cuda")
    _lib.check(L.nbp_pack_conv1x1_weight_split_dgrad(w.data_ptr(), N, C, planes_t.data_ptr(), wamax.data_ptr(), st), "pack_dgrad")
    dx = run(dy, N, planes_t, wamax, C, torch.zeros(C, device="cuda"))
    want = dy.double().view(M, N) @ wd
    mag = dy.double().abs().view(M, N) @ wd.abs()
    err = (dx.double().view(M, C) - want).abs()
    assert float((err / mag.clamp_min(1e-30)).max()) < 4e-7, float((err / mag.clamp_min(1e-30)).max())
