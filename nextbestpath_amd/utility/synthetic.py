"""Deterministic synthetic inputs (no AiMDoom meshes / released weights are available
offline): NBP weights with non-trivial BatchNorm statistics, point-count maps, point clouds.
Used by tests, the golden-vector generator and bench.py.  numpy PCG64 streams, so the same
seed gives the same tensors in the build container and on the GPU box."""
from __future__ import annotations

import numpy as np
import torch


def nbp_state_dict_spec():
    from ..networks.nbp_model import NBP
    with torch.device("meta"):
        m = NBP()
    return [(k, tuple(v.shape), v.dtype) for k, v in m.state_dict().items()]


def make_nbp_state_dict(seed: int = 9):
    """Kaiming-scaled conv weights, BN gamma/beta/mean/var away from the identity."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape, dtype in nbp_state_dict_spec():
        if name == "log_vars":
            t = np.zeros(shape, np.float32)
        elif name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        elif name.endswith("running_var"):
            t = rng.uniform(0.6, 1.4, shape).astype(np.float32)
        elif name.endswith("running_mean"):
            t = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif len(shape) == 4:                                  # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            b = np.sqrt(6.0 / fan_in)
            t = (rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(b)
        elif name.endswith(".weight"):                         # BN gamma
            t = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        else:                                                  # conv bias / BN beta
            t = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        sd[name] = torch.from_numpy(np.ascontiguousarray(t))
    return sd


def make_explorer_state_dict(seed: int = 9):
    """Synthetic weights for ROLLOUTS: same as make_nbp_state_dict but with a silent obstacle head
    (Final2 weight 0, bias -4 => out2 = 0.018 < 0.13 everywhere).  With plain random weights the
    obstacle head fires on most pixels, every lattice edge is blocked and the agent only turns in
    place; with a silent head the observed walls (height-band projection) do the blocking, which is
    what a trained network converges to in explored space.  Cost of the forward is unchanged."""
    sd = make_nbp_state_dict(seed)
    sd["Final2.0.weight"] = torch.zeros_like(sd["Final2.0.weight"])
    sd["Final2.0.bias"] = torch.full_like(sd["Final2.0.bias"], -4.0)
    return sd


def make_count_maps(B: int, S: int, seed: int = 0, lam: float = 0.3):
    """[B,5,S,S] fp32: Poisson(lam) counts inside a random disc on the 4 slab channels,
    a sparse trajectory on channel 4 (SURVEY.md section 8d, config 3 input recipe)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, 5, S, S), np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        cy, cx = rng.uniform(0.3 * S, 0.7 * S, 2)
        r = rng.uniform(0.25 * S, 0.5 * S)
        disc = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        for c in range(4):
            x[b, c] = rng.poisson(lam, (S, S)).astype(np.float32) * disc
        n = int(rng.integers(5, 60))
        py = np.clip(np.cumsum(rng.integers(-2, 3, n)) + S // 2, 0, S - 1)
        px = np.clip(np.cumsum(rng.integers(-2, 3, n)) + S // 2, 0, S - 1)
        np.add.at(x[b, 4], (py, px), 1.0)
    return torch.from_numpy(x)


def make_point_cloud(N: int, seed: int = 0, extent: float = 55.0, y_range=(0.0, 30.0)):
    """Wall-like cloud: points on random vertical segments + floor clutter, [N,3] fp32 (y up)."""
    rng = np.random.default_rng(seed)
    n_walls = 40
    a = rng.uniform(-extent, extent, (n_walls, 2))
    d = rng.uniform(-25, 25, (n_walls, 2))
    w = rng.integers(0, n_walls, N)
    t = rng.random(N)
    xz = a[w] + d[w] * t[:, None] + rng.normal(0, 0.02, (N, 2))
    y = rng.uniform(y_range[0], y_range[1], N)
    return torch.from_numpy(np.stack([xz[:, 0], y, xz[:, 1]], 1).astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------------------------
# Block-level backward cases (tests/golden/nbp_blocks_bwd.npz; VERDICT r05 Next 5).  One case = one of the reference's three block
# classes (nbp_model.py: conv_block :8-21, up_conv :23-34, Attention_block :36-62) in TRAIN mode with parameters, inputs and an
# upstream gradient drawn here.  The BatchNorm shifts keep every ReLU pre-activation clear of zero (beta = +4 against a unit-variance
# normalised activation scaled by gamma <= 0.7: a 5.7 sigma margin; the gates' sum of two at gamma <= 0.45: 6.3 sigma), so the block is a SMOOTH function of its inputs and parameters
# and two fp32 arithmetics agree to rounding -- no mask flips, unlike the whole-network gradients of nbp_train_*.npz.
BLOCK_CASES = [
    # tag,            kind,         cin (per source), cout, B,  H,  W
    ("cb_64_128",     "conv_block", (64,),            128,  2, 32, 32),
    ("cb_cat_128_64", "conv_block", (64, 64),         64,   1, 32, 64),      # decoder form: cat(x0, x1) as two sources
    ("cb_256_256_16", "conv_block", (256,),           256,  2, 16, 16),      # a 16-pixel-wide level
    ("up_128_64",     "up_conv",    (128,),           64,   2, 16, 32),      # output 32 x 64
    ("up_512_256",    "up_conv",    (512,),           256,  1, 16, 16),      # output 32 x 32 (16-pixel-wide low-resolution tiles)
    ("att_64_32",     "attention",  (64, 64),         32,   2, 32, 32),      # F_g = F_l = 64, F_int = 32
    ("att_256_128",   "attention",  (256, 256),       128,  1, 16, 32),
]


def block_param_spec(kind, cin, cout):
    """[(state_dict key of the reference block, shape, role)] in the reference module's own key order."""
    C = sum(cin)
    bn = lambda p, n: [(p + ".weight", (n,), "gamma"), (p + ".bias", (n,), "beta"), (p + ".running_mean", (n,), "mean"),
                       (p + ".running_var", (n,), "var")]
    if kind == "conv_block":
        return ([("conv.0.weight", (cout, C, 3, 3), "w"), ("conv.0.bias", (cout,), "b")] + bn("conv.1", cout) +
                [("conv.3.weight", (cout, cout, 3, 3), "w"), ("conv.3.bias", (cout,), "b")] + bn("conv.4", cout))
    if kind == "up_conv":
        return [("up.1.weight", (cout, C, 3, 3), "w"), ("up.1.bias", (cout,), "b")] + bn("up.2", cout)
    if kind == "attention":
        Fg, Fl = cin
        return ([("W_g.0.weight", (cout, Fg, 1, 1), "w"), ("W_g.0.bias", (cout,), "b")] + bn("W_g.1", cout) +
                [("W_x.0.weight", (cout, Fl, 1, 1), "w"), ("W_x.0.bias", (cout,), "b")] + bn("W_x.1", cout) +
                [("psi.0.weight", (1, cout, 1, 1), "w"), ("psi.0.bias", (1,), "b"),
                 ("psi.1.weight", (1,), "gamma_psi"), ("psi.1.bias", (1,), "beta_psi"), ("psi.1.running_mean", (1,), "mean"),
                 ("psi.1.running_var", (1,), "var")])
    raise ValueError(kind)


def make_block_case(tag: str):
    """-> (kind, cin, cout, state_dict, inputs [NCHW fp32 tensors], dy NCHW fp32) for one of BLOCK_CASES; numpy PCG64 streams."""
    import zlib
    row = next(r for r in BLOCK_CASES if r[0] == tag)
    _, kind, cin, cout, B, H, W = row
    rng = np.random.default_rng(zlib.crc32(tag.encode()))
    sd = {}
    for key, shape, role in block_param_spec(kind, cin, cout):
        if role == "w":
            fan_in = shape[1] * shape[2] * shape[3]
            t = rng.normal(0.0, 1.0 / np.sqrt(fan_in), shape)
        elif role == "b":
            t = rng.uniform(-0.1, 0.1, shape)
        elif role == "gamma":
            t = rng.uniform(0.2, 0.45, shape) if kind == "attention" else rng.uniform(0.3, 0.7, shape)
        elif role == "beta":                       # ReLU stays open: see the section comment
            t = rng.uniform(3.8, 4.2, shape) / (2.0 if kind == "attention" else 1.0)      # (the gate's ReLU sees g1 + x1: 2 + 2)
        elif role == "gamma_psi":
            t = rng.uniform(0.8, 1.2, shape)
        elif role == "beta_psi":
            t = rng.uniform(-0.3, 0.3, shape)
        elif role == "mean":
            t = np.zeros(shape)
        else:
            t = np.ones(shape)
        sd[key] = torch.from_numpy(np.ascontiguousarray(t.astype(np.float32)))
    for key in list(sd):
        if key.endswith("running_var"):
            sd[key.replace("running_var", "num_batches_tracked")] = torch.zeros((), dtype=torch.int64)
    inputs = [torch.from_numpy((rng.normal(0.5, 1.0, (B, c, H, W))).astype(np.float32)) for c in cin]
    up = 2 if kind == "up_conv" else 1
    out_c = cin[1] if kind == "attention" else cout
    dy = torch.from_numpy(rng.normal(0.0, 1.0, (B, out_c, H * up, W * up)).astype(np.float32))
    return kind, cin, cout, sd, inputs, dy
