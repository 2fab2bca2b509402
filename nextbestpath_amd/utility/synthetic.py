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
