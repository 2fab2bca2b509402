"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the bf16 data path of nbp_forward_bf16 (include/nbp_hip.h):
the network of next_best_path/networks/nbp_model.py:110-160 with the rounding points of the HIP path --
3x3 / attention-gate weights rounded to bf16 (attention weights after the BatchNorm scale is multiplied in),
every stored activation rounded to bf16 once, accumulation exact (float64 here, fp32 on the MFMA), epilogues
(folded BatchNorm + bias, ReLU, sigmoid, psi gate) in fp32, first conv and the two heads with fp32 weights.

Pinning: the rounding model is this repo's own design (the reference has no bf16 path), so the checker is
pinned only through nbp_net.nbp_forward (the golden-vector-pinned fp32 restatement): tests compare both the
HIP bf16 path and this restatement against it within the bf16 tolerance stated in tests/test_gpu_bf16.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def rbf(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest-even bf16 -> fp32."""
    return x.float().to(torch.bfloat16).to(torch.float32)


def _fold(sd, conv, bn, eps=1e-5):
    bias = sd[conv + ".bias"].double()
    if bn is None:
        return torch.ones_like(bias).float(), bias.float()
    g, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
    mu, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    scale = g / torch.sqrt(var + eps)
    return scale.float(), ((bias - mu) * scale + b).float()


def _affine(acc, scale, shift):
    return acc.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def _conv3(sd, conv, bn, x, w_bf16=True):
    w = sd[conv + ".weight"].float()
    if w_bf16:
        w = rbf(w)
    scale, shift = _fold(sd, conv, bn)
    acc = F.conv2d(x.double(), w.double(), None, padding=1)
    return rbf(F.relu(_affine(acc, scale, shift)))


def _block(sd, p, x, first=False):
    x = _conv3(sd, p + ".conv.0", p + ".conv.1", x, w_bf16=not first)
    return _conv3(sd, p + ".conv.3", p + ".conv.4", x)


def _attention(sd, p, g, x):
    sg, tg = _fold(sd, p + ".W_g.0", p + ".W_g.1")
    sx, tx = _fold(sd, p + ".W_x.0", p + ".W_x.1")
    wg = rbf(sd[p + ".W_g.0.weight"].float() * sg.view(-1, 1, 1, 1))
    wx = rbf(sd[p + ".W_x.0.weight"].float() * sx.view(-1, 1, 1, 1))
    acc = F.conv2d(g.double(), wg.double()) + F.conv2d(x.double(), wx.double())
    q = rbf(F.relu(acc.float() + (tg + tx).view(1, -1, 1, 1)))
    sp, tp = _fold(sd, p + ".psi.0", p + ".psi.1")
    z = F.conv2d(q.double(), sd[p + ".psi.0.weight"].double()).float() * sp.view(1, -1, 1, 1) + tp.view(1, -1, 1, 1)
    return rbf(x * torch.sigmoid(z))


def nbp_forward_bf16(sd, x):
    x1 = _block(sd, "Conv1", x.float(), first=True)
    x2 = _block(sd, "Conv2", F.max_pool2d(x1, 2, 2))
    x3 = _block(sd, "Conv3", F.max_pool2d(x2, 2, 2))
    x4 = _block(sd, "Conv4", F.max_pool2d(x3, 2, 2))
    x5 = _block(sd, "Conv5", F.max_pool2d(x4, 2, 2))
    skips = {5: x4, 4: x3, 3: x2, 2: x1}
    outs = {}
    for d, levels in ((1, (5, 4)), (2, (5, 4, 3, 2))):
        cur = x5
        for L in levels:
            dd = _conv3(sd, f"Up{L}_{d}.up.1", f"Up{L}_{d}.up.2", F.interpolate(cur, scale_factor=2))
            a = _attention(sd, f"Att{L}_{d}", dd, skips[L])
            cur = _block(sd, f"Up_conv{L}_{d}", torch.cat((a, dd), dim=1))
        outs[d] = cur
    out1 = (F.conv2d(outs[1].double(), sd["Final1.weight"].double()).float() + sd["Final1.bias"].view(1, -1, 1, 1))
    out2 = torch.sigmoid(F.conv2d(outs[2].double(), sd["Final2.0.weight"].double()).float()
                         + sd["Final2.0.bias"].view(1, -1, 1, 1))
    return out1, out2
