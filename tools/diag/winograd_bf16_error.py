#!/usr/bin/env python
"""CPU experiment (no GPU): how much accuracy would a Winograd F(2x2, 3x3) form cost the bf16 path?
Direct bf16 convolution = bf16 activations and weights, exact products, fp32-like accumulation.  Winograd bf16 = the same bf16
tensors, input transform B^T d B and weight transform G w G^T in fp32, BOTH rounded to bf16 (they are the MFMA operands),
products summed over channels in fp32, output transform A^T m A in fp32.  Errors against an fp64 convolution of the UNROUNDED
fp32 tensors, relative to the output's range.
    python tools/diag/winograd_bf16_error.py"""
import numpy as np
import torch
import torch.nn.functional as F


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float64)


def winograd(x, w):          # x [C, H, W] (bf16-valued, float64), w [N, C, 3, 3]; returns [N, H-2, W-2] (valid)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    U = bf16(torch.einsum("ij,ncjk,lk->ncil", G, w, G).float())                      # [N, C, 4, 4], fp32 transform, bf16 operand
    C, H, W = x.shape
    th, tw = (H - 2) // 2, (W - 2) // 2
    tiles = x.unfold(1, 4, 2).unfold(2, 4, 2)                                         # [C, th, tw, 4, 4]
    V = bf16(torch.einsum("ij,cyxjk,lk->cyxil", Bt, tiles, Bt).float())               # [C, th, tw, 4, 4]
    M = torch.einsum("ncil,cyxil->nyxil", U, V)                                       # exact products, fp64 sum (>= fp32 accumulation)
    Y = torch.einsum("ij,nyxjk,lk->nyxil", At, M, At)                                 # [N, th, tw, 2, 2]
    return Y.permute(0, 1, 3, 2, 4).reshape(-1, 2 * th, 2 * tw)


torch.manual_seed(0)
for name, C, N, S, make in (("uniform activations", 64, 64, 34, lambda s: torch.rand(s) * 4 - 0.3),
                            ("post-ReLU, heavy tail", 128, 128, 34, lambda s: torch.relu(torch.randn(s)) * torch.exp(torch.randn(s))),
                            ("count-map like (first layers)", 64, 64, 34, lambda s: torch.floor(torch.rand(s) ** 6 * 3000))):
    x = make((C, S, S)).float()
    w = (torch.randn(N, C, 3, 3) / np.sqrt(9 * C)).float()
    ref = F.conv2d(x.double()[None], w.double())[0]
    rng = float(ref.abs().max())
    direct = F.conv2d(bf16(x)[None], bf16(w))[0]
    wino = winograd(bf16(x), bf16(w))
    e_d, e_w = (direct - ref).abs(), (wino - ref).abs()
    print(f"{name:32s} C={C:3d} range {rng:9.3g}   direct bf16: max {float(e_d.max()) / rng:.2e} mean {float(e_d.mean()) / rng:.2e}   "
          f"winograd bf16: max {float(e_w.max()) / rng:.2e} mean {float(e_w.mean()) / rng:.2e}   ratio (mean) {float(e_w.mean() / e_d.mean()):.2f}")
