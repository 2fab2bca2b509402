"""CPU restatement of the NBP network on a plain state_dict (torch fp32 functional ops).

Follows next_best_path/networks/nbp_model.py: conv_block :8-21, up_conv :23-34,
Attention_block :36-62, NBP.forward :110-160, NBP.loss :162-173.  The network is a
floating-point kernel, so the oracle is a torch fp32 reference of the same ops (the
reference's own arithmetic is exactly these ATen calls).  Pinned by
tests/golden/nbp_fwd_*.npz (outputs of the reference module itself).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bn(sd, p, x, train):
    if train:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.1, 1e-5)


def _conv(sd, p, x, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)


def conv_block(sd, p, x, train=False):            # ref :8-21
    x = F.relu(_bn(sd, p + ".conv.1", _conv(sd, p + ".conv.0", x, 1), train))
    return F.relu(_bn(sd, p + ".conv.4", _conv(sd, p + ".conv.3", x, 1), train))


def up_conv(sd, p, x, train=False):               # ref :23-34
    x = F.interpolate(x, scale_factor=2)          # nn.Upsample default mode = nearest
    return F.relu(_bn(sd, p + ".up.2", _conv(sd, p + ".up.1", x, 1), train))


def attention(sd, p, g, x, train=False):          # ref :36-62
    g1 = _bn(sd, p + ".W_g.1", _conv(sd, p + ".W_g.0", g, 0), train)
    x1 = _bn(sd, p + ".W_x.1", _conv(sd, p + ".W_x.0", x, 0), train)
    psi = F.relu(g1 + x1)
    psi = torch.sigmoid(_bn(sd, p + ".psi.1", _conv(sd, p + ".psi.0", psi, 0), train))
    return x * psi


def nbp_forward(sd, x, train=False, return_intermediates=False):   # ref :110-160
    inter = {}
    x1 = conv_block(sd, "Conv1", x, train)
    x2 = conv_block(sd, "Conv2", F.max_pool2d(x1, 2, 2), train)
    x3 = conv_block(sd, "Conv3", F.max_pool2d(x2, 2, 2), train)
    x4 = conv_block(sd, "Conv4", F.max_pool2d(x3, 2, 2), train)
    x5 = conv_block(sd, "Conv5", F.max_pool2d(x4, 2, 2), train)
    inter.update(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5)
    skips = {5: x4, 4: x3, 3: x2, 2: x1}
    outs = {}
    for d, levels in ((1, (5, 4)), (2, (5, 4, 3, 2))):
        cur = x5
        for L in levels:
            dd = up_conv(sd, f"Up{L}_{d}", cur, train)
            a = attention(sd, f"Att{L}_{d}", dd, skips[L], train)
            cur = conv_block(sd, f"Up_conv{L}_{d}", torch.cat((a, dd), dim=1), train)
            inter[f"d{L}_{d}"] = cur
        outs[d] = cur
    out1 = F.conv2d(outs[1], sd["Final1.weight"], sd["Final1.bias"])
    out2 = torch.sigmoid(F.conv2d(outs[2], sd["Final2.0.weight"], sd["Final2.0.bias"]))
    if return_intermediates:
        return out1, out2, inter
    return out1, out2


def nbp_loss(log_vars, pred1, target1, pred2, target2):            # ref :162-173
    s1 = torch.exp(2 * log_vars[0])
    s2 = torch.exp(2 * log_vars[1])
    l1 = (1.0 / (2.0 * s1)) * F.mse_loss(pred1, target1) + log_vars[0]
    l2 = (1.0 / s2) * F.binary_cross_entropy(pred2, target2) + log_vars[1]
    return l1 + l2
