"""Diagnostic (GPU box): which layer's arithmetic error dominates the network's distance to fp64 on a rollout-derived input.

The network is evaluated in fp64 (torch CPU) with ONE 3x3 conv + BN + ReLU layer at a time replaced by (a) the split kernel,
(b) the fp32 MFMA pipe, (c) stock torch CPU fp32 -- each fed the fp64 input rounded to fp32 -- and the resulting out1 is
compared with the all-fp64 out1: local error of that layer x its amplification through the rest of the (exact) network.
Usage: layer_substitution.py [steps] [hard]   |   layer_substitution.py <steps> scene <k>   (the input of step <steps> of parity_long's scene k)"""
import os, sys, tempfile
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_rollout_parity import _both_rollouts
from hip_helpers import conv3x3_split, conv_igemm, nchw, nhwc, pack_conv, pack_conv_split, pack_upconv_split, upconv3x3_split
from nextbestpath_amd.networks.packing import fold_affine

N_STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 9
tmp = tempfile.mkdtemp()
if len(sys.argv) > 3:                       # a scene of tools/diag/parity_long.py: layer_substitution.py <steps> scene <k>
    k = int(sys.argv[3])
    hip_ro, ora, mesh = _both_rollouts(tmp, cells=8, size=4.8, tess=0.3, scene_seed=k, seed=5 + k)
elif len(sys.argv) > 2 and sys.argv[2] == "hard":
    hip_ro, ora, mesh = _both_rollouts(tmp, cells=12, size=7.2, tess=0.15, scene_seed=101, seed=9)
else:
    hip_ro, ora, mesh = _both_rollouts(tmp, cells=8, size=4.8, tess=0.3, scene_seed=0, seed=5)
for s in range(N_STEPS):
    hip_ro.pre()
    with torch.no_grad():
        o1, o2 = hip_ro.nbp(hip_ro.st.net_in)
    hip_ro.plan_enqueue(o1, o2); torch.cuda.synchronize(); hip_ro.plan_finish(); hip_ro.post()
hip_ro.pre()
x_in = hip_ro.st.net_in.cpu()
sd = ora.sd
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
torch.set_num_threads(min(64, os.cpu_count() or 8))
dev = "cuda"


def bn(sdx, p, x):
    return F.batch_norm(x, sdx[p + ".running_mean"], sdx[p + ".running_var"], sdx[p + ".weight"], sdx[p + ".bias"], False, 0.1, 1e-5)


def layer_hip(p, q, srcs, ups, mode):
    w = sd[p + ".weight"]
    N = w.shape[0]
    scale, shift = fold_affine(sd, p, q)
    scd, shd = scale.float().to(dev), shift.float().to(dev)
    s32 = [t.float() for t in srcs]
    wd = w.to(dev).contiguous()
    x0d = nhwc(s32[0]).to(dev); x1d = nhwc(s32[1]).to(dev) if len(s32) > 1 else None
    if mode == "t32":
        xin = torch.cat(s32, 1) if len(s32) > 1 else s32[0]
        if ups: xin = F.interpolate(xin, scale_factor=2)
        return F.relu(bn(sd, q, F.conv2d(xin, w, sd[p + ".bias"], padding=1))).double()
    if mode == "f32pipe":
        return nchw(conv_igemm(x0d, x1d, int(ups), pack_conv(wd), N, 3, scd, shd, True, 0, 0)).cpu().double()
    if ups:
        return nchw(upconv3x3_split(x0d, pack_upconv_split(wd), N, scd, shd, True, 0)).cpu().double()
    return nchw(conv3x3_split(x0d, x1d, 0, pack_conv_split(wd), N, scd, shd, True, 0)).cpu().double()


names = []


def forward(subst=None, mode=None):
    def c3(p, q, srcs, ups=False):
        if p not in names: names.append(p)
        if subst is not None and (subst == "ALL" or p == subst) and srcs[0].shape[1] != 5:
            return layer_hip(p, q, srcs, ups, mode)
        x = torch.cat(srcs, 1) if len(srcs) > 1 else srcs[0]
        if ups: x = F.interpolate(x, scale_factor=2)
        return F.relu(bn(sd64, q, F.conv2d(x, sd64[p + ".weight"], sd64[p + ".bias"], padding=1)))
    def block(name, srcs):
        return c3(name + ".conv.3", name + ".conv.4", [c3(name + ".conv.0", name + ".conv.1", srcs)])
    def att(p, g, x):
        g1 = bn(sd64, p + ".W_g.1", F.conv2d(g, sd64[p + ".W_g.0.weight"], sd64[p + ".W_g.0.bias"]))
        x1 = bn(sd64, p + ".W_x.1", F.conv2d(x, sd64[p + ".W_x.0.weight"], sd64[p + ".W_x.0.bias"]))
        psi = torch.sigmoid(bn(sd64, p + ".psi.1", F.conv2d(F.relu(g1 + x1), sd64[p + ".psi.0.weight"], sd64[p + ".psi.0.bias"])))
        return x * psi
    with torch.no_grad():
        x = x_in.double()
        x1 = block("Conv1", [x]); x2 = block("Conv2", [F.max_pool2d(x1, 2, 2)]); x3 = block("Conv3", [F.max_pool2d(x2, 2, 2)])
        x4 = block("Conv4", [F.max_pool2d(x3, 2, 2)]); x5 = block("Conv5", [F.max_pool2d(x4, 2, 2)])
        skips = {5: x4, 4: x3, 3: x2, 2: x1}
        cur = x5
        for L in (5, 4):
            dd = c3(f"Up{L}_1.up.1", f"Up{L}_1.up.2", [cur], ups=True)
            cur = block(f"Up_conv{L}_1", [att(f"Att{L}_1", dd, skips[L]), dd])
        return F.conv2d(cur, sd64["Final1.weight"], sd64["Final1.bias"])


ref = forward()
rng = ref.abs().max().item()
print(f"out1 range {rng:.1f}; columns: mean |err| / range (max |err| / range) of out1 when ONLY this layer is inexact")
for mode in ("split", "f32pipe", "t32"):
    o = forward("ALL", mode)
    e = (o - ref).abs()
    print(f"ALL 3x3 layers of the value decoder's path via {mode:8s}: mean {e.mean().item()/rng:.2e} max {e.max().item()/rng:.2e}", flush=True)
print(f"{'layer':22s} | {'split':>21s} | {'fp32 pipe':>21s} | {'torch fp32':>21s}")
for p in [n for n in names if not n.startswith("Conv1.conv.0")]:
    cols = []
    for mode in ("split", "f32pipe", "t32"):
        e = (forward(p, mode) - ref).abs()
        cols.append(f"{e.mean().item()/rng:.2e} ({e.max().item()/rng:.2e})")
    print(f"{p:22s} | " + " | ".join(cols), flush=True)
