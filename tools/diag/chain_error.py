"""Diagnostic (GPU box): where the forward's distance to fp64 comes from on rollout-derived inputs.

For every 3x3 conv + BN + ReLU layer of NBP the layer's fp64 input (rounded to fp32) is put through
  * the split kernel (one chain per output, automatic split-K, forced split-K),
  * the fp32 MFMA pipe (one chain),
  * stock torch CPU fp32 (conv2d + batch_norm + relu: the reference's arithmetic),
and each result is compared with the fp64 evaluation of the same layer: LOCAL error, nothing propagated.
Usage: chain_error.py [steps] [B]"""
import os, sys, tempfile
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_rollout_parity import _both_rollouts
from hip_helpers import conv3x3_split, conv_igemm, nchw, nhwc, pack_conv, pack_conv_split, pack_upconv_split, upconv3x3_split
from nextbestpath_amd.networks.packing import fold_affine

N_STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tmp = tempfile.mkdtemp()
if len(sys.argv) > 2 and sys.argv[2] == "hard":          # the 29 k-face scene of test_hip_rollout_equals_oracle_rollout_hard_scene
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

# fp64 walk, recording (conv prefix, bn prefix, [sources], ups) for every 3x3 layer
layers = []
def bn(sdx, p, x):
    return F.batch_norm(x, sdx[p + ".running_mean"], sdx[p + ".running_var"], sdx[p + ".weight"], sdx[p + ".bias"], False, 0.1, 1e-5)
def c3(p, q, srcs, ups=False):
    x = torch.cat(srcs, 1) if len(srcs) > 1 else srcs[0]
    if ups: x = F.interpolate(x, scale_factor=2)
    y = F.relu(bn(sd64, q, F.conv2d(x, sd64[p + ".weight"], sd64[p + ".bias"], padding=1)))
    layers.append((p, q, [t.clone() for t in srcs], ups, y))
    return y
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
    for d, levels in ((1, (5, 4)), (2, (5, 4, 3, 2))):
        cur = x5
        for L in levels:
            dd = c3(f"Up{L}_{d}.up.1", f"Up{L}_{d}.up.2", [cur], ups=True)
            a = att(f"Att{L}_{d}", dd, skips[L])
            cur = block(f"Up_conv{L}_{d}", [a, dd])

from oracle import nbp_net
with torch.no_grad():
    d1, _ = nbp_net.nbp_forward(sd64, x_in.double())
    c1, _ = nbp_net.nbp_forward(sd, x_in)
    rngo = d1.abs().max().item()
    line = f"whole network, out1 range {rngo:.1f}: torch32 mean {(c1.double()-d1).abs().mean().item()/rngo:.2e}"
    for prec in ("fp32_split", "fp32"):
        hip_ro.nbp.conv_precision = prec
        h1, _ = hip_ro.nbp(x_in.cuda())
        line += f" | {prec} mean {(h1.cpu().double()-d1).abs().mean().item()/rngo:.2e} max {(h1.cpu().double()-d1).abs().max().item()/rngo:.2e}"
    hip_ro.nbp.conv_precision = "fp32_split"
print(line, flush=True)
if os.environ.get("CHAIN_ERROR_NET_ONLY"):
    sys.exit(0)
print(f"input max {x_in.max().item():.0f}; columns: mean|err| / max|out64| (max|err| / max|out64|); last column: in-tensor range of the layer's input = max / median of the non-zero |x|")
print(f"{'layer':22s} {'K':>5s} {'range':>9s} | {'torch32':>19s} | {'split sk=1':>19s} | {'split auto':>19s} | {'split sk=4':>19s} | {'fp32 pipe sk=1':>19s}")
dev = "cuda"
tot = {}
for p, q, srcs, ups, y64 in layers:
    if srcs[0].shape[1] == 5:
        continue                        # first conv: its own kernel
    w = sd[p + ".weight"]
    N, C = w.shape[0], w.shape[1]
    scale, shift = fold_affine(sd, p, q)
    scd, shd = scale.float().to(dev), shift.float().to(dev)
    s32 = [t.float() for t in srcs]
    with torch.no_grad():
        xin = torch.cat(s32, 1) if len(s32) > 1 else s32[0]
        if ups: xin = F.interpolate(xin, scale_factor=2)
        t32 = F.relu(bn(sd, q, F.conv2d(xin, w, sd[p + ".bias"], padding=1))).double()
    rng = float(y64.abs().max())
    def err(t):
        e = (t - y64).abs()
        return float(e.mean()) / rng, float(e.max()) / rng
    wd = w.to(dev).contiguous()
    x0d = nhwc(s32[0]).to(dev); x1d = nhwc(s32[1]).to(dev) if len(s32) > 1 else None
    res = {"torch32": err(t32)}
    if ups:
        pk = pack_upconv_split(wd)
        for nm, sk in (("split sk=1", 1), ("split auto", 0), ("split sk=4", 4)):
            res[nm] = err(nchw(upconv3x3_split(x0d, pk, N, scd, shd, True, sk)).cpu().double())
    else:
        pk = pack_conv_split(wd)
        for nm, sk in (("split sk=1", 1), ("split auto", 0), ("split sk=4", 4)):
            res[nm] = err(nchw(conv3x3_split(x0d, x1d, 0, pk, N, scd, shd, True, sk)).cpu().double())
    res["fp32 pipe sk=1"] = err(nchw(conv_igemm(x0d, x1d, int(ups), pack_conv(wd), N, 3, scd, shd, True, 1, 0)).cpu().double())
    cols = " | ".join(f"{res[k][0]:.2e} ({res[k][1]:.2e})" for k in ("torch32", "split sk=1", "split auto", "split sk=4", "fp32 pipe sk=1"))
    xa = xin.abs()
    dyn = float(xa.max() / xa[xa > 0].median())
    print(f"{p:22s} {9 * C:5d} {rng:9.3g} | {cols} | 2^{np.log2(dyn):.1f}", flush=True)
    for k, v in res.items():
        tot.setdefault(k, []).append(v[0])
print("geometric mean of mean-error ratios to torch32:",
      {k: float(np.exp(np.mean(np.log(np.array(v) / np.array(tot['torch32']))))) for k, v in tot.items()})
