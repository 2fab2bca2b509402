"""Diagnostic (GPU box): per BatchNorm layer, forward error vs fp64 and number of ReLU-mask mismatches -- HIP vs torch32."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from nextbestpath_amd.networks import training as tr
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
from oracle import nbp_net
g = np.load("/root/repo/tests/golden/nbp_train_S128B4.npz")
sd = make_nbp_state_dict(9)
x = torch.from_numpy(g["x"])
rec = {"hip": [], "t32": [], "t64": []}
orig_bn = nbp_net._bn
def make(tag):
    def f(sd_, p, xx, train):
        y = orig_bn(sd_, p, xx, train); rec[tag].append((p, y.detach().double())); return y
    return f
torch.set_num_threads(32)
nbp_net._bn = make("t32"); nbp_net.nbp_forward(sd, x, train=True)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
nbp_net._bn = make("t64"); nbp_net.nbp_forward(sd64, x.double(), train=True)
orig = tr._bn
def hip_bn(mod, xx, relu):
    y = orig(mod, xx, False)                      # record the pre-ReLU value, then apply the ReLU the way the fused path does
    rec["hip"].append((None, y.detach().permute(0, 3, 1, 2).cpu().double()))
    return tr.AddReluFn.apply(y, torch.zeros_like(y)) if relu else y
tr._bn = hip_bn
net = NBP(); net.load_state_dict(sd); net = net.to("cuda").train()
net(x.cuda())
for (p, y64), (_, y32), (_, yh) in zip(rec["t64"], rec["t32"], rec["hip"]):
    sc = float(y64.abs().max())
    eh, et = float((yh - y64).abs().max()) / sc, float((y32 - y64).abs().max()) / sc
    fh, ft = int(((yh > 0) != (y64 > 0)).sum()), int(((y32 > 0) != (y64 > 0)).sum())
    near = int((y64.abs() < 1e-6 * sc).sum())
    print(f"{p:22s} n={y64.numel():8d} err hip {eh:.1e} t32 {et:.1e} | sign flips hip {fh:6d} t32 {ft:6d} | |y|<1e-6 scale: {near}")
