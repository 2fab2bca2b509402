"""Diagnostic (GPU box): gradients w.r.t. the tensors inside Up_conv4_1 during the FULL network backward -- HIP and torch32 vs fp64."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, torch.nn.functional as F
from nextbestpath_amd.networks import training as tr
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
from oracle import nbp_net
g = np.load("/root/repo/tests/golden/nbp_train_S128B4.npz")
sd = make_nbp_state_dict(9)
x, coords, gains, gt = (torch.from_numpy(g[k]) for k in ("x", "coords", "gains", "gt"))
BLOCK = sys.argv[1] if len(sys.argv) > 1 else "Up_conv4_1"
torch.set_num_threads(32)

def ref(dt):
    sdd = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    keep = {}
    orig = nbp_net.conv_block
    def cb(sd_, p, xx, train=False):
        if p != BLOCK: return orig(sd_, p, xx, train)
        c0 = nbp_net._conv(sd_, p + ".conv.0", xx, 1); c0.retain_grad()
        r1 = F.relu(nbp_net._bn(sd_, p + ".conv.1", c0, train)); r1.retain_grad()
        c3 = nbp_net._conv(sd_, p + ".conv.3", r1, 1); c3.retain_grad()
        out = F.relu(nbp_net._bn(sd_, p + ".conv.4", c3, train)); out.retain_grad()
        xx.retain_grad()
        keep.update(x_in=xx, conv0=c0, relu1=r1, conv3=c3, out=out)
        return out
    nbp_net.conv_block = cb
    o1, o2 = nbp_net.nbp_forward(sdd, x.to(dt), train=True)
    nbp_net.conv_block = orig
    pred = o1[coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3]]
    nbp_net.nbp_loss(sdd["log_vars"], pred, gains.to(dt), o2, gt.to(dt)).backward()
    return {k: v.grad.double() for k, v in keep.items()}, {k: v.detach().double() for k, v in keep.items()}
g64, a64 = ref(torch.float64)
g32, a32 = ref(torch.float32)
net = NBP(); net.load_state_dict(sd); net = net.to("cuda").train()
keep = {}
orig_block = tr._block
def hb(seq, x0, x1=None):
    if seq is not getattr(net, BLOCK).conv: return orig_block(seq, x0, x1)
    x0.retain_grad(); x1 is not None and x1.retain_grad()
    y = tr.ConvFn.apply(x0, x1, seq[0].weight, seq[0].bias, False); y.retain_grad()
    r = tr._bn(seq[1], y, True); r.retain_grad()
    c = tr.ConvFn.apply(r, None, seq[3].weight, seq[3].bias, False); c.retain_grad()
    o = tr._bn(seq[4], c, True); o.retain_grad()
    keep.update(x0=x0, x1=x1, conv0=y, relu1=r, conv3=c, out=o)
    return o
tr._block = hb
o1, o2 = net(x.cuda())
pred = tr.gather_values(o1, coords[:, 0].cuda(), coords[:, 1:].cuda())
net.loss(pred, gains.cuda(), o2, gt.cuda()).backward()
def nchw(t): return t.permute(0, 3, 1, 2).cpu().double()
hg = {k: nchw(v.grad) for k, v in keep.items() if v is not None and k not in ("x0", "x1")}
ha = {k: nchw(v.detach()) for k, v in keep.items() if v is not None and k not in ("x0", "x1")}
if keep.get("x1") is not None:
    hg["x_in"] = torch.cat([nchw(keep["x0"].grad), nchw(keep["x1"].grad)], 1)
else:
    hg["x_in"] = nchw(keep["x0"].grad)[:, :g64["x_in"].shape[1]]
for k in ("out", "conv3", "relu1", "conv0", "x_in"):
    n = float(g64[k].norm())
    print(f"grad wrt {k:6s}: hip {float((hg[k]-g64[k]).norm())/n:.2e}  torch32 {float((g32[k]-g64[k]).norm())/n:.2e}   (|g| {n:.3e})", end="")
    if k in ha:
        na = float(a64[k].norm()); print(f"   activation err hip {float((ha[k]-a64[k]).norm())/na:.1e} t32 {float((a32[k]-a64[k]).norm())/na:.1e}")
    else: print()
# per-channel breakdown of the conv0 gradient error
e = (hg["conv0"] - g64["conv0"]).pow(2).sum((0, 2, 3)).sqrt(); r = g64["conv0"].pow(2).sum((0, 2, 3)).sqrt()
top = torch.argsort(e, descending=True)[:6]
var = a64["conv0"].var((0, 2, 3), unbiased=False)
for c in top.tolist(): print(f"  channel {c:3d}: err {float(e[c]):.2e} of |g_c| {float(r[c]):.2e}; var(conv0_c) {float(var[c]):.3e}")
