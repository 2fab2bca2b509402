"""Diagnostic (GPU box): relative L2 error vs fp64 of the pieces of one conv block's backward (HIP vs torch fp32)."""
import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from nextbestpath_amd.networks import training as tr
D = "cuda"
torch.manual_seed(0)
B, H, W, C, N = 4, 32, 32, 256, 256
x = torch.randn(B, C, H, W) * 0.5
w = torch.randn(N, C, 3, 3) * (3.0 / (C * 9)) ** 0.5
b = torch.randn(N) * 0.1
g, be = torch.rand(N) * 0.4 + 0.8, torch.randn(N) * 0.1
for label, gy in (("dense dy", torch.randn(B, N, H, W)),
                  ("sparse dy", torch.zeros(B, N, H, W).index_put_((torch.randint(0, B, (40,)), torch.randint(0, N, (40,)),
                                                                     torch.randint(0, H, (40,)), torch.randint(0, W, (40,))),
                                                                    torch.randn(40)))):
    def ref(dt):
        xx, ww, bb, gg, bbe = (t.to(dt).clone().requires_grad_(True) for t in (x, w, b, g, be))
        y = F.conv2d(xx, ww, bb, padding=1)
        z = F.relu(F.batch_norm(y, None, None, gg, bbe, True, 0.1, 1e-5))
        z.backward(gy.to(dt))
        return [t.grad.double() for t in (xx, ww, gg, bbe)]
    r64, r32 = ref(torch.float64), ref(torch.float32)
    xd = x.permute(0, 2, 3, 1).contiguous().to(D).requires_grad_(True)
    wd, bd, gd, bed = (t.to(D).requires_grad_(True) for t in (w, b, g, be))
    y = tr.ConvFn.apply(xd, None, wd, bd, False)
    rm, rv = torch.zeros(N, device=D), torch.ones(N, device=D)
    z = tr.BNFn.apply(y, gd, bed, rm, rv, 1e-5, 0.1, True)
    z.backward(gy.permute(0, 2, 3, 1).contiguous().to(D))
    hip = [xd.grad.permute(0, 3, 1, 2).cpu().double(), wd.grad.cpu().double(), gd.grad.cpu().double(), bed.grad.cpu().double()]
    for nm, h, a, c in zip(("dx", "dW", "dgamma", "dbeta"), hip, r32, r64):
        n = float(c.norm())
        print(f"{label:10s} {nm:7s} hip {float((h-c).norm())/n:.2e}  torch32 {float((a-c).norm())/n:.2e}")
    # pieces alone: conv only
    def refc(dt):
        xx, ww = x.to(dt).clone().requires_grad_(True), w.to(dt).clone().requires_grad_(True)
        F.conv2d(xx, ww, None, padding=1).backward(gy.to(dt)); return xx.grad.double(), ww.grad.double()
    c64, c32 = refc(torch.float64), refc(torch.float32)
    xd = x.permute(0, 2, 3, 1).contiguous().to(D).requires_grad_(True); wd = w.to(D).requires_grad_(True); bd = b.to(D).requires_grad_(True)
    tr.ConvFn.apply(xd, None, wd, bd, False).backward(gy.permute(0, 2, 3, 1).contiguous().to(D))
    for nm, h, a, c in zip(("conv dx", "conv dW"), (xd.grad.permute(0, 3, 1, 2).cpu().double(), wd.grad.cpu().double()), c32, c64):
        n = float(c.norm()); print(f"{label:10s} {nm:7s} hip {float((h-c).norm())/n:.2e}  torch32 {float((a-c).norm())/n:.2e}")
