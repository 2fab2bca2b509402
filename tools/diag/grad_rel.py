"""Diagnostic (GPU box): relative L2 error of every parameter gradient vs an fp64 evaluation -- HIP, torch-CPU fp32, and the
fp64 evaluation with 1-ulp-perturbed conv weights (the gradient's own sensitivity), in network order."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_training import _ref_step, _hip_step
from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
g = np.load("/root/repo/tests/golden/nbp_train_S128B4.npz")
sd = make_nbp_state_dict(9)
x, coords, gains, gt = (torch.from_numpy(g[k]) for k in ("x", "coords", "gains", "gt"))
torch.set_num_threads(32)
_, _, rl, rsd = _ref_step(sd, x, coords, gains, gt)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
_, _, ql, qsd = _ref_step(sd64, x.double(), coords, gains.double(), gt.double())
gen = torch.Generator().manual_seed(1)
sd64p = {k: (v * (1 + 1e-7 * torch.randn(v.shape, generator=gen, dtype=torch.float64)) if v.dim() == 4 else v) for k, v in sd64.items()}
_, _, _, psd = _ref_step(sd64p, x.double(), coords, gains.double(), gt.double())
net, o1, o2, loss = _hip_step(sd, x, coords, gains, gt)
for name, p in net.named_parameters():
    if not name.endswith("weight") or p.dim() != 4: continue
    r64 = qsd[name].grad; n = float(r64.norm())
    if n < 1e-9: continue
    eh = float((p.grad.cpu().double() - r64).norm()) / n
    et = float((rsd[name].grad.double() - r64).norm()) / n
    eu = float((psd[name].grad - r64).norm()) / n
    print(f"{name:28s} hip {eh:.2e}  torch32 {et:.2e}  ulp-sens {eu:.2e}  {'<<<' if eh > 3*max(et,eu)+2e-3 else ''}")
