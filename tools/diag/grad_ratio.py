import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_gpu_training import _inputs, _ref_step, _hip_step
from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
w = make_nbp_state_dict(9)
x, coords, gains, gt2, sd = _inputs(64, w)
r1, r2, rl, rsd = _ref_step(sd, x, coords, gains, gt2)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
_, _, ql, qsd = _ref_step(sd64, x.double(), coords, gains.double(), gt2.double())
net, o1, o2, loss = _hip_step(sd, x, coords, gains, gt2)
rows = []
for name, p in net.named_parameters():
    ref64, ref32 = qsd[name].grad, rsd[name].grad
    e_hip = (p.grad.cpu().double() - ref64).abs().max().item()
    e_t32 = (ref32.double() - ref64).abs().max().item()
    scale = ref64.abs().max().item()
    rows.append((e_hip / max(e_t32, 1e-30), name, e_hip, e_t32, scale))
rows.sort(reverse=True)
for r in rows[:6]: print("%.2f %s e_hip %.3e e_t32 %.3e scale %.3e" % r)
