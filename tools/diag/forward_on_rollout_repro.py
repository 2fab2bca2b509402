"""Diagnostic (GPU box): at every step of a HIP rollout run the NBP forward twice on the step's own input and compare the two results
bit for bit, and the first against an fp64 CPU evaluation -- looks for transient errors (races, stale scratch) that a fixed-input
loop does not show."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_rollout_parity import _both_rollouts
from oracle import nbp_net
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.set_num_threads(64)
for rep in range(reps):
    tmp = tempfile.mkdtemp()
    hip_ro, ora, mesh = _both_rollouts(tmp, cells=8, size=4.8, tess=0.3, scene_seed=scene, seed=5 + scene)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in ora.sd.items()}
    for s in range(steps):
        hip_ro.pre()
        with torch.no_grad():
            o1, o2 = hip_ro.nbp(hip_ro.st.net_in)
            a1, a2 = o1.clone(), o2.clone()
            x = hip_ro.st.net_in.clone()
            b1, b2 = hip_ro.nbp(x)
            same = torch.equal(a1, b1) and torch.equal(a2, b2)
            d1, d2 = nbp_net.nbp_forward(sd64, x.cpu().double())
            c1, c2 = nbp_net.nbp_forward(ora.sd, x.cpu())
            hip_ro.nbp.conv_precision = "fp32"
            p1, p2 = hip_ro.nbp(x)
            hip_ro.nbp.conv_precision = "fp32_split"
        ec, ep = (c1.double() - d1).abs(), (p1.cpu().double() - d1).abs()
        rng = float(d1.abs().max())
        e = (a1.cpu().double() - d1).abs()
        flag = "" if (same and e.max() <= 5e-4 * rng and e.mean() <= 2e-6 * rng) else "   <-- !!"
        print(f"rep {rep} step {s:2d}: second forward identical {same}; vs fp64 max {e.max().item():.3e} mean {e.mean().item():.3e} of range {rng:.1f} | torch cpu fp32: max {ec.max().item():.3e} mean {ec.mean().item():.3e} | hip fp32 pipe: max {ep.max().item():.3e} mean {ep.mean().item():.3e}{flag}", flush=True)
        hip_ro.plan_enqueue(o1, o2); torch.cuda.synchronize(); hip_ro.plan_finish(); hip_ro.post()
