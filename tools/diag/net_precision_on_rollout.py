"""Diagnostic (GPU box): NBP forward on rollout-derived inputs -- HIP fp32 vs stock torch CPU fp32 vs torch CPU fp64."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_rollout_parity import _both_rollouts
from oracle import nbp_net
tmp = tempfile.mkdtemp()
hip_ro, ora, mesh = _both_rollouts(tmp, cells=8, size=4.8, tess=0.3, scene_seed=0, seed=5)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in ora.sd.items()}
N_STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for s in range(N_STEPS):
    hip_ro.pre()
    with torch.no_grad():
        o1, o2 = hip_ro.nbp(hip_ro.st.net_in)
        x = hip_ro.st.net_in.cpu()
        c1, c2 = nbp_net.nbp_forward(ora.sd, x)
        d1, d2 = nbp_net.nbp_forward(sd64, x.double())
        hip_ro.nbp.conv_precision = "fp32"
        p1, p2 = hip_ro.nbp(hip_ro.st.net_in)
        hip_ro.nbp.conv_precision = "fp32_split"
    hip_ro.plan_enqueue(o1, o2); torch.cuda.synchronize(); hip_ro.plan_finish(); hip_ro.post()
    o1, o2, p1 = o1.cpu().double(), o2.cpu().double(), p1.cpu().double()
    rng = d1.abs().max().item()
    print(f"step {s:2d} in.max {x.max().item():7.0f} out1 range {rng:9.3f} | hip-f64 {(o1-d1).abs().max().item():.3e} "
          f"(mean {(o1-d1).abs().mean().item():.2e}) hip fp32 pipe-f64 {(p1-d1).abs().max().item():.3e} cpu32-f64 {(c1.double()-d1).abs().max().item():.3e} (mean {(c1.double()-d1).abs().mean().item():.2e}) hip-cpu32 {(o1-c1.double()).abs().max().item():.3e} | out2: "
          f"hip-f64 {(o2-d2).abs().max().item():.2e} cpu32-f64 {(c2.double()-d2).abs().max().item():.2e}", flush=True)
