"""Diagnostic (GPU box): whole-network distance to fp64 on a rollout-derived input at B = 1 and B = 12 (the bench's group batch:
fewer split-K slices, longer accumulation chains), split path and fp32 pipe, beside stock torch CPU fp32."""
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
every = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for s in range(N_STEPS):
    hip_ro.pre()
    with torch.no_grad():
        o1, o2 = hip_ro.nbp(hip_ro.st.net_in)
    if s % every == every - 1 or s == N_STEPS - 1:
        x = hip_ro.st.net_in.cpu()
        with torch.no_grad():
            c1, _ = nbp_net.nbp_forward(ora.sd, x)
            d1, _ = nbp_net.nbp_forward(sd64, x.double())
        rng = d1.abs().max().item()
        e_cpu = (c1.double() - d1).abs()
        line = f"step {s:2d} range {rng:8.1f} torch32 mean {e_cpu.mean().item()/rng:.2e} max {e_cpu.max().item()/rng:.2e}"
        for prec in os.environ.get("NET_ERR_PATHS", "fp32_split,fp32").split(","):
            hip_ro.nbp.conv_precision = prec
            for B in [int(b) for b in os.environ.get("NET_ERR_BATCHES", "1,12").split(",")]:
                with torch.no_grad():
                    h1, _ = hip_ro.nbp(hip_ro.st.net_in.expand(B, -1, -1, -1).contiguous())
                e = (h1[B - 1:B].cpu().double() - d1).abs()
                line += f" | {prec} B={B}: mean {e.mean().item()/rng:.2e} ({e.mean().item()/e_cpu.mean().item():.1f}x) max {e.max().item()/rng:.2e} ({e.max().item()/e_cpu.max().item():.1f}x)"
        hip_ro.nbp.conv_precision = "fp32_split"
        print(line, flush=True)
    hip_ro.plan_enqueue(o1, o2); torch.cuda.synchronize(); hip_ro.plan_finish(); hip_ro.post()
