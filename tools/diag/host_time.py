"""Diagnostic (GPU box): where a lock-step of MultiRollout goes -- host busy time vs time blocked on the GPU."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.simulator import scene as sc
from nextbestpath_amd.simulator.mesh import make_maze_scene
from nextbestpath_amd.testers import nbp_planning as tp
from nextbestpath_amd.utility.synthetic import make_explorer_state_dict

dev = torch.device("cuda")
params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
tmp = tempfile.mkdtemp()
net = NBP(); net.load_state_dict(make_explorer_state_dict(9)); net = net.to(dev).eval()
R = int(os.environ.get("R", "8"))
ros = []
for k in range(R):
    make_maze_scene(os.path.join(tmp, f"m{k}"), seed=100 + k, cells=10, size=6.0, height=1.2, tess=0.25)
    ds = sc.SceneDataset(tmp, [f"m{k}"])
    ros.append(tp.build_rollout(params, net, ds, (0, 0), dev, seed=8 + k))
multi = tp.MultiRollout(ros, net, dev)
for _ in range(45):
    multi.step()
multi.flush(); torch.cuda.synchronize()
blocked = [0.0]
orig = torch.cuda.Event.synchronize
def timed_sync(self):
    t = time.perf_counter(); orig(self); blocked[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed_sync
parts = {"pre": 0.0, "fwd_launch": 0.0, "plan_enqueue": 0.0, "plan_finish": 0.0, "post": 0.0, "_pre_group": 0.0, "_post_group": 0.0}
for cls, name in ((tp.Rollout, "pre"), (tp.Rollout, "plan_enqueue"), (tp.Rollout, "plan_finish"), (tp.Rollout, "post"),
                  (tp.MultiRollout, "_pre_group"), (tp.MultiRollout, "_post_group")):
    f = getattr(cls, name)
    def wrap(f=f, name=name):
        def g(self, *a, **k):
            t = time.perf_counter(); r = f(self, *a, **k); parts[name] += time.perf_counter() - t; return r
        return g
    setattr(cls, name, wrap())
fwd = net.forward
mf = tp.MultiRollout._forward
def fwd_t(self, x):
    t = time.perf_counter(); r = mf(self, x); parts["fwd_launch"] += time.perf_counter() - t; return r
tp.MultiRollout._forward = fwd_t
n = 20
t0 = time.perf_counter()
for _ in range(n):
    multi.step()
multi.flush(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"R={R}: {n*R/dt:.1f} steps/s, lock-step {dt/n*1e3:.2f} ms; blocked on events {blocked[0]/n*1e3:.2f} ms/lock-step; "
      f"host parts per lock-step (ms): " + ", ".join(f"{k} {v/n*1e3:.2f}" for k, v in parts.items()))
if os.environ.get('HOST_TIME_NO_FWD'):
    sys.exit(0)
x4 = multi.net_in[0]
for B in (1, 2, 4, 8):
    x = torch.cat(multi.net_in)[:B].contiguous() if B <= R else None
    if x is None: continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fwd(x); e0.record()
    for _ in range(10): fwd(x)
    e1.record(); torch.cuda.synchronize()
    print(f"forward B={B}: {e0.elapsed_time(e1)/10:.3f} ms")
