"""Diagnostic (GPU box): wall time of every lock-step of R concurrent rollouts over a whole trajectory (steps 5-100): mean rate and the lock-steps that
take > 1.5 x the median (full garbage collections, re-captures, allocator stalls ...).   R=48 NBP_TUNING=1 NBP_GC_FREEZE=0 python tools/diag/lockstep_spikes.py"""
import gc, os, sys, tempfile, time
import torch
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
R = int(os.environ.get("R", "48"))
ros = []
for k in range(R):
    make_maze_scene(os.path.join(tmp, f"m{k}"), seed=100 + k, cells=10, size=6.0, height=1.2, tess=0.25)
    ds = sc.SceneDataset(tmp, [f"m{k}"])
    ros.append(tp.build_rollout(params, net, ds, (0, 0), dev, seed=8 + k))
multi = tp.MultiRollout(ros, net, dev)
for _ in range(5):
    multi.step()
multi.flush(); torch.cuda.synchronize()
gcs = []
gc.callbacks.append(lambda ph, info: gcs.append((ph, info["generation"], time.perf_counter())))
ts = []
t_all = time.perf_counter()
for i in range(95):
    t0 = time.perf_counter()
    multi.step()
    ts.append((time.perf_counter() - t0, t0))
multi.flush(); torch.cuda.synchronize()
dt = time.perf_counter() - t_all
med = sorted(d for d, _ in ts)[len(ts) // 2]
spikes = [(5 + i, round(1e3 * d, 1), [g for p, g, t in gcs if p == "start" and t0 <= t <= t0 + d]) for i, (d, t0) in enumerate(ts) if d > 1.5 * med]
print(f"R={R} NBP_GC_FREEZE={os.environ.get('NBP_GC_FREEZE', '1')}: {95 * R / dt:.1f} steps/s over lock-steps 5-100 (median lock-step {1e3 * med:.2f} ms = "
      f"{R / med:.1f} steps/s); lock-steps > 1.5 x median (step, ms, gc generations):", spikes, "full collections:", sum(1 for p, g, t in gcs if p == "start" and g == 2))
