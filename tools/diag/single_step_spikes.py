"""Diagnostic (GPU box): per-step wall time of a single rollout (steps 40-100 of the bench scene), the steps that take > 4 ms.
   GC=0: Python's cyclic collector off."""
import gc, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.simulator import scene as sc
from nextbestpath_amd.simulator.mesh import make_maze_scene
from nextbestpath_amd.testers import nbp_planning as tp
from nextbestpath_amd.utility.synthetic import make_explorer_state_dict

if os.environ.get("GC") == "0":
    gc.disable()
dev = torch.device("cuda")
params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
tmp = tempfile.mkdtemp()
net = NBP(); net.load_state_dict(make_explorer_state_dict(9)); net = net.to(dev).eval()
make_maze_scene(os.path.join(tmp, "m"), seed=115, cells=10, size=6.0, height=1.2, tess=0.25)
ds = sc.SceneDataset(tmp, ["m"])
ro = tp.build_rollout(params, net, ds, (0, 0), dev, seed=23)
for _ in range(40):
    ro.step()
torch.cuda.synchronize()
gcs = []
gc.callbacks.append(lambda ph, info: gcs.append((ph, info["generation"], time.perf_counter())))
ts = []
t_all = time.perf_counter()
for i in range(60):
    t0 = time.perf_counter()
    ro.step()
    ts.append((time.perf_counter() - t0, t0))
torch.cuda.synchronize()
print(f"GC={os.environ.get('GC', '1')}: {60 / (time.perf_counter() - t_all):.1f} steps/s over steps 40-100; steps > 4 ms:",
      [(40 + i, round(1e3 * d, 1), [g for p, g, t in gcs if p == 'start' and t0 <= t <= t0 + d]) for i, (d, t0) in enumerate(ts) if d > 4e-3],
      "gen-2 collections:", sum(1 for p, g, t in gcs if p == "start" and g == 2))
