"""Diagnostic (GPU box): single-rollout steps/s (bench.py's `single_rollout_steps_per_s` stage alone) and the replanning share of
its window.   NBP_TUNING=1 NBP_STEP_OVERLAP=0 python tools/diag/single_rollout_ab.py   for the in-stream-order step."""
import os, sys, tempfile, time
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
make_maze_scene(os.path.join(tmp, "m"), seed=115, cells=10, size=6.0, height=1.2, tess=0.25)
ds = sc.SceneDataset(tmp, ["m"])
ro = tp.build_rollout(params, net, ds, (0, 0), dev, seed=23)
for _ in range(40):
    ro.step()
torch.cuda.synchronize()
for rep in range(3):
    r0 = ro.n_replans
    t0 = time.perf_counter()
    for _ in range(20):
        ro.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"overlap={tp._STEP_OVERLAP} window {rep}: {20 / dt:.1f} steps/s  ({ro.n_replans - r0} of 20 steps replanned)")
cov = ro.coverage_evolution(100)
print("coverage checksum", sum(cov), "cloud", int(ro.st.cloud_count.item()), "pose", tuple(ro.camera.cam_idx))
