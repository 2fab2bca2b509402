"""Diagnostic (GPU box): trajectory_collection (next_best_path/utility/nbp_utils.py:470-852) on a synthetic scene set: poses/s, records/s and the host's
top functions (cProfile)."""
import cProfile, os, pstats, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.simulator import scene as sc
from nextbestpath_amd.simulator.mesh import make_maze_scene
from nextbestpath_amd.testers import nbp_planning as tp
from nextbestpath_amd.utility import nbp_utils as nu
from nextbestpath_amd.utility.synthetic import make_explorer_state_dict

params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
tmp = tempfile.mkdtemp()
names = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    make_maze_scene(os.path.join(tmp, f"m{k}"), seed=300 + k, cells=10, size=6.0, height=1.2, tess=0.25)
    names.append(f"m{k}")
net = NBP(); net.load_state_dict(make_explorer_state_dict(9)); net = net.cuda().eval()
env = nu.open_experience_db(os.path.join(tmp, "db"))
cov = []
nu.trajectory_collection(params, 0, sc.SceneDataset(tmp, names[:1]), env, (256, 256), (64, 64), (-40, 40), net, cov, None, torch.device("cuda"),
                         n_poses=20, n_gt_points=20000)       # warm
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = env.entries()
pr.enable()
n = nu.trajectory_collection(params, 0, sc.SceneDataset(tmp, names), env, (256, 256), (64, 64), (-40, 40), net, cov, None, torch.device("cuda"),
                             n_poses=100, n_gt_points=20000)
pr.disable()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{len(names)} scenes x 100 poses: {dt:.2f} s = {100 * len(names) / dt:.1f} poses/s, {env.entries() - n0} records stored ({(env.entries() - n0) / dt:.1f}/s), container {type(env).__name__}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
