"""Diagnostic (GPU box): cProfile of the host side of MultiRollout.step()."""
import cProfile, pstats, os, sys, tempfile
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
ros = []
for k in range(int(os.environ.get("R", "8"))):
    make_maze_scene(os.path.join(tmp, f"m{k}"), seed=100 + k, cells=10, size=6.0, height=1.2, tess=0.25)
    ros.append(tp.build_rollout(params, net, sc.SceneDataset(tmp, [f"m{k}"]), (0, 0), dev, seed=8 + k))
multi = tp.MultiRollout(ros, net, dev)
for _ in range(45):
    multi.step()
multi.flush(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    multi.step()
multi.flush(); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
