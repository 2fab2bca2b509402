"""Diagnostic (GPU box, under `rocprofv3 --kernel-trace`): one rollout, a few steps; a marker kernel (torch.zeros(7777)) separates
the steps so that the kernel sequence of one step can be read from the trace."""
import os, sys, tempfile
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
make_maze_scene(os.path.join(tmp, "m0"), seed=100, cells=10, size=6.0, height=1.2, tess=0.25)
ro = tp.build_rollout(params, net, sc.SceneDataset(tmp, ["m0"]), (0, 0), dev, seed=8)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    ro.step()
    torch.full((7777,), 1.0, device=dev)
torch.cuda.synchronize()
