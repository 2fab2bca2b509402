"""Diagnostic (GPU box): how many DISTINCT (channel, cell) keys do 64 / 1024 / 8192 consecutive cloud points fall on in the map
accumulation (mid-trajectory rollout state)?  Decides whether wave-level pre-aggregation could cut the LDS atomics."""
import os, sys, tempfile
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
make_maze_scene(os.path.join(tmp, "m0"), seed=100, cells=10, size=6.0, height=1.2, tess=0.25)
ro = tp.build_rollout(params, net, sc.SceneDataset(tmp, ["m0"]), (0, 0), dev, seed=8)
for _ in range(50):
    ro.step()
n = int(ro.st.cloud_count.item())
p = ro.st.cloud[:n].cpu().numpy()
pose = ro.camera.get_pose_from_idx(ro.camera.cam_idx)[0]
sc_ = np.float32(256 / 80.0)
i0 = np.rint((-(p[:, 2] - np.float32(pose[2])) + 40) * sc_).astype(np.int64)
i1 = np.rint((-(p[:, 0] - np.float32(pose[0])) + 40) * sc_).astype(np.int64)
ok = (i0 >= 0) & (i0 < 256) & (i1 >= 0) & (i1 < 256)
yb = np.asarray(ro.y_bins, np.float32)[:-1]
ch = np.clip((p[:, 1][:, None] > yb[None]).sum(1) - 1, -1, 4); ch[ch < 0] = 4; ch[ch > 3] = 4
key = np.where(ok, (ch * 65536 + i0 * 256 + i1), -1)
print(f"{n} points, {ok.mean():.3f} inside the window, {len(np.unique(key))} distinct keys overall")
for g in (64, 256, 1024, 8192):
    m = n // g * g
    k = key[:m].reshape(-1, g)
    d = np.array([len(np.unique(r)) for r in k[:: max(1, len(k) // 2000)]])
    srt = np.sort(k[:: max(1, len(k) // 2000)], 1)
    print(f"groups of {g}: distinct keys mean {d.mean():.1f} median {np.median(d):.0f} max {d.max()}  (={d.mean()/g:.3f} of the points)")
# adjacent equal
print("P(key[i] == key[i-1]) =", float((key[1:] == key[:-1]).mean()))
