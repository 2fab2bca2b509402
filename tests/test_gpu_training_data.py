"""GPU: GT obstacle label kernel vs its restatement (bit-exact), trajectory collection into the replay store,
and the collection-driven train_nbp entry point on a tiny procedural dataset."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle.slice_raster import slice_obstacle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from nextbestpath_amd.simulator.mesh import make_maze_scene
    d = tmp_path_factory.mktemp("synth_train")
    for i in range(2):
        make_maze_scene(str(d / f"maze_{i:02d}"), seed=10 + i, cells=6, size=4.8, height=1.2, tess=0.4, hull="shell")
    return str(d)


def _mesh(dataset):
    from nextbestpath_amd.simulator import scene as sc
    ds = sc.SceneDataset(dataset)
    sd = ds[0]
    return sc.load_scene(os.path.join(ds.data_path, sd["scene_name"], sd["obj_name"]), 10.0, torch.device("cuda"))


@pytest.mark.parametrize("pose", [(0.0, 3.3, 0.0), (7.3, 5.0, -11.9), (-30.0, 1.0, 25.0), (100.0, 3.0, 0.0)])
def test_slice_obstacle_matches_restatement(hip, dataset, pose):
    from nextbestpath_amd.utility import hipops
    mesh = _mesh(dataset)
    got = hipops.slice_obstacle(mesh.verts, mesh.faces, pose[1], pose[0], pose[2]).cpu().numpy()
    ref = slice_obstacle(mesh.verts_host, mesh.faces_host, pose[1], pose[0], pose[2])
    assert np.array_equal(got, ref)
    if pose[0] < 50:
        assert got.sum() > 50            # the maze walls are in view
    else:
        assert got.sum() == 0            # camera far outside: empty window
    got512 = hipops.slice_obstacle(mesh.verts, mesh.faces, pose[1], pose[0], pose[2], 512, (-80, 80)).cpu().numpy()
    assert np.array_equal(got512, slice_obstacle(mesh.verts_host, mesh.faces_host, pose[1], pose[0], pose[2], 512, -80, 80))


@pytest.mark.parametrize("pose", [(0.0, 3.3, 0.0), (7.3, 5.0, -11.9), (-30.0, 1.0, 25.0), (100.0, 3.0, 0.0)])
def test_slice_obstacle_on_the_reference_pixel_grid_matches_restatement(hip, dataset, pose):
    """nbp_slice_obstacle_fig_f32 (the default of nbp_utils.get_binary_obstacle_array since round 6) == oracle, bit for bit."""
    from nextbestpath_amd.utility import hipops, nbp_utils
    from oracle.slice_raster import slice_obstacle_fig
    mesh = _mesh(dataset)
    got = nbp_utils.get_binary_obstacle_array(mesh, (pose[0], pose[1], pose[2], 0.0, 0.0)).cpu().numpy()
    assert np.array_equal(got, slice_obstacle_fig(mesh.verts_host, mesh.faces_host, pose[1], pose[0], pose[2]))
    assert (got.sum() > 50) if pose[0] < 50 else (got.sum() == 0)
    old = nbp_utils.get_binary_obstacle_array(mesh, (pose[0], pose[1], pose[2], 0.0, 0.0), reference_label_semantics=False).cpu().numpy()
    assert np.array_equal(old, slice_obstacle(mesh.verts_host, mesh.faces_host, pose[1], pose[0], pose[2]))
    got512 = hipops.slice_obstacle_fig(mesh.verts, mesh.faces, pose[1], pose[0], pose[2], 512, 160.0).cpu().numpy()
    assert np.array_equal(got512, slice_obstacle_fig(mesh.verts_host, mesh.faces_host, pose[1], pose[0], pose[2], 512, 160.0))


def test_slice_obstacle_kernel_vs_reference_golden(hip, golden_dir):
    """The HIP label against the labels the reference's own matplotlib / PIL stage produced (tests/golden/obstacle_label.npz)."""
    from nextbestpath_amd.utility import hipops
    from test_training_data_cpu import _label_agreement
    g = np.load(os.path.join(golden_dir, "obstacle_label.npz"))
    verts = torch.from_numpy(g["verts"]).cuda()
    faces = torch.from_numpy(g["faces"].astype(np.int32)).cuda()
    for pose, packed in zip(g["poses"], g["labels"]):
        ref = np.unpackbits(packed)[:256 * 256].reshape(256, 256)
        got = hipops.slice_obstacle_fig(verts, faces, pose[1], pose[0], pose[2]).cpu().numpy()
        iou, r1, o1, r2, o2 = _label_agreement(got, ref)
        assert r2 == 0 and o2 == 0 and r1 <= 0.002 and o1 <= 0.002 and iou >= 0.7, (pose, iou, r1, o1, r2, o2)


def test_trajectory_collection_fills_store(hip, dataset, tmp_path):
    from nextbestpath_amd.networks.nbp_model import NBP
    from nextbestpath_amd.simulator import scene as sc
    from nextbestpath_amd.testers import nbp_planning as tp
    from nextbestpath_amd.utility import nbp_utils as nu
    from nextbestpath_amd.utility.synthetic import make_explorer_state_dict
    params = tp.load_params(os.path.join(ROOT, "configs/macarons/macarons_default_training_config.json"))
    net = NBP()
    net.load_state_dict(make_explorer_state_dict(9))
    net = net.cuda().eval()
    env = nu.LogEnv(str(tmp_path / "db"))
    cov = []
    n = nu.trajectory_collection(params, 0, sc.SceneDataset(dataset), env, (256, 256), (64, 64), (-40, 40), net, cov, None,
                                 torch.device("cuda"), n_poses=60, n_gt_points=8000)
    assert n > 0 and env.entries() == n
    recs = [nu.unpack_record(v) for _, v in env.items()]
    for r in recs:
        assert r["current_model_input"].shape == (1, 5, 256, 256) and r["current_model_input"].dtype == np.float32
        assert r["current_gt_2d_layout"].shape == (1, 1, 256, 256)
        assert set(np.unique(r["current_gt_2d_layout"])) <= {0.0, 1.0} and r["current_gt_2d_layout"].sum() > 0
        px, g = r["target_value_map_pixel"], r["actual_coverage_gain"]
        assert px.dtype == np.int64 and px.ndim == 2 and px.shape[1] == 3 and len(px) == len(g) >= 1
        assert (px[:, 0] >= 0).all() and (px[:, 0] < 8).all() and (px[:, 1:] >= 0).all() and (px[:, 1:] < 64).all()
        assert (g >= 0).all() and np.isfinite(g).all()
        assert r["current_model_input"][0, 4].sum() > 0         # trajectory channel holds the camera history
        assert r["current_model_input"][0, :4].sum() > 0        # slab maps hold the accumulated cloud
    # thin wall traces, not filled regions
    assert all(r["current_gt_2d_layout"].mean() < 0.2 for r in recs)
    # a later pose of the same path reached with more coverage gives a positive gain somewhere
    assert max(float(r["actual_coverage_gain"].max()) for r in recs) > 0


def test_train_nbp_entry_point_with_collection(hip, dataset, tmp_path):
    cfg = json.load(open(os.path.join(ROOT, "configs/nbp/nbp_default_training_config.json")))
    cfg["_data"]["data_path"] = dataset
    cfg["_scene_management"]["n_gt_surface_points"] = 8000
    cfg["_nbp"].update({"nbp_model_name": "nbp_t", "nbp_batch_size": 4, "epochs": 1, "inner_epochs": 1, "n_validation": 4,
                        "n_collect_poses": 60, "output_dir": str(tmp_path / "w"), "collect": True})
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from nextbestpath_amd.testers.nbp_planning import load_params\n"
            "from nextbestpath_amd.trainers.train_nbp_model import run_training_nbp\n"
            "h = run_training_nbp(load_params(%r)); print('HIST', h)\n") % (ROOT, str(path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    hist = json.load(open(tmp_path / "w" / "loss.json"))
    assert "1" in hist and np.isfinite(hist["1"]["training_loss"]) and np.isfinite(hist["1"]["validation_loss"])
    assert os.path.exists(tmp_path / "w" / "nbp_t_best_val.pth")


_DDP_TRAIN = r"""
import json, os, sys, torch
sys.path.insert(0, sys.argv[1])
from nextbestpath_amd.testers.nbp_planning import load_params
from nextbestpath_amd.trainers import train_nbp_model as T
p = load_params(sys.argv[2])
orig = T.train_nbp
state = {}
def spy(db, params, optimizer, nbp, *a, **k):
    state["nbp"] = nbp
    return orig(db, params, optimizer, nbp, *a, **k)
T.train_nbp = spy
T.run_training_nbp(p)
import torch.distributed as dist
if "nbp" in state:
    chk = torch.cat([q.detach().double().flatten() for q in state["nbp"].parameters()]).cpu()
    print("CHK", dist.get_rank(), f"{chk.sum().item():.10e}", f"{chk.abs().sum().item():.10e}", flush=True)
"""


def test_two_rank_collection_and_training(hip, dataset, tmp_path):
    """torchrun x2 with collection: each rank explores its own scene into its own store, validation is split off per rank,
    the training epoch averages gradients; both ranks finish (no rank is left waiting in a collective)."""
    cfg = json.load(open(os.path.join(ROOT, "configs/nbp/nbp_default_training_config.json")))
    cfg["_data"]["data_path"] = dataset
    cfg["_scene_management"]["n_gt_surface_points"] = 8000
    cfg["_nbp"].update({"nbp_model_name": "nbp_ddp_c", "nbp_batch_size": 4, "epochs": 1, "inner_epochs": 1, "n_validation": 2,
                        "n_collect_poses": 60, "output_dir": str(tmp_path / "w"), "collect": True})
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    script = tmp_path / "ddp_train.py"
    script.write_text(_DDP_TRAIN)
    env = dict(os.environ, NBP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29643", str(script), ROOT, str(path)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    assert out.stdout.count("collected") >= 4                      # 2 ranks x epochs 0 and 1
    lines = [l.split()[2:] for l in out.stdout.splitlines() if l.startswith("CHK")]
    if lines:                                                      # training ran: replicas identical
        assert len(lines) == 2 and lines[0] == lines[1], out.stdout


def test_two_rank_training_keeps_replicas_identical(hip, tmp_path):
    """torchrun x2 (gloo rendezvous, both ranks on the one GPU): different data per rank, averaged gradients,
    identical parameters afterwards."""
    cfg = json.load(open(os.path.join(ROOT, "configs/nbp/nbp_default_training_config.json")))
    cfg["_nbp"].update({"nbp_model_name": "nbp_ddp", "nbp_batch_size": 2, "grid_size": 64, "epochs": 1, "inner_epochs": 1,
                        "samples_per_epoch": 4, "n_validation_synthetic": 2, "output_dir": str(tmp_path / "w"),
                        "collect": False})
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    script = tmp_path / "ddp_train.py"
    script.write_text(_DDP_TRAIN)
    env = dict(os.environ, NBP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", str(script), ROOT, str(path)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    lines = [l.split()[2:] for l in out.stdout.splitlines() if l.startswith("CHK")]
    assert len(lines) == 2 and lines[0] == lines[1], out.stdout
    assert os.path.exists(tmp_path / "w" / "loss.json")
