"""GPU: the RCCL ("nccl" backend) side of SURVEY 8(e) / 8(f)4 on the ONE GPU a test box has -- a 1-rank process group.
Every collective of the product (the coverage all_gather of parallel_rollout.gather_results, the bucketed gradient all-reduce,
the batch-count / validation-loss / BatchNorm-buffer reductions of the trainer, bench.py's max-over-ranks time) goes through
`dist.init_process_group("nccl", ...)` on device tensors; the multi-rank *logic* is covered by the world_size-2 gloo tests
(tests/test_distributed_cpu.py, tests/test_gpu_rollout.py, tests/test_gpu_training_data.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env1():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NBP_DIST_BACKEND", None)
    return env


_SCRIPT = r'''
import json, sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from nextbestpath_amd import parallel_rollout as pr
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.trainers import train_nbp_model as tm
rank, world, local_rank = pr.init_distributed()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local_rank)
out = {}
# 1. the rollout path's one collective, on device tensors
runs = [(i, 0) for i in range(3)]
res = [{"run_id": i, "coverage": np.linspace(0, 0.2 * (i + 1), 7).astype(np.float32).tolist(), "X_cam_history": [[i, 0, 0]]} for i in range(3)]
assert pr.collective_device(dev).type == "cuda"
g = pr.gather_results(res, runs, rank, world, dev, 7)
out["gather"] = [(r["run_id"], r["final"], r["auc"], r["X_cam_history"]) for r in g]
# 2. the trainer's collectives: bucketed gradient all-reduce over the 200 MB of fp32 gradients, scalar reductions, BN buffers
torch.manual_seed(0)
net = NBP().to(dev)
gen = torch.Generator(device=dev).manual_seed(1)
for p in net.parameters():
    p.grad = torch.randn(p.shape, device=dev, generator=gen)
before = [p.grad.clone() for p in net.parameters()]
n_bytes = sum(p.grad.numel() * 4 for p in net.parameters())
torch.cuda.synchronize(); t0 = time.perf_counter()
tm.allreduce_gradients(net)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["allreduce_equal"] = all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
out["allreduce_bytes"] = n_bytes
out["allreduce_buckets"] = -(-n_bytes // tm.BUCKET_BYTES)
out["allreduce_ms"] = dt * 1e3
out["common_count"] = tm._common_count(5, dev)
out["mean_over_ranks"] = tm._mean_over_ranks(0.375, dev)
bufs = {k: v.clone() for k, v in net.named_buffers()}
tm.sync_buffers(net)
out["buffers_equal"] = all(torch.equal(v, bufs[k]) for k, v in net.named_buffers())
# 3. a raw RCCL all-reduce / all-gather of a bucket-sized device tensor (what a multi-rank step moves per bucket)
t = torch.ones(tm.BUCKET_BYTES // 4, device=dev)
dist.all_reduce(t)
parts = [torch.empty(1024, device=dev)]
dist.all_gather(parts, torch.arange(1024, device=dev, dtype=torch.float32))
out["raw"] = [float(t.sum().item()), float(parts[0].sum().item())]
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_rccl_one_rank_collectives(hip):
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=_env1())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    from nextbestpath_amd.utility.long_term_utils import compute_auc
    for rid, final, auc, hist in out["gather"]:
        cov = np.linspace(0, 0.2 * (rid + 1), 7).astype(np.float32)
        assert abs(final - cov[-1]) < 1e-7 and abs(auc - compute_auc(cov)) < 1e-6 and hist == [[rid, 0, 0]]
    assert [g[0] for g in out["gather"]] == [0, 1, 2]
    assert out["allreduce_equal"] and out["buffers_equal"]                   # world 1: sum / 1 is the identity, bit for bit
    assert out["allreduce_bytes"] > 190e6 and out["allreduce_buckets"] >= 3  # the real 200 MB of gradients, in 64 MB buckets
    assert out["common_count"] == 5 and out["mean_over_ranks"] == 0.375
    assert out["raw"] == [float((64 << 20) // 4), float(1023 * 1024 // 2)]


def test_entry_point_under_one_rank_torchrun_uses_rccl(hip, tmp_path):
    """`torchrun --nproc-per-node 1 test_nbp_planning.py`: the coverage gather runs over RCCL and the JSON equals the plain run."""
    from nextbestpath_amd.simulator.mesh import make_maze_scene
    data = tmp_path / "ds"
    make_maze_scene(str(data / "maze_00"), seed=3, cells=4, size=2.4, height=1.2, tess=0.3)
    make_maze_scene(str(data / "maze_01"), seed=4, cells=4, size=2.4, height=1.2, tess=0.3)
    cfg = {"numGPU": 0, "dataset_path": str(data), "test_scenes": [], "params_name": "macarons_default_training_config.json",
           "model_name": "x.pth", "results_json_name": "out_test_rccl1.json", "test_resolution": 0.05,
           "use_perfect_depth_map": True, "compute_collision": False, "load_json": False, "random_seed": 8, "torch_seed": 9,
           "nbp_weights": "./weights/none.pth"}
    cfg_path = os.path.join(ROOT, "configs/test/_pytest_rccl1.json")
    try:
        with open(cfg_path, "w") as fh:
            json.dump(cfg, fh)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NBP_TRACE_DIST="1")
        env.pop("NBP_DIST_BACKEND", None)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                            "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "test_nbp_planning.py"), "-c",
                            "_pytest_rccl1.json", "--n-poses", "4"], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert "[dist] backend nccl world 1" in r.stdout, r.stdout[-2000:]
        one = json.load(open(os.path.join(ROOT, "data", "out_test_rccl1.json")))
        cfg["results_json_name"] = "out_test_rccl0.json"
        with open(cfg_path, "w") as fh:
            json.dump(cfg, fh)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "test_nbp_planning.py"), "-c", "_pytest_rccl1.json", "--n-poses", "4"],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        zero = json.load(open(os.path.join(ROOT, "data", "out_test_rccl0.json")))
        assert sorted(one) == sorted(zero) == ["maze_00", "maze_01"]
        for scene in one:
            assert one[scene]["0"]["coverage"] == zero[scene]["0"]["coverage"] and one[scene]["0"]["auc"] == zero[scene]["0"]["auc"]
    finally:
        if os.path.exists(cfg_path):
            os.remove(cfg_path)


def test_bench_under_one_rank_torchrun_uses_rccl(hip):
    """bench.py as the driver launches it for N > 1, with N = 1: process group on RCCL, barrier + max-over-ranks reduction on
    device tensors, one JSON line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NBP_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                        "--warmup", "1", "--advance", "2", "--rollouts-per-gpu", "2", "--no-live-traffic", "--no-extra-stages",
                        "--no-cpu-baseline", "--strong-scenes", "2", "--strong-advance", "1"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["distributed"]["backend"] == "nccl" and d["distributed"]["world"] == 1
    assert d["tuning"]["numerics_affecting"] == [] and d["power"]["timed_region"] is None or d["power"]["timed_region"]["board_power_w_mean"] > 0
