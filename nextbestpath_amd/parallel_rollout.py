"""Scene-parallel sharding of independent rollouts.

Each (scene, start pose) rollout is independent (next_best_path/testers/nbp_planning.py:414-467
runs them in nested loops on one GPU), so rank r takes runs r, r+W, r+2W, ... and the only
collective is ONE all_gather of a padded fp32 tensor [runs_per_rank, n_poses + 3] holding
(run id, final coverage, AUC, coverage curve) -- a few KB over RCCL/xGMI, latency bound.  One
process per GPU; backend "nccl" (= RCCL on ROCm) with CUDA tensors, "gloo" on CPU-only hosts
(that path is what the world_size-2 CPU tests exercise)."""
from __future__ import annotations

import os

import numpy as np
import torch


def init_distributed():
    """(rank, world, local_rank); initialises torch.distributed when launched under torchrun -- also with ONE rank
    (`torchrun --nproc-per-node 1`): the collectives then run through RCCL on the single GPU, which is how the `nccl`
    path is exercised on a 1-GPU box (tests/test_gpu_rccl.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or under_torchrun():
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("NBP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:       # gloo: CPU-only hosts, or several ranks sharing one GPU (NBP_DIST_BACKEND=gloo) in tests
                dist.init_process_group("gloo")
            if rank == 0 and os.environ.get("NBP_TRACE_DIST"):
                print(f"[dist] backend {dist.get_backend()} world {dist.get_world_size()}", flush=True)
    if torch.cuda.is_available():
        local_rank = local_rank % torch.cuda.device_count()
    return rank, world, local_rank


def under_torchrun():
    return all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT"))


def group_is_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def collective_device(device):
    """Tensors of the gather live on the GPU for RCCL and on the host for gloo."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def shard(runs, rank, world):
    return list(runs)[rank::world]


def runs_per_rank(n_runs, world):
    return (n_runs + world - 1) // world


def pack_results(results, runs, n_poses, rows):
    """Local results -> padded [rows, n_poses + 3] fp32 (run id, final coverage, AUC, curve); pad rows = -1."""
    from .utility.long_term_utils import compute_auc
    t = np.full((rows, n_poses + 3), -1.0, np.float32)
    for j, r in enumerate(results):
        cov = np.asarray(r["coverage"], np.float32)
        t[j, 0] = r["run_id"]
        t[j, 1] = cov[-1]
        t[j, 2] = compute_auc(cov)
        t[j, 3:3 + len(cov)] = cov
    return t


def gather_results(results, runs, rank, world, device, n_poses):
    """All ranks call this; every rank gets the list of {run_id, coverage, final, auc, scene, start}
    in run order (only the coverage metrics travel; pose histories stay in the per-rank results)."""
    rows = runs_per_rank(len(runs), world)
    collective = world > 1 or group_is_up()          # a 1-rank process group still goes through the backend
    local = torch.from_numpy(pack_results(results, runs, n_poses, rows)).to(collective_device(device) if collective else device)
    if collective:
        import torch.distributed as dist
        buf = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(buf, local)
        allt = torch.stack(buf).cpu().numpy().reshape(-1, n_poses + 3)
    else:
        allt = local.cpu().numpy()
    by_id = {int(r["run_id"]): r for r in results}
    out = []
    for row in allt:
        if row[0] < 0:
            continue
        rid = int(row[0])
        rec = dict(by_id[rid]) if rid in by_id else {}
        rec.update(run_id=rid, coverage=[float(v) for v in row[3:3 + n_poses]], final=float(row[1]), auc=float(row[2]))
        out.append(rec)
    out.sort(key=lambda r: r["run_id"])
    return out
