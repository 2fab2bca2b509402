"""world_size-2 gloo test (CPU) of the scene-parallel sharding + the single all_gather."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_runs, n_poses, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from nextbestpath_amd import parallel_rollout as pr
    r, w, _ = pr.init_distributed()
    runs = [(i // 2, i % 2) for i in range(n_runs)]
    mine = pr.shard(runs, r, w)
    results = []
    for run in mine:
        rid = runs.index(run)
        cov = np.linspace(0, 0.1 * (rid + 1), n_poses).astype(np.float32)
        results.append({"run_id": rid, "coverage": cov.tolist(), "X_cam_history": [[rid, 0, 0]]})
    out = pr.gather_results(results, runs, r, w, torch.device("cpu"), n_poses)
    q.put((r, [(o["run_id"], o["final"], o["auc"], o["coverage"][1]) for o in out], [m for m in mine]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    n_runs, n_poses, world = 5, 11, 2          # odd count: rank 1 has a padded row
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_runs, n_poses, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from nextbestpath_amd.utility.long_term_utils import compute_auc
    shards = {r: m for r, _, m in got}
    assert sorted(shards[0] + shards[1]) == [(i // 2, i % 2) for i in range(n_runs)]      # disjoint cover
    assert len(shards[0]) == 3 and len(shards[1]) == 2
    for _, rows, _ in got:                                                               # every rank sees all runs
        assert [r[0] for r in rows] == list(range(n_runs))
        for rid, final, auc, c1 in rows:
            cov = np.linspace(0, 0.1 * (rid + 1), n_poses).astype(np.float32)
            assert abs(final - cov[-1]) < 1e-6 and abs(c1 - cov[1]) < 1e-6
            assert abs(auc - compute_auc(cov)) < 1e-5


def _check_gather(n_runs, n_poses, world, starts_per_scene=1):
    """`world` gloo ranks shard an n_runs list (run i = scene i // starts, start i % starts) as test_nbp_planning does and gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, n_runs, n_poses, starts_per_scene, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from nextbestpath_amd.utility.long_term_utils import compute_auc
    runs = [(i // starts_per_scene, i % starts_per_scene) for i in range(n_runs)]
    shards = {r: m for r, _, m in got}
    assert sorted(sum((shards[r] for r in range(world)), [])) == runs                     # disjoint cover of the run list
    for r in range(world):
        assert shards[r] == runs[r::world]                                                # rank r takes runs r, r + W, ...
        assert len(shards[r]) == len(range(r, n_runs, world))
    for _, rows, _ in got:                                                               # every rank sees all runs, in run order
        assert [x[0] for x in rows] == list(range(n_runs))
        for rid, final, auc, c1 in rows:
            cov = np.linspace(0, 0.01 * (rid + 1), n_poses).astype(np.float32)
            assert abs(final - cov[-1]) < 1e-6 and abs(c1 - cov[1]) < 1e-6 and abs(auc - compute_auc(cov)) < 1e-5


def _worker_n(rank, world, port, n_runs, n_poses, starts, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from nextbestpath_amd import parallel_rollout as pr
    r, w, _ = pr.init_distributed()
    runs = [(i // starts, i % starts) for i in range(n_runs)]
    mine = pr.shard(runs, r, w)
    results = []
    for run in reversed(mine):                  # local completion order must not matter: rows carry their run id
        rid = runs.index(run)
        cov = np.linspace(0, 0.01 * (rid + 1), n_poses).astype(np.float32)
        results.append({"run_id": rid, "coverage": cov.tolist()})
    out = pr.gather_results(results, runs, r, w, torch.device("cpu"), n_poses)
    q.put((r, [(o["run_id"], o["final"], o["auc"], o["coverage"][1]) for o in out], [m for m in mine]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_world8_configs3_forty_scenes_five_per_rank():
    """BASELINE configs[3]: 40 AiMDoom_hard scenes over the 8 GPUs of a node, 5 per rank, no padded rows (the rank-indexing logic of
    the first 8-GPU run, on gloo: no 8-GPU node has been available to this build)."""
    _check_gather(40, 101, 8)


def test_world8_uneven_list_has_padded_rows():
    """37 runs over 8 ranks: ranks 0-4 hold 5 runs, ranks 5-7 hold 4 and one padded (-1) row each; two start poses per scene."""
    _check_gather(37, 21, 8, starts_per_scene=2)


def test_world8_fewer_runs_than_ranks():
    """3 runs over 8 ranks: five ranks have nothing but padding and still take part in the collective."""
    _check_gather(3, 5, 8)


def test_single_process_path():
    from nextbestpath_amd import parallel_rollout as pr
    runs = [(0, 0), (1, 0)]
    res = [{"run_id": i, "coverage": [0.0, 0.5 * (i + 1)]} for i in range(2)]
    out = pr.gather_results(res, runs, 0, 1, torch.device("cpu"), 2)
    assert [o["final"] for o in out] == [0.5, 1.0]


def _worker_one(port, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from nextbestpath_amd import parallel_rollout as pr
    from nextbestpath_amd.trainers import train_nbp_model as tm
    r, w, _ = pr.init_distributed()                      # a 1-rank torchrun environment still builds the process group
    ok = dist.is_initialized() and dist.get_world_size() == 1 and pr.group_is_up()
    res = [{"run_id": i, "coverage": [0.0, 0.25 * (i + 1)]} for i in range(2)]
    out = pr.gather_results(res, [(0, 0), (1, 0)], r, w, torch.device("cpu"), 2)      # through the backend's all_gather
    lin = torch.nn.Linear(3, 2)
    for p in lin.parameters():
        p.grad = torch.ones_like(p)
    tm.allreduce_gradients(lin)
    q.put((ok, [o["final"] for o in out], tm._common_count(4, torch.device("cpu")), tm._mean_over_ranks(0.5, torch.device("cpu")),
           [float(p.grad.sum()) for p in lin.parameters()]))
    dist.destroy_process_group()


def test_one_rank_group_still_runs_the_collectives():
    """`torchrun --nproc-per-node 1`: the group is built and every collective goes through the backend (on a GPU box this is the
    RCCL exercise of tests/test_gpu_rccl.py; here gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one, args=(_free_port(), q))
    p.start()
    ok, finals, cc, mo, gs = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and ok and finals == [0.25, 0.5] and cc == 4 and mo == 0.5 and gs == [6.0, 2.0]
