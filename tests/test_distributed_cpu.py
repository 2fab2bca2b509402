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


def test_single_process_path():
    from nextbestpath_amd import parallel_rollout as pr
    runs = [(0, 0), (1, 0)]
    res = [{"run_id": i, "coverage": [0.0, 0.5 * (i + 1)]} for i in range(2)]
    out = pr.gather_results(res, runs, 0, 1, torch.device("cpu"), 2)
    assert [o["final"] for o in out] == [0.5, 1.0]
