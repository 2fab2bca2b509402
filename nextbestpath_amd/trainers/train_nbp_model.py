"""NBP training -- host-side mirror of next_best_path/trainers/train_nbp_model.py:40-157 and the
training half of next_best_path/utility/nbp_utils.py (initialize_nbp :213-231, validation_model
:293-338, train_experience_data :340-395, train_nbp :430-468): same function names, same record
schema ('current_model_input', 'current_gt_2d_layout', 'target_value_map_pixel',
'actual_coverage_gain', 'pose_i'), same micro-batch / 8-step gradient accumulation / AdamW /
ReduceLROnPlateau logic.  Forward and backward run on the HIP kernels (networks/training.py).

The replay store and the trajectory collection that fills it live in utility/nbp_utils.py (SURVEY.md 8f
rank 2); ``make_synthetic_experiences`` synthesises records of the same schema for the config-3 benchmark.
Under torchrun every rank collects its own scenes into its own store, trains on its own
samples and the gradients are averaged with bucketed RCCL all-reduces before each optimizer step."""
from __future__ import annotations

import json
import os
import random

import numpy as np
import torch

from .. import _lib
from ..networks import training as tr
from ..networks.nbp_model import NBP


def make_optimizer(nbp, lr=0.001):
    """The reference's optimizer (nbp_utils.py:228): AdamW(lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight decay 0.01).  On the device the
    update of the 200 MB of parameters runs as torch's FUSED multi-tensor kernel (one pass over p, g, m, v instead of the ~10
    element-wise passes of the default `foreach` form: 3 % of a B = 32 training step); same update rule."""
    kw = dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    params = list(nbp.parameters())
    if params and all(p.is_cuda for p in params) and _lib.tune("NBP_TRAIN_FUSED_ADAMW", "1") == "1":
        try:
            return torch.optim.AdamW(params, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):     # a torch build without the fused kernel
            pass
    return torch.optim.AdamW(params, **kw)


def initialize_nbp(params, nbp, torch_seed=9, initialize=False, pretrained=False, ddp_rank=None):
    """ref nbp_utils.py:213-231: AdamW(lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight decay 0.01)."""
    return nbp, make_optimizer(nbp), 10000.0, 0


def make_synthetic_experiences(n, S=256, seed=0):
    """Replay records with the reference's schema and the input recipe of SURVEY.md 8d (config 3)."""
    from ..utility.synthetic import make_count_maps
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        k = int(rng.integers(1, 41))
        out.append({
            "current_model_input": make_count_maps(1, S, seed=seed * 100003 + i).numpy(),
            "current_gt_2d_layout": (rng.random((1, 1, S, S)) < 0.1).astype(np.float32),
            "target_value_map_pixel": np.stack([rng.integers(0, 8, k), rng.integers(0, S // 4, k),
                                                rng.integers(0, S // 4, k)], 1).astype(np.int64),
            "actual_coverage_gain": rng.uniform(0, 5, k).astype(np.float32),
            "pose_i": int(rng.integers(0, 101)),
        })
    return out


_STAGE_BATCHES = _lib.tune("NBP_TRAIN_STAGE_BATCHES", "1") == "1"      # 0: every batch through _collate's synchronous copies


def _collate(batch_data, device):
    # (np.concatenate, not torch.cat of 32 x 1.3 MB: on a 256-core host torch's intra-op thread pool made that concatenation cost
    # 50-600 ms per batch, tools/diag/train_loop_ab.py)
    xs = torch.from_numpy(np.concatenate([d["current_model_input"] for d in batch_data])).to(device)
    gt = torch.from_numpy(np.concatenate([d["current_gt_2d_layout"] for d in batch_data])).to(device)
    coords = [torch.from_numpy(np.copy(d["target_value_map_pixel"])) for d in batch_data]
    # the reference indexes predicted_value_map[b, c, x, y] (nbp_utils.py:379), which raises on a bad coordinate; the
    # device gather has no exception path, so replay records (possibly written elsewhere) are range-checked here
    V = xs.shape[-1] // 4
    for d, c in zip(batch_data, coords):
        if c.numel() and (c.min() < 0 or c[:, 0].max() >= 8 or c[:, 1:].max() >= V):
            raise IndexError(f"target_value_map_pixel out of range for an [8,{V},{V}] value map (pose_i={d.get('pose_i')})")
    gains = torch.cat([torch.from_numpy(np.copy(d["actual_coverage_gain"])) for d in batch_data]).to(device)
    sizes = torch.tensor([len(c) for c in coords])
    bidx = torch.repeat_interleave(torch.arange(len(coords)), sizes).to(device)
    return xs, gt, torch.cat(coords).to(device), gains, bidx


class _BatchStager:
    """The next batch's host-to-device copies on a stream of their own, from pinned staging buffers (two sets, used alternately), so
    that they run under the current batch's forward / backward instead of in front of the next one's: `.to(device)` from pageable
    memory is host-synchronous AND stream-ordered behind the kernels already queued (the reference pays exactly that,
    nbp_utils.py:352-355).  Same tensors on the device, bit for bit."""

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.pins = [{}, {}]
        self.done = [None, None]
        self.k = 0

    def _pin(self, slot, name, shape, dtype):
        numel = int(np.prod(shape)) if len(shape) else 1
        buf = self.pins[slot].get(name)
        if buf is None or buf.numel() < numel or buf.dtype != dtype:
            buf = self.pins[slot][name] = torch.empty(max(numel, 1), dtype=dtype).pin_memory()
        return buf[:numel].view(shape)

    def begin(self):
        """Next staging slot: waits (host) for the copies that last read its pinned buffers (two batches ago)."""
        self.slot = self.k & 1
        self.k += 1
        if self.done[self.slot] is not None:
            self.done[self.slot].synchronize()
        self.filled = {}

    def buffer(self, name, shape, dtype):
        """A pinned tensor of this slot to be filled by the caller (records are copied straight into it: no torch.cat, whose
        intra-op thread pool made a 42 MB concatenation cost 50 ms on the 256-core host)."""
        t = self._pin(self.slot, name, tuple(shape), dtype)
        self.filled[name] = t
        return t

    def commit(self):
        """-> ({name: device tensor}, event the consumer's stream must wait for)"""
        out = {}
        with torch.cuda.stream(self.stream):
            for name, p in self.filled.items():
                out[name] = torch.empty(p.shape, dtype=p.dtype, device=self.device)
                out[name].copy_(p, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.done[self.slot] = ev
        return out, ev


def _collate_staged(batch_data, device, stager):
    """_collate with the copies on the stager's stream: returns the same five tensors; the CURRENT stream waits for them."""
    n = len(batch_data)
    x0 = batch_data[0]["current_model_input"]
    S = x0.shape[-1]
    coords = [np.asarray(d["target_value_map_pixel"]) for d in batch_data]
    V = S // 4
    for d, c in zip(batch_data, coords):
        if c.size and (c.min() < 0 or c[:, 0].max() >= 8 or c[:, 1:].max() >= V):
            raise IndexError(f"target_value_map_pixel out of range for an [8,{V},{V}] value map (pose_i={d.get('pose_i')})")
    K = int(sum(len(c) for c in coords))
    stager.begin()
    xs = stager.buffer("xs", (n,) + tuple(x0.shape[1:]), torch.float32).numpy()
    gt = stager.buffer("gt", (n,) + tuple(batch_data[0]["current_gt_2d_layout"].shape[1:]), torch.float32).numpy()
    cd = stager.buffer("coords", (K, 3), torch.int64).numpy()
    gn = stager.buffer("gains", (K,), torch.float32).numpy()
    bi = stager.buffer("bidx", (K,), torch.int64).numpy()
    k = 0
    for i, (d, c) in enumerate(zip(batch_data, coords)):
        xs[i] = d["current_model_input"][0]
        gt[i] = d["current_gt_2d_layout"][0]
        m = len(c)
        cd[k:k + m] = c
        gn[k:k + m] = d["actual_coverage_gain"]
        bi[k:k + m] = i
        k += m
    dev, ev = stager.commit()
    return (dev["xs"], dev["gt"], dev["coords"], dev["gains"], dev["bidx"]), ev


def _collate_any(batch_data, device, stager):
    """Staged collation for records of the replay store's shape (one map per record); anything else through _collate."""
    if all(d["current_model_input"].shape[0] == 1 and d["current_gt_2d_layout"].shape[0] == 1 for d in batch_data):
        return _collate_staged(batch_data, device, stager)
    ev = torch.cuda.Event()
    out = _collate(batch_data, device)
    ev.record(torch.cuda.current_stream(device))
    return out, ev


BUCKET_BYTES = 64 << 20      # xGMI rings are per-link bound (~153 GB/s): 4 buckets cover the 200 MB of gradients


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None      # also a 1-rank group (RCCL on one GPU)


def allreduce_gradients(nbp):
    """Averages the accumulated gradients over the ranks: parameters are packed into flat buckets of at most
    BUCKET_BYTES in registration order, one all-reduce (RCCL on ROCm) per bucket."""
    dist = _dist()
    if dist is None:
        return
    world = dist.get_world_size()
    params = [p for p in nbp.parameters() if p.grad is not None]
    i = 0
    while i < len(params):
        bucket, size = [], 0
        while i < len(params) and (not bucket or size + params[i].grad.numel() * 4 <= BUCKET_BYTES):
            bucket.append(params[i]); size += params[i].grad.numel() * 4; i += 1
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        if dist.get_backend() == "gloo" and flat.is_cuda:      # CPU rendezvous in tests; RCCL reduces in place on the GPU
            host = flat.cpu()
            dist.all_reduce(host)
            flat.copy_(host)
        else:
            dist.all_reduce(flat)
        flat /= world
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad)); off += n


def _common_count(n, device):
    """Every rank must take the same number of optimizer steps: min over ranks of the local batch count."""
    dist = _dist()
    if dist is None:
        return n
    t = torch.tensor([n], dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def _mean_over_ranks(x: float, device) -> float:
    """Scalars that steer the optimiser (validation loss -> ReduceLROnPlateau) must agree on every rank."""
    dist = _dist()
    if dist is None:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


def train_experience_data(training_set_db, params, optimizer, nbp, device, current_epoch):
    """ref nbp_utils.py:340-395 (GradScaler without autocast is the identity scale for fp32; omitted).  As in the
    reference the early poses (pose_i <= 10) are dropped INSIDE each batch during epoch 1 (:348-362), a batch left empty
    is skipped before the optimiser-step test (:364-365), and the step fires every 8 non-empty batches or on the batch
    that reaches the end of the set (:385).  Under torchrun the ranks agree on "empty" and on the batch count, because
    the step contains the gradient all-reduce."""
    random.shuffle(training_set_db)
    # the batch losses of an accumulation window stay on the device until its optimizer step (the reference's `batch_loss.item()`
    # per batch, nbp_utils.py:384, is a device synchronisation per batch: the GPU then idles through the next batch's collation and
    # host-to-device copy).  Same numbers: the same fp32 losses, converted and added as Python floats in the same order.
    training_loss, pending, updates = [], [], 0
    accumulation_steps = 8
    bs = params.nbp_batch_size
    n_batches = _common_count((len(training_set_db) + bs - 1) // bs, device)
    multi_rank = _dist() is not None
    stager = _BatchStager(device) if (torch.device(device).type == "cuda" and _STAGE_BATCHES) else None

    def batch_of(bi):
        batch = training_set_db[bi * bs:(bi + 1) * bs]
        if current_epoch == 1:
            batch = [d for d in batch if d["pose_i"] > 10]
        return batch

    staged = None            # (batch index, tensors, event) of the batch whose copies were started under the previous one's compute
    for bi in range(n_batches):
        batch = batch_of(bi)
        have = 1 if batch else 0
        if multi_rank:
            have = _common_count(have, device)
        if not have:
            continue
        if stager is None:
            xs, gt, coords, gains, bidx = _collate(batch, device)
        else:
            if staged is None or staged[0] != bi:
                staged = (bi,) + _collate_any(batch, device, stager)
            (xs, gt, coords, gains, bidx), ev = staged[1], staged[2]
            torch.cuda.current_stream(device).wait_event(ev)
            for t in (xs, gt, coords, gains, bidx):
                t.record_stream(torch.cuda.current_stream(device))
        out1, out2 = nbp(xs)
        pred = tr.gather_values(out1, bidx, coords)
        loss = nbp.loss(pred, gains, out2, gt)
        loss.backward()
        pending.append(loss.detach())
        updates += 1
        if stager is not None and bi + 1 < n_batches:       # the next batch's collation and copies, under this batch's kernels
            nb = batch_of(bi + 1)
            staged = ((bi + 1,) + _collate_any(nb, device, stager)) if nb else None
        if updates % accumulation_steps == 0 or bi + 1 == n_batches:
            allreduce_gradients(nbp)
            optimizer.step()
            optimizer.zero_grad()
            accumulated = 0.0
            for v in torch.stack(pending).tolist():       # (one device -> host copy per optimizer step)
                accumulated += v
            training_loss.append(accumulated / accumulation_steps)
            pending, updates = [], 0
    return training_loss


def sync_buffers(nbp):
    """BatchNorm running statistics are per rank during training; before validation / checkpointing they are averaged so
    that every rank evaluates (and rank 0 saves) the same eval-mode network.  No-op without torch.distributed."""
    dist = _dist()
    if dist is None:
        return
    world = dist.get_world_size()
    for name, buf in nbp.named_buffers():
        if buf.dtype.is_floating_point:
            t = buf.data if dist.get_backend() == "nccl" else buf.data.cpu()
            dist.all_reduce(t)
            buf.data.copy_(t / world)
        # num_batches_tracked: identical on every rank (same number of forward passes)


def validation_model(training_set_db, params, nbp, device):
    """ref nbp_utils.py:293-338: plain MSE + BCE in eval mode."""
    parts, count = [], 0
    bs = params.nbp_batch_size
    stager = _BatchStager(device) if (torch.device(device).type == "cuda" and _STAGE_BATCHES) else None
    for i in range(0, len(training_set_db), bs):
        if stager is None:
            xs, gt, coords, gains, bidx = _collate(training_set_db[i:i + bs], device)
        else:
            (xs, gt, coords, gains, bidx), ev = _collate_any(training_set_db[i:i + bs], device, stager)
            torch.cuda.current_stream(device).wait_event(ev)
            for t in (xs, gt, coords, gains, bidx):
                t.record_stream(torch.cuda.current_stream(device))
        out1, out2 = nbp(xs)
        pred = tr.gather_values(out1, bidx, coords)
        parts.append((tr.MeanLossFn.apply(pred, gains, 0) + tr.MeanLossFn.apply(out2, gt, 1)).detach())    # (no sync per batch)
        count += 1
    total = 0.0
    for v in (torch.stack(parts).tolist() if parts else []):
        total += v
    return total / max(count, 1)


def train_nbp(training_set_db, params, optimizer, nbp, device, current_epoch, validation_data, lr_patience=2,
              lr_factor=0.1, num_epochs=5):
    """ref nbp_utils.py:430-468: 5 inner epochs, validation after each, ReduceLROnPlateau."""
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=lr_factor, patience=lr_patience)
    tl, vl = [], []
    for _ in range(num_epochs):
        nbp.train()
        losses = train_experience_data(training_set_db, params, optimizer, nbp, device, current_epoch)
        tl.append(float(np.mean(losses)) if losses else float("nan"))
        sync_buffers(nbp)
        nbp.eval()
        with torch.no_grad():
            vl.append(_mean_over_ranks(validation_model(validation_data, params, nbp, device), device))
        sched.step(vl[-1])
    return sum(tl) / len(tl), sum(vl) / len(vl)


def run_training_nbp(params):
    """ref train_nbp_model.py:40-157.  With a dataset at params.data_path: epoch 0 collects trajectories and
    moves the validation records out of the store, every later epoch collects again and trains on
    read_combined_data; without one (offline benchmark mode) the records are synthesised."""
    from ..parallel_rollout import init_distributed
    from ..simulator import scene as sim_scene
    from ..utility import nbp_utils as nu
    rank, world, local_rank = init_distributed()
    device = torch.device("cuda", local_rank if world > 1 else getattr(params, "numGPU", 0))
    torch.cuda.set_device(device)
    random.seed(params.random_seed + rank); np.random.seed(params.random_seed + rank); torch.manual_seed(params.torch_seed)
    nbp = NBP().to(device)
    nbp, optimizer, best_loss, _ = initialize_nbp(params, nbp, params.torch_seed)
    S = getattr(params, "grid_size", 256)
    os.makedirs(params.output_dir, exist_ok=True)
    data_path = getattr(params, "data_path", None)
    collect = bool(getattr(params, "collect", True)) and data_path and os.path.isdir(data_path)
    history = {}

    def save(epoch, vl, tl):
        nonlocal best_loss
        if rank != 0:
            return
        history[epoch] = {"training_loss": tl, "validation_loss": vl}
        ck = {"epoch": epoch, "model_state_dict": nbp.state_dict(), "optimizer_state_dict": optimizer.state_dict()}
        if vl < best_loss:
            best_loss = vl
            torch.save(ck, os.path.join(params.output_dir, params.nbp_model_name + "_best_val.pth"))
        if epoch % 3 == 0:
            torch.save(ck, os.path.join(params.output_dir, f"{params.nbp_model_name}_epoch{epoch}.pth"))
        with open(os.path.join(params.output_dir, "loss.json"), "w") as fh:
            json.dump(history, fh)

    if not collect:
        validation = make_synthetic_experiences(getattr(params, "n_validation_synthetic", 16), S, seed=1)
        for epoch in range(1, params.epochs + 1):
            db = make_synthetic_experiences(params.samples_per_epoch, S, seed=100 + epoch + 1000 * rank)
            tl, vl = train_nbp(db, params, optimizer, nbp, device, epoch, validation, num_epochs=params.inner_epochs)
            print(f"epoch {epoch}: training {tl:.4f} validation {vl:.4f}")
            save(epoch, vl, tl)
        return history

    dataset = sim_scene.SceneDataset(data_path, getattr(params, "train_scenes", []))
    db_dir = getattr(params, "db_path", os.path.join(params.output_dir, "db"))
    env = nu.open_experience_db(os.path.join(db_dir, f"{params.nbp_model_name}.rank{rank}"))
    validation = None
    for epoch in range(0, params.epochs + 1):
        cov = []
        with torch.no_grad():
            n = nu.trajectory_collection(params, epoch, dataset, env, (S, S), (S // 4, S // 4), (-40 * S // 256, 40 * S // 256),
                                         nbp, cov, None, device, rank=rank, world=world,
                                         n_poses=getattr(params, "n_collect_poses", 100),
                                         n_gt_points=getattr(params, "n_gt_surface_points", 50000))
        print(f"[rank {rank}] epoch {epoch}: collected {n} records ({env.entries()} in the store)")
        if epoch == 0:
            validation = nu.store_validation_data(env, getattr(params, "n_validation", 1200))
            continue
        db = nu.read_combined_data(env, sample_m=None) if epoch == 1 else nu.read_combined_data(env)   # ref :436-440
        # every rank must take the same branch (the training loop below contains collectives)
        if _common_count(1 if (db and validation) else 0, device) == 0:
            continue
        tl, vl = train_nbp(db, params, optimizer, nbp, device, epoch, validation, num_epochs=params.inner_epochs)
        print(f"epoch {epoch}: training {tl:.4f} validation {vl:.4f}")
        save(epoch, vl, tl)
    env.close()
    return history
