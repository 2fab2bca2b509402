"""NBP training -- host-side mirror of next_best_path/trainers/train_nbp_model.py:40-157 and the
training half of next_best_path/utility/nbp_utils.py (initialize_nbp :213-231, validation_model
:293-338, train_experience_data :340-395, train_nbp :430-468): same function names, same record
schema ('current_model_input', 'current_gt_2d_layout', 'target_value_map_pixel',
'actual_coverage_gain', 'pose_i'), same micro-batch / 8-step gradient accumulation / AdamW /
ReduceLROnPlateau logic.  Forward and backward run on the HIP kernels (networks/training.py).

Not built yet: the LMDB / msgpack replay store and the DAgger-style
trajectory collection that fills it; experiences are kept in an in-memory list and, offline, are
synthesised by ``make_synthetic_experiences`` (lmdb is not installable here)."""
from __future__ import annotations

import json
import os
import random

import numpy as np
import torch

from ..networks import training as tr
from ..networks.nbp_model import NBP


def initialize_nbp(params, nbp, torch_seed=9, initialize=False, pretrained=False, ddp_rank=None):
    """ref nbp_utils.py:213-231: AdamW(lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight decay 0.01)."""
    optimizer = torch.optim.AdamW(nbp.parameters(), lr=0.001, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    return nbp, optimizer, 10000.0, 0


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


def _collate(batch_data, device):
    xs = torch.cat([torch.from_numpy(np.copy(d["current_model_input"])) for d in batch_data]).to(device)
    gt = torch.cat([torch.from_numpy(np.copy(d["current_gt_2d_layout"])) for d in batch_data]).to(device)
    coords = [torch.from_numpy(np.copy(d["target_value_map_pixel"])) for d in batch_data]
    gains = torch.cat([torch.from_numpy(np.copy(d["actual_coverage_gain"])) for d in batch_data]).to(device)
    sizes = torch.tensor([len(c) for c in coords])
    bidx = torch.repeat_interleave(torch.arange(len(coords)), sizes).to(device)
    return xs, gt, torch.cat(coords).to(device), gains, bidx


def train_experience_data(training_set_db, params, optimizer, nbp, device, current_epoch):
    """ref nbp_utils.py:340-395 (GradScaler without autocast is the identity scale for fp32; omitted)."""
    random.shuffle(training_set_db)
    training_loss, accumulated, updates = [], 0.0, 0
    accumulation_steps = 8
    bs = params.nbp_batch_size
    for i in range(0, len(training_set_db), bs):
        batch = [d for d in training_set_db[i:i + bs] if (d["pose_i"] > 10 and current_epoch == 1) or current_epoch > 1]
        if not batch:
            continue
        xs, gt, coords, gains, bidx = _collate(batch, device)
        out1, out2 = nbp(xs)
        pred = tr.gather_values(out1, bidx, coords)
        loss = nbp.loss(pred, gains, out2, gt)
        loss.backward()
        accumulated += loss.item()
        updates += 1
        if updates % accumulation_steps == 0 or (i + bs) >= len(training_set_db):
            optimizer.step()
            optimizer.zero_grad()
            training_loss.append(accumulated / accumulation_steps)
            accumulated, updates = 0.0, 0
    return training_loss


def validation_model(training_set_db, params, nbp, device):
    """ref nbp_utils.py:293-338: plain MSE + BCE in eval mode."""
    total, count = 0.0, 0
    bs = params.nbp_batch_size
    for i in range(0, len(training_set_db), bs):
        xs, gt, coords, gains, bidx = _collate(training_set_db[i:i + bs], device)
        out1, out2 = nbp(xs)
        pred = tr.gather_values(out1, bidx, coords)
        total += (tr.MeanLossFn.apply(pred, gains, 0) + tr.MeanLossFn.apply(out2, gt, 1)).item()
        count += 1
    return total / max(count, 1)


def train_nbp(training_set_db, params, optimizer, nbp, device, current_epoch, validation_data, lr_patience=2,
              lr_factor=0.1, num_epochs=5):
    """ref nbp_utils.py:430-468: 5 inner epochs, validation after each, ReduceLROnPlateau."""
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=lr_factor, patience=lr_patience)
    tl, vl = [], []
    for _ in range(num_epochs):
        nbp.train()
        tl.append(float(np.mean(train_experience_data(training_set_db, params, optimizer, nbp, device, current_epoch))))
        nbp.eval()
        with torch.no_grad():
            vl.append(validation_model(validation_data, params, nbp, device))
        sched.step(vl[-1])
    return sum(tl) / len(tl), sum(vl) / len(vl)


def run_training_nbp(params):
    """ref train_nbp_model.py:40-157, single GPU (the reference's ddp / jz branches are `pass`)."""
    device = torch.device("cuda", getattr(params, "numGPU", 0))
    torch.cuda.set_device(device)
    random.seed(params.random_seed); np.random.seed(params.random_seed); torch.manual_seed(params.torch_seed)
    nbp = NBP().to(device)
    nbp, optimizer, best_loss, _ = initialize_nbp(params, nbp, params.torch_seed)
    S = getattr(params, "grid_size", 256)
    validation = make_synthetic_experiences(getattr(params, "n_validation", 16), S, seed=1)
    history = {}
    os.makedirs(params.output_dir, exist_ok=True)
    for epoch in range(1, params.epochs + 1):
        db = make_synthetic_experiences(params.samples_per_epoch, S, seed=100 + epoch)
        tl, vl = train_nbp(db, params, optimizer, nbp, device, epoch, validation, num_epochs=params.inner_epochs)
        history[epoch] = {"training_loss": tl, "validation_loss": vl}
        print(f"epoch {epoch}: training {tl:.4f} validation {vl:.4f}")
        ck = {"epoch": epoch, "model_state_dict": nbp.state_dict(), "optimizer_state_dict": optimizer.state_dict()}
        if vl < best_loss:
            best_loss = vl
            torch.save(ck, os.path.join(params.output_dir, params.nbp_model_name + "_best_val.pth"))
        if epoch % 3 == 0:
            torch.save(ck, os.path.join(params.output_dir, f"{params.nbp_model_name}_epoch{epoch}.pth"))
        with open(os.path.join(params.output_dir, "loss.json"), "w") as fh:
            json.dump(history, fh)
    return history
