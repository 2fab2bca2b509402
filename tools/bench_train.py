#!/usr/bin/env python
"""Config 3 of BASELINE.json: NBP fwd + bwd + AdamW step, batch of 256x256 maps, fp32, 1 MI355X.
    python tools/bench_train.py [--batch 32] [--steps 5] [--size 256]
Prints one JSON line: train maps/s, TFLOP/s against 546.9 GFLOP/map, and the torch-CPU baseline
(stock autograd on the same weights = the reference's arithmetic) on a bounded sample."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextbestpath_amd.networks import training as tr  # noqa: E402
from nextbestpath_amd.networks.nbp_model import NBP  # noqa: E402
from nextbestpath_amd.trainers.train_nbp_model import _collate, make_optimizer, make_synthetic_experiences  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--cpu-batch", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(9)
    net = NBP().to(dev).train()
    opt = make_optimizer(net)
    db = make_synthetic_experiences(a.batch, a.size, seed=3)
    xs, gt, coords, gains, bidx = _collate(db, dev)

    def step():
        o1, o2 = net(xs)
        loss = net.loss(tr.gather_values(o1, bidx, coords), gains, o2, gt)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flop_map = 546.9e9 * (a.size / 256) ** 2
    # CPU baseline: torch autograd on the oracle network (bounded sample: cpu-batch maps, 1 step; --cpu-batch 0: none -- profiling runs,
    # whose copy counts would otherwise include the 327 device -> host copies of the state dict)
    cpu = None
    if a.cpu_batch > 0:
        from oracle import nbp_net
        sd = {k: (v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k))
              for k, v in net.state_dict().items()}
        nb = a.cpu_batch
        xc, gc = xs[:nb].cpu(), gt[:nb].cpu()
        sel = (bidx < nb).cpu()
        cc, gn, bi = coords.cpu()[sel], gains.cpu()[sel], bidx.cpu()[sel]
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        o1, o2 = nbp_net.nbp_forward(sd, xc, train=True)
        l = nbp_net.nbp_loss(sd["log_vars"], o1[bi, cc[:, 0], cc[:, 1], cc[:, 2]], gn, o2, gc)
        l.backward()
        cpu_dt = time.perf_counter() - t0
        cpu = {"value": round(nb / cpu_dt, 4), "unit": "maps/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"one fwd+bwd of {nb} maps with torch CPU autograd ({cpu_dt:.1f} s)"}
    print(json.dumps({
        "metric": "NBP training maps/s (fwd+bwd+AdamW, fp32)", "value": round(a.batch / dt, 3), "unit": "maps/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 2), "dtype": "f32",
        "data": "synthetic", "config": {"workload": f"configs[2]: train step, batch {a.batch} x {a.size}x{a.size}"},
        "tflops_reference_formulation": round(a.batch * flop_map / dt / 1e12, 2), "frac_of_split_ceiling_reference_formulation": round(a.batch * flop_map / dt / (2500e12 / 3), 4),
        "loss": float(loss.item()), "producer_notes": dict(tr.HANDOFF_STATS),
        "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
