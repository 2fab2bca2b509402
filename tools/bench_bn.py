#!/usr/bin/env python
"""Training-mode BatchNorm passes alone (GPU box): forward (statistics + apply) and backward (sums + apply) on the tensor shapes of
the B = 32, 256 x 256 train step; reports the HBM rate of each call (forward: 2 reads + 1 write of the tensor, backward: dy, x, y
read twice... counted as the bytes a perfect two-pass implementation moves: forward 3 T, backward 7 T with ReLU, T = tensor bytes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nextbestpath_amd import _lib  # noqa: E402

L = _lib.lib()
dev = "cuda"
st = _lib.current_stream


def ev(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for M, C in ((32 * 256 * 256, 64), (32 * 128 * 128, 128), (32 * 64 * 64, 256), (32 * 32 * 32, 512), (32 * 16 * 16, 1024)):
    x = torch.randn(M, C, device=dev)
    dy = torch.randn(M, C, device=dev)
    g = torch.rand(C, device=dev) + 0.5
    b = torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    dg, db, cs = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    slot = torch.zeros(64, dtype=torch.int32, device=dev)
    ws = torch.empty(L.nbp_colreduce_workspace_bytes(M, C), dtype=torch.uint8, device=dev)
    T = M * C * 4

    def fwd():
        rc = L.nbp_bn_train_forward_amax_f32(x.data_ptr(), M, C, g.data_ptr(), b.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), 1,
                                             mean.data_ptr(), invstd.data_ptr(), y.data_ptr(), slot.data_ptr(), ws.data_ptr(), ws.numel(), st())
        assert rc == 0, rc

    def bwd():
        rc = L.nbp_bn_train_backward_fused_f32(dy.data_ptr(), x.data_ptr(), y.data_ptr(), M, C, mean.data_ptr(), invstd.data_ptr(), g.data_ptr(), 1,
                                               dx.data_ptr(), dg.data_ptr(), db.data_ptr(), cs.data_ptr(), slot.data_ptr(), ws.data_ptr(),
                                               ws.numel(), st())
        assert rc == 0, rc

    tf, tb = ev(fwd), ev(bwd)
    print(f"M={M:8d} C={C:5d} ({T/1e6:6.1f} MB): forward {tf*1e3:7.1f} us = {3*T/tf/1e9:6.2f} TB/s (3 T)   backward {tb*1e3:7.1f} us = "
          f"{7*T/tb/1e9:6.2f} TB/s (7 T)")
