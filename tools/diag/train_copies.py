#!/usr/bin/env python
"""Which Python lines of a training step issue device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer)?
    python tools/diag/train_copies.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import training as tr  # noqa: E402
from nextbestpath_amd.networks.nbp_model import NBP  # noqa: E402
from nextbestpath_amd.trainers.train_nbp_model import _collate, make_synthetic_experiences  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(9)
net = NBP().to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
db = make_synthetic_experiences(8, 256, seed=3)
xs, gt, coords, gains, bidx = _collate(db, dev)


def step():
    o1, o2 = net(xs)
    loss = net.loss(tr.gather_values(o1, bidx, coords), gains, o2, gt)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


step(); step()
torch.cuda.synchronize()
# count Tensor.copy_ / clone / contiguous calls that copy, by caller line, through a dispatch-mode hook
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("aten.copy_", "aten.clone", "aten._to_copy", "aten.contiguous", "aten.cat", "aten.add.Tensor", "aten.add_.Tensor")):
            t = args[0] if args and isinstance(args[0], torch.Tensor) else None
            if t is None or t.is_cuda:
                fr = [f for f in traceback.extract_stack() if "nextbestpath_amd" in f.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:70]}" if fr else "(autograd engine / optimizer)"
                n = t.numel() if t is not None else 0
                self.hits[(name, where, "big" if n > 65536 else "small")] += 1
        return func(*args, **(kwargs or {}))


with Spy() as spy:
    step()
torch.cuda.synchronize()
for (name, where, size), n in spy.hits.most_common(40):
    print(f"{n:5d}  {name:28s} {size:5s} {where}")
