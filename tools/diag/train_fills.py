"""Diagnostic (GPU box): which Python lines launch the ATen fill / copy / add kernels of a training step (torch.profiler with stacks)."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import training as tr  # noqa: E402
from nextbestpath_amd.networks.nbp_model import NBP  # noqa: E402
from nextbestpath_amd.trainers.train_nbp_model import _collate, make_optimizer, make_synthetic_experiences  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(9)
net = NBP().to(dev).train()
opt = make_optimizer(net)
xs, gt, coords, gains, bidx = _collate(make_synthetic_experiences(8, 256, seed=3), dev)


def step():
    o1, o2 = net(xs)
    loss = net.loss(tr.gather_values(o1, bidx, coords), gains, o2, gt)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::contiguous")
cnt = Counter()
for ev in prof.events():
    if ev.name in want:
        st = [s for s in ev.stack if "nextbestpath_amd" in s or "torch/optim" in s or "autograd" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:5d}  {name:18s} {where[:150]}")
