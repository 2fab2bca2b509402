"""Diagnostic (GPU box): is the training step bound by the host's enqueue rate?  Times the ENQUEUE of a step (no synchronise inside)
against its wall time with the GPU drained before and after."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import training as tr
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.trainers.train_nbp_model import _collate, make_optimizer, make_synthetic_experiences
dev = torch.device("cuda")
torch.manual_seed(9)
net = NBP().to(dev).train()
opt = make_optimizer(net)
xs, gt, coords, gains, bidx = _collate(make_synthetic_experiences(32, 256, seed=3), dev)
def step():
    o1, o2 = net(xs)
    loss = net.loss(tr.gather_values(o1, bidx, coords), gains, o2, gt)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
enq, wall = [], []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print("enqueue ms per step:", [round(v, 1) for v in enq], " wall ms:", [round(v, 1) for v in wall])
# forward / backward split of the enqueue time
torch.cuda.synchronize(); t0 = time.perf_counter(); o1, o2 = net(xs); loss = net.loss(tr.gather_values(o1, bidx, coords), gains, o2, gt); t1 = time.perf_counter()
loss.backward(); t2 = time.perf_counter(); opt.step(); opt.zero_grad(set_to_none=True); t3 = time.perf_counter(); torch.cuda.synchronize()
print(f"enqueue: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, optimizer {1e3*(t3-t2):.1f} ms")
