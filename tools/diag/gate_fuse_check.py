"""Diagnostic (GPU box): the fused attention-gate middle (GateMidFn) against the separate Functions on the FULL network, same weights, same batch:
first-step loss, every parameter gradient (relative to the tensor's maximum) and the BatchNorm buffers after one forward; then the loss of eight AdamW
steps from both (Adam's early updates are +-lr whatever the gradient's size, so rounding-level differences grow from step to step: reported, not a bound)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import training as tr  # noqa: E402
from nextbestpath_amd.networks.nbp_model import NBP  # noqa: E402
from nextbestpath_amd.trainers.train_nbp_model import _collate, make_optimizer, make_synthetic_experiences  # noqa: E402

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 256
dev = torch.device("cuda")
db = make_synthetic_experiences(B, S, seed=3)
xs, gt, coords, gains, bidx = _collate(db, dev)


def run(fuse, steps, fanout=True, first_conv=True):
    tr._GATE_FUSE = fuse
    tr._FANOUT = fanout
    tr._FIRST_CONV = first_conv
    torch.manual_seed(9)
    net = NBP().to(dev).train()
    opt = make_optimizer(net)
    losses, grads, bufs = [], None, None
    for k in range(steps):
        o1, o2 = net(xs)
        loss = net.loss(tr.gather_values(o1, bidx, coords), gains, o2, gt)
        loss.backward()
        if k == 0:
            grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
            bufs = {n: b.detach().clone() for n, b in net.named_buffers() if b.dtype.is_floating_point}
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss.item()))
    return losses, grads, bufs


la, ga, ba = run(True, 8)
lb, gb, bb = run(False, 8)
lc, gc, bc = run(False, 8)                  # the separate Functions twice: what run-to-run identity looks like
ld, gd, bd = run(False, 1, fanout=False)    # yardstick 1: another rounding order in the BACKWARD only (autograd's pairwise gradient adds)
le, ge, be = run(False, 1, first_conv=False)      # yardstick 2: another legitimate rounding in the FORWARD (Conv1.conv.0 through the padded 64-channel kernels)
print("first-step loss  fused %.9g  separate %.9g  rel diff %.2e" % (la[0], lb[0], abs(la[0] - lb[0]) / abs(lb[0])))
# a bias in front of a BatchNorm has an exactly zero gradient: what is computed there is rounding noise, left out of the comparison
live = [n for n in gb if not (n.endswith(".bias") and (".conv.0." in n or ".conv.3." in n or ".up.1." in n or ".W_g.0." in n or ".W_x.0." in n
                                                       or ".psi.0." in n))]
rel = lambda x, y, n: float((x[n] - y[n]).abs().max() / y[n].abs().max().clamp_min(1e-30))
worst = sorted(((rel(ga, gb, n), n) for n in live), reverse=True)
yard = sorted(((rel(gd, gb, n), n) for n in live), reverse=True)
print("first-step gradients, max |a - b| / max |b| per tensor (%d tensors with a non-zero gradient):" % len(live))
print("   fused vs separate            worst", [(f"{v:.2e}", n) for v, n in worst[:3]], "median %.2e" % worst[len(worst) // 2][0])
print("   pairwise-add vs n-ary sums   worst", [(f"{v:.2e}", n) for v, n in yard[:3]], "median %.2e" % yard[len(yard) // 2][0], " (the yardstick:")
print("   another rounding order of the backward's sums: no ReLU mask moves)")
yard2 = sorted(((rel(ge, gb, n), n) for n in live), reverse=True)
print("   Conv1.conv.0 on the padded kernels vs the NCHW kernel   worst", [(f"{v:.2e}", n) for v, n in yard2[:3]], "median %.2e" % yard2[len(yard2) // 2][0],
      " first-step loss rel diff %.2e" % (abs(le[0] - lb[0]) / abs(lb[0])), "(another rounding of the FORWARD: ReLU masks at y = 0 move)")
wb = sorted(((float((ba[n] - bb[n]).abs().max() / bb[n].abs().max().clamp_min(1e-30)), n) for n in bb), reverse=True)
print("BatchNorm buffers after one forward: worst", [(f"{v:.2e}", n) for v, n in wb[:3]])
print("separate run twice identical:", lb == lc and all(torch.equal(gb[n], gc[n]) for n in gb))
print("loss per step  fused   ", ["%.5f" % v for v in la])
print("loss per step  separate", ["%.5f" % v for v in lb])
