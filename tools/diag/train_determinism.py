"""Diagnostic (GPU box): are two identical training runs bit-identical?  Prints the per-batch losses of both."""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.trainers import train_nbp_model as T
D = torch.device("cuda")
def run():
    import random
    torch.manual_seed(3); random.seed(3); np.random.seed(3)
    params = types.SimpleNamespace(nbp_batch_size=4)
    db = T.make_synthetic_experiences(16, S=64, seed=5)
    net = NBP().to(D)
    _, opt, _, _ = T.initialize_nbp(params, net)
    net.train()
    out = []
    for ep in range(3):
        out += T.train_experience_data(list(db), params, opt, net, D, current_epoch=2)
    g = torch.cat([p.detach().flatten() for p in net.parameters()]).double().sum().item()
    return out, g
a, ga = run(); b, gb = run()
print("losses run 1:", [f"{v:.9g}" for v in a]); print("losses run 2:", [f"{v:.9g}" for v in b])
print("identical losses:", a == b, " parameter checksums:", ga, gb, ga == gb)
