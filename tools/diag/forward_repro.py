"""Diagnostic (GPU box): is the eval forward bit-reproducible?  Runs it N times on the same inputs (several batch sizes, a second
stream keeping the GPU busy in between) and counts outputs that differ from the first run."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nextbestpath_amd.networks import packing
from nextbestpath_amd.utility.synthetic import make_count_maps, make_explorer_state_dict
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32_split"
packed = packing.pack_state_dict(make_explorer_state_dict(9), dev, precision=prec)
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
for B in (1, 2, 3, 4, 8, 12):
    x = (make_count_maps(B, 256, seed=B) * 40).to(dev)
    o1, o2 = packing.forward_packed(packed, x)
    r1, r2 = o1.clone(), o2.clone()
    bad = 0; worst = 0.0
    for k in range(N):
        if k % 3 == 0:
            with torch.cuda.stream(side):
                junk2 = junk @ junk            # unrelated work on another stream
        o1, o2 = packing.forward_packed(packed, x)
        if not (torch.equal(o1, r1) and torch.equal(o2, r2)):
            bad += 1
            worst = max(worst, float((o1 - r1).abs().max()), float((o2 - r2).abs().max()))
    torch.cuda.synchronize()
    print(f"{prec} B={B}: {bad} of {N} runs differ from the first (max |diff| {worst:.3e}, out1 range {float(r1.abs().max()):.1f})", flush=True)
