#!/usr/bin/env python
"""Replayed hipGraph of the eval forward (packing.ForwardGraph) against the eager call when the input tensor's CONTENT changes
between replays: max |difference| of (out1 / out2) for the captured content (A), new content (B: eager twice, replay twice, a
graph captured on B) and the first content again.  Everything must be 0: the replay is the same launches on the same buffers.
It was not while AmaxBook zeroed its slots with hipMemsetAsync -- ROCm 7.2 replays a captured memset node out of order with the
kernel nodes (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hides it); the slots are zeroed by a kernel now (csrc/nbp_forward.hip).
    python tools/diag/graph_replay_check.py [fp32_split|fp32|bf16]"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import packing
from nextbestpath_amd.utility.synthetic import make_count_maps, make_explorer_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32_split"
packed = packing.pack_state_dict(make_explorer_state_dict(9), torch.device("cuda"), precision=prec)
def d(a, b):
    return f"{(a[0].float()-b[0].float()).abs().max().item():.3e}/{(a[1].float()-b[1].float()).abs().max().item():.3e}"
for B, S in ((1, 256), (2, 128), (24, 256)):
    x = make_count_maps(B, S, seed=B).cuda()
    with torch.no_grad():
        ea = [t.clone() for t in packing.forward_packed(packed, x)]
        g = packing.ForwardGraph(packed, x)
        ga = [t.clone() for t in g()]
        x.copy_(make_count_maps(B, S, seed=B + 10).cuda())
        eb = [t.clone() for t in packing.forward_packed(packed, x)]
        eb2 = [t.clone() for t in packing.forward_packed(packed, x)]
        gb = [t.clone() for t in g()]
        gb2 = [t.clone() for t in g()]
        g2 = packing.ForwardGraph(packed, x)
        gc = [t.clone() for t in g2()]
        x.copy_(make_count_maps(B, S, seed=B).cuda())
        ec = [t.clone() for t in packing.forward_packed(packed, x)]
        gd = [t.clone() for t in g()]
        print(prec, B, S, "A: e-g", d(ea, ga), "| B: e-e", d(eb, eb2), "g-g", d(gb, gb2), "e-g", d(eb, gb), "e-newgraph", d(eb, gc), "g-newgraph", d(gb, gc),
              "| back to A: e-eA", d(ec, ea), "g-gA", d(gd, ga), "range", f"{eb[0].abs().max().item():.3e}", flush=True)
