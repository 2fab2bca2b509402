#!/usr/bin/env python
"""The roofline workload alone, for rocprofv3 PMC passes (bench.py runs it under `rocprofv3 --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE`; tools/profile.sh under more counters): 2 forwards of 8 maps of 256x256 through the fp32 network --
every launch of a conv kernel in this process belongs to that forward, so per-kernel means are per-launch figures on
the basis of bench.py's roofline.achieved -- and 2 map accumulations over a wall-like cloud of --points points
(y range of the mazes, so that all six channels are exercised)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextbestpath_amd import _lib  # noqa: E402
from nextbestpath_amd.networks import packing  # noqa: E402
from nextbestpath_amd.utility import utils as hu  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_count_maps, make_nbp_state_dict, make_point_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_500_000)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--precision", default=None, choices=packing.PRECISIONS)
    a = ap.parse_args()
    prec = a.precision or ("bf16" if a.bf16 else "fp32")
    L = _lib.lib()
    dev = torch.device("cuda")
    packed = packing.pack_state_dict(make_nbp_state_dict(9), dev, precision=prec)
    B, S = a.batch, a.size
    x = make_count_maps(B, S, seed=1).to(dev)
    o1 = torch.empty(B, 8, S // 4, S // 4, device=dev)
    o2 = torch.empty(B, 1, S, S, device=dev)
    nws = getattr(L, packing._FWD[prec][1])(B, S)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fwd = getattr(L, packing._FWD[prec][0])
    for _ in range(2):
        _lib.check(fwd(packed.handle, x.data_ptr(), B, S, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(),
                       _lib.current_stream()), "forward")
    torch.cuda.synchronize()       # host allocations / copies below must not overlap the forwards (they stall running kernels)
    if a.points > 0:
        pc = make_point_cloud(a.points, seed=1, extent=32.0, y_range=(0.0, 12.0)).to(dev)
        pose = np.array([1.0, 3.3, -2.0, 0, 0], np.float32)
        ybins = torch.arange(0.5, 11.5 + 2.75, 2.75)
        out = torch.empty(6, 256, 256, device=dev)
        for _ in range(2):
            hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), out=out)
        # the step loop's build: the tile-binned shadow copy (utils.CloudBins).  The first call files every point (a tail-only
        # launch, not what a step does); the two launches that follow are steady-state builds from the pages, and a last one meets
        # one step's worth of new points: bench.py reads the launches by kernel name and takes the LAST three
        n_new = min(29_180, a.points // 4)
        n_dev = torch.tensor([a.points - n_new], dtype=torch.int64, device=dev)
        bins = hu.CloudBins((-58.0, -58.0), (58.0, 58.0), pc.shape[0], dev)
        for k in range(4):
            if k == 3:
                n_dev.fill_(a.points)
            hu.accumulate_step_maps(pc, pose, ybins, 256, (-40, 40), n_dev=n_dev, out=out, bins=bins)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
