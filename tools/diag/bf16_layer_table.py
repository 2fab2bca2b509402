#!/usr/bin/env python
"""configs[4] forward (8 maps of 512 x 512, bf16): per layer the executed FLOPs, the algorithmic HBM bytes (each source once, the
packed weights once, the output once; bf16 activations, fp32 network input / outputs), the event-bracketed time, and what bounds
the layer: TFLOP/s against the dense bf16 peak (2500) and the 16-bit MFMA stream the board sustains at its power cap (1709), GB/s
against 8 TB/s and the ~6.3 TB/s a streaming kernel reaches.  VERDICT r03 item 6: every kernel below 0.4 of the MFMA peak should
either sit at >= 0.75 of achievable HBM or be fixed.
    python tools/diag/bf16_layer_table.py [--batch 8] [--size 512] [--reps 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd import _lib  # noqa: E402
from nextbestpath_amd.networks import packing  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_count_maps, make_nbp_state_dict  # noqa: E402

TILES = {1: "igemm 128x128", 2: "igemm 256x64", 3: "igemm 256x32", 4: "igemm 128x64", 5: "igemm 64x128", 6: "halo <4,1> 128ch",
         7: "halo <2,2> 64ch", 12: "halo-up <4,1>", 13: "halo-up <2,2>", -1: "element-wise"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda")
    packed = packing.pack_state_dict(make_nbp_state_dict(9), dev, precision="bf16")
    B, S = a.batch, a.size
    x = make_count_maps(B, S, seed=1).to(dev)
    o1 = torch.empty(B, 8, S // 4, S // 4, device=dev)
    o2 = torch.empty(B, 1, S, S, device=dev)
    ws = torch.empty(L.nbp_forward_workspace_bytes_bf16(B, S), dtype=torch.uint8, device=dev)
    acc = {}
    for rep in range(a.reps + 2):
        arr = (_lib.LayerTiming * 128)()
        n = C.c_int(0)
        _lib.check(L.nbp_forward_timed_bf16(packed.handle, x.data_ptr(), B, S, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _lib.current_stream(), arr, 128, C.byref(n)), "timed")
        if rep < 2:
            continue
        for i, t in enumerate(arr[:n.value]):
            r = acc.setdefault(i, dict(name=t.name.decode(), flops=t.flops, ms=[], tile=t.tile, split=t.split_k, M=t.M, N=t.N, K=t.K))
            r["ms"].append(t.ms)
    tot_ms = tot_b = 0.0
    print(f"{'layer':26s} {'kernel':18s} {'M':>9s} {'N':>5s} {'K':>5s} sk {'us':>8s} {'TF exec':>8s} {'of 2500':>7s} {'of 1709':>7s} "
          f"{'MB':>7s} {'GB/s':>7s} {'of 6300':>7s}  bound")
    for i in sorted(acc):
        r = acc[i]
        ms = sorted(r["ms"])[len(r["ms"]) // 2]
        name, M, N, K = r["name"], r["M"], r["N"], r["K"]
        g = 2 if "{1,2}" in name else 1
        up = ".up.1" in name
        taps = 1 if (".W_g" in name or "psi" in name or "Final" in name) else 9
        fl = r["flops"] * (4.0 / 9.0 if r["tile"] in (12, 13) else 1.0)
        if "first" in name:
            byt = M * 5 * 4 + M * N * 2 + 45 * 64 * 4
        elif "Maxpool" in name:
            byt = M * N * 2 * 5                                   # reads 4 M N, writes M N
        elif "psi" in name:
            byt = g * (M * K * 2 + 2 * M * 2 * K * 2)             # q, x in, gated out (C = 2 F)
        elif name.startswith("Final"):
            byt = M * K * 2 + M * N * 4
        else:
            cin = K // taps
            src = M * cin / (4 if up else 1)
            byt = g * (src * 2 + K * N * 2 + M * N * 2)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        gbs = byt / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        bound = "MFMA" if tf / 1709 >= gbs / 6300 else "HBM"
        tot_ms += ms
        tot_b += byt
        print(f"{name:26s} {TILES.get(r['tile'], str(r['tile'])):18s} {M:9d} {N:5d} {K:5d} {r['split']:2d} {ms * 1e3:8.1f} {tf:8.1f} "
              f"{tf / 2500:7.3f} {tf / 1709:7.3f} {byt / 1e6:7.1f} {gbs:7.0f} {gbs / 6300:7.3f}  {bound}")
    print(f"sum of layers {tot_ms:.3f} ms, {tot_b / 1e9:.2f} GB algorithmic -> {tot_b / (tot_ms * 1e-3) / 1e9:.0f} GB/s over the forward")


if __name__ == "__main__":
    main()
