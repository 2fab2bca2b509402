#!/usr/bin/env python
"""The eval forward's launch-plan knobs measured AT THE POWER CAP: each setting runs in its own process (the knobs are read once),
warms for --warm seconds so that the board has settled at its sustained clock (the first ~0.5 s of a forward loop runs 15 % faster
than the rest: profiles/r04/power_trace_*.txt), then times --run seconds of back-to-back forwards.  Short bursts -- how the plan
constants were first chosen -- see the unthrottled clock and can prefer plans that cost more energy per forward.

    python tools/diag/knobs_at_cap.py [--batch 24] [--size 256] [--precision fp32_split] [--set NAME=V ...]
    (child)  python tools/diag/knobs_at_cap.py --child"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEFAULT_SETS = [
    "", "NBP_SPLIT_MIN_BLOCKS=128", "NBP_SPLIT_MIN_BLOCKS=512", "NBP_SPLIT_DEEP=0", "NBP_SPLIT_DEEP=8", "NBP_SPLIT_R8_BLOCKS=0",
    "NBP_SPLIT_R8_BLOCKS=512", "NBP_SPLIT_R8_TAIL=0", "NBP_SPLIT_R8_TAIL=256", "NBP_XCD_REMAP=0", "NBP_XCD_REMAP=1", "NBP_CONV_POOL=0",
    "NBP_CONV_HEAD=0", "NBP_GATE_PSI=0", "NBP_SPLIT_GATE=0", "NBP_SPLIT_UP=0", "NBP_SPLIT_MAX_K=4608", "NBP_SPLIT_MAX_K=9216",
    "NBP_SPLIT_MAX_K_SMALL=2304", "",
]


def child(a):
    import torch
    sys.path.insert(0, ROOT)
    from nextbestpath_amd.networks import packing
    from nextbestpath_amd.utility.synthetic import make_count_maps, make_explorer_state_dict
    dev = torch.device("cuda")
    packed = packing.pack_state_dict(make_explorer_state_dict(9), dev, precision=a.precision)
    x = make_count_maps(a.batch, a.size, seed=a.batch).to(dev)
    ws = None
    with torch.no_grad():
        packing.forward_packed(packed, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.warm:
            for _ in range(10):
                packing.forward_packed(packed, x)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        e0.record()
        while time.perf_counter() - t0 < a.run:
            for _ in range(10):
                packing.forward_packed(packed, x)
            n += 10
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
    print(f"RESULT {e0.elapsed_time(e1) / n:.4f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="fp32_split")
    ap.add_argument("--warm", type=float, default=2.5)
    ap.add_argument("--run", type=float, default=2.5)
    ap.add_argument("--set", nargs="*", default=None, help="NAME=V[,NAME=V] settings to try ('' = the defaults)")
    a = ap.parse_args()
    if a.child:
        return child(a)
    for s in (a.set if a.set is not None else DEFAULT_SETS):
        env = dict(os.environ, NBP_TUNING="1")
        for kv in filter(None, s.split(",")):
            k, v = kv.split("=")
            env[k] = v
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch), "--size", str(a.size), "--precision",
               a.precision, "--warm", str(a.warm), "--run", str(a.run)]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
            ms = [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
            print(f"{s or '(defaults)':34s} {ms[0] if ms else 'FAILED ' + out.stderr[-200:]} ms", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{s:34s} timeout", flush=True)


if __name__ == "__main__":
    main()
