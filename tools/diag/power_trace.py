#!/usr/bin/env python
"""Board power / core clock while the B = 24 forward runs back to back, eager and as a replayed hipGraph.

Question (round 4): a replayed graph of the B = 24 forward measured 8.1 ms in its first ~60 replays and 9.65 ms -- the eager
figure -- afterwards.  If the forward sits at the board's power cap, removing the launch gaps cannot buy time: the power
controller lowers the clock until the average power is the cap again.  This script samples hwmon (power1_average / power1_input,
freq1_input = sclk) every 20 ms from a thread while each mode runs for `--seconds`, and prints per-250-ms means.
    python tools/diag/power_trace.py [--batch 24] [--seconds 4]"""
import argparse
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextbestpath_amd.networks import packing  # noqa: E402
from nextbestpath_amd.utility.synthetic import make_count_maps, make_explorer_state_dict  # noqa: E402


def hwmon_files():
    """hwmon directory of the GPU torch runs on (the box has several cards; ours is matched by its PCI address)."""
    out = {}
    want = None
    try:
        p = torch.cuda.get_device_properties(0)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}."
    except Exception as e:
        print("no PCI id from torch:", e)
    dirs = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
    if want:
        mine = [d for d in dirs if want in os.path.realpath(os.path.dirname(os.path.dirname(d)))]
        print("PCI", want, "->", mine, flush=True)
        dirs = mine or dirs
    for d in dirs:
        for name in ("power1_average", "power1_input", "freq1_input", "temp1_input", "power1_cap"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out


def read(path):
    try:
        with open(path) as fh:
            return float(fh.read().strip())
    except Exception:
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, files, dt=0.02):
        super().__init__(daemon=True)
        self.files, self.dt, self.rows, self.stop = files, dt, [], False

    def run(self):
        pw = self.files.get("power1_average") or self.files.get("power1_input")
        fq = self.files.get("freq1_input")
        tp = self.files.get("temp1_input")
        while not self.stop:
            self.rows.append((time.perf_counter(), read(pw) / 1e6 if pw else float("nan"), read(fq) / 1e6 if fq else float("nan"),
                              read(tp) / 1e3 if tp else float("nan")))
            time.sleep(self.dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--precision", default="fp32_split")
    a = ap.parse_args()
    files = hwmon_files()
    print("hwmon:", files, "cap W:", read(files["power1_cap"]) / 1e6 if "power1_cap" in files else None, flush=True)
    dev = torch.device("cuda")
    packed = packing.pack_state_dict(make_explorer_state_dict(9), dev, precision=a.precision)
    S = 512 if a.precision == "bf16" else 256
    x = make_count_maps(a.batch, S, seed=3).to(dev)
    g = packing.ForwardGraph(packed, x)
    modes = [("idle", None), ("eager", lambda: packing.forward_packed(packed, x)), ("graph", g), ("eager", lambda: packing.forward_packed(packed, x)),
             ("graph", g)]
    for name, fn in modes:
        torch.cuda.synchronize()
        time.sleep(1.0)
        s = Sampler(files)
        s.start()
        t0 = time.perf_counter()
        n, marks = 0, []
        if fn is None:
            time.sleep(1.0)
        else:
            while time.perf_counter() - t0 < a.seconds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                e1.synchronize()
                marks.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 10))
                n += 10
        s.stop = True
        s.join()
        rows = [(t - t0, p, f, c) for t, p, f, c in s.rows]
        print(f"== {name}: {n} forwards", flush=True)
        w = 0.25
        k = 0
        while k * w < (a.seconds if fn else 1.0):
            sel = [r for r in rows if k * w <= r[0] < (k + 1) * w]
            ms = [m for t, m in marks if k * w <= t < (k + 1) * w]
            if sel:
                print(f"   t={k * w:5.2f}s  power {sum(r[1] for r in sel) / len(sel):7.1f} W  sclk {sum(r[2] for r in sel) / len(sel):7.1f} MHz  "
                      f"temp {sum(r[3] for r in sel) / len(sel):5.1f} C  ms/forward {sum(ms) / len(ms) if ms else float('nan'):.3f}", flush=True)
            k += 1


if __name__ == "__main__":
    main()
