"""Diagnostic: from a rocprofv3 kernel trace of bench.py, the fraction of wall time in which at least one forward kernel is
running (union), the summed forward-kernel time, and the same for the other kernels, inside the last T ms."""
import csv, sys
FWD = ("conv3x3_halo", "splitk_reduce", "maxpool2", "gate1x1", "psi_gate", "final_1x1", "conv_first", "amax_kernel", "igemm")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 200e6
skip = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0
t1 = rows[-1][1] - skip
t0 = t1 - win
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)
f = [(s, e) for s, e, n in sel if any(k in n for k in FWD)]
o = [(s, e) for s, e, n in sel if not any(k in n for k in FWD)]
span = sel[-1][1] - sel[0][0]
print(f"window {span/1e6:.1f} ms: any kernel {union(f+o)/span:.3f}, a forward kernel {union(f)/span:.3f} (sum {sum(e-s for s,e in f)/span:.3f}), "
      f"another kernel {union(o)/span:.3f} (sum {sum(e-s for s,e in o)/span:.3f}); launches fwd {len(f)} other {len(o)}")
