"""Analyse a rocprofv3 kernel trace (csv): busy union, idle gaps, time per kernel family inside the last T ms."""
import csv, sys, collections
path = sys.argv[1]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
t_end = rows[-1][1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 300e6
t0 = t_end - win
sel = [r for r in rows if r[0] >= t0]
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = sel[-1][1] - sel[0][0]
print(f"window {span/1e6:.2f} ms, GPU busy (union) {busy/1e6:.2f} ms = {busy/span:.3f}; kernels {len(sel)}")
fam = collections.Counter(); cnt = collections.Counter()
for s, e, n, *_ in sel:
    base = n.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]
    key = "conv/igemm" if ("conv" in n or "igemm" in n or "splitk" in n or "psi_gate" in n or "maxpool" in n or "final_1x1" in n or "gate1x1" in n or "amax" in n) else base[-44:]
    fam[key] += e - s; cnt[key] += 1
for k, v in fam.most_common(25):
    print(f"  {k:45s} {v/1e6:9.3f} ms  {cnt[k]:6d} calls  avg {v/cnt[k]/1e3:7.1f} us")
# overlap: time where a conv kernel runs concurrently with a non-conv kernel

# GPU time during which NO conv kernel runs (the forward-bound rate loses exactly this)
conv = sorted((s_, e_) for s_, e_, n, *_ in sel if ("conv" in n or "igemm" in n or "splitk" in n or "gate1x1" in n))
cb, cs, ce = 0, None, None
for s_, e_ in conv:
    if ce is None or s_ > ce:
        if ce is not None: cb += ce - cs
        cs, ce = s_, e_
    else:
        ce = max(ce, e_)
cb += ce - cs
print(f"conv kernels busy (union) {cb/1e6:.2f} ms = {cb/span:.3f} of the window; non-conv-only + idle {(span-cb)/1e6:.2f} ms")
