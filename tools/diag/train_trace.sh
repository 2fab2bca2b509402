# Diagnostic (GPU box): per-dispatch kernel trace of ONE training step (tools/bench_train.py --steps 1 --warmup 2): for every kernel the
# launches of the last step grouped by grid size -- which layer sizes a kernel's time sits in, and the GB/s or TFLOP/s they imply.
# usage: bash tools/diag/train_trace.sh [out.txt]
OUT=${1:-gpurun_out/train_trace.txt}
REPO=$(pwd)
cd /tmp; export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/tr3
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tr3 -o trace -- python $REPO/tools/bench_train.py --steps 1 --warmup 2 --cpu-batch 0 > $REPO/gpurun_out/tr3.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob("gpurun_out/tr3/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = the last third of the launches by count (3 identical steps)
n = len(rows) // 3
last = rows[-n:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
acc = collections.OrderedDict()
for r in last:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:70]
    grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    a = acc.setdefault(name, collections.OrderedDict()).setdefault(grid, [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[1], "w") as fh:
    def p(s):
        print(s); fh.write(s + "\n")
    p(f"one training step: {n} launches, wall {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms")
    for name, grids in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in grids.values())
        if tot < 0.002 * busy:
            continue
        p(f"{name:72s} {sum(v[0] for v in grids.values()):4d} launches {tot / 1e3:9.1f} us {100.0 * tot / busy:5.2f} %")
        for g, (c, t) in sorted(grids.items(), key=lambda kv: -kv[1][1])[:8]:
            p(f"      grid {str(g):22s} x{c:3d}  {t / c / 1e3:9.1f} us each")
PY
rm -rf gpurun_out/tr3
