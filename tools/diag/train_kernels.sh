# Diagnostic (GPU box): per-kernel share of the training step (rocprofv3 --kernel-trace --stats over tools/bench_train.py)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/tr2 -o trace -- python /root/repo/tools/bench_train.py --steps 3 --warmup 1 --cpu-batch 0 > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tr2/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-84s %5s %9.1f us %6.2f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
print("GPU time per step (4 steps): %.2f ms" % (tot / 1e6 / 4))
PY
rm -rf gpurun_out/tr2
