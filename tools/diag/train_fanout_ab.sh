# Diagnostic (GPU box): the training step with / without FanOutFn (n-ary gradient sums): step time + per-kernel shares
cd /tmp; export TMPDIR=/tmp
for f in 1 0; do
  echo "== NBP_TRAIN_FANOUT=$f"
  NBP_TUNING=1 NBP_TRAIN_FANOUT=$f python /root/repo/tools/bench_train.py --steps 5 --warmup 2 --cpu-batch 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'maps/s', d['ms_per_step'], 'ms', d.get('producer_notes'))"
  NBP_TUNING=1 NBP_TRAIN_FANOUT=$f rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/fan$f -o trace -- python /root/repo/tools/bench_train.py --steps 3 --warmup 1 --cpu-batch 0 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("/root/repo/gpurun_out/fan$f/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%-100s %5s %9.1f us %8.2f ms %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["TotalDurationNs"]) / tot * 100))
print("GPU time per step (4 steps): %.2f ms, launches %d" % (tot / 1e6 / 4, sum(int(r["Calls"]) for r in rows) / 4))
PY
  rm -rf /root/repo/gpurun_out/fan$f
done
