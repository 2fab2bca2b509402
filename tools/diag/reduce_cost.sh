cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/red -o t -- python /root/repo/tools/bench_forward.py --batch 4 --split --quiet --reps 5 2>/dev/null | tail -1
python - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/red/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][-60:], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1), r["Percentage"])
PY
