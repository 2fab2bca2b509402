# PMC passes over the training step (tools/bench_train.py): MFMA busy and LDS conflicts of the weight-gradient / convolution kernels
set -u
OUT=${1:-gpurun_out/train_pmc}; REPO=$(pwd); mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 500 rocprofv3 --pmc $pass --output-format csv -d "$REPO/$OUT/pmc_$name" -o pmc -- python $REPO/tools/bench_train.py --steps 2 --warmup 1 --cpu-batch 1 > "$REPO/$OUT/$name.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = []
for k, c in acc.items():
    g = c.get("GRBM_GUI_ACTIVE", [0, 1]); n = max(g[1], 1)
    rows.append((g[0], k, n, {cn: v[0] / max(v[1], 1) for cn, v in c.items()}))
rows.sort(reverse=True)
with open(out + "/summary.txt", "w") as fh:
    for tot, k, n, m in rows[:25]:
        gui = m.get("GRBM_GUI_ACTIVE", 0)
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024) if gui else 0
        lds = m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)
        wait = m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_ACTIVE_INST_ANY", 1), 1)
        mb = (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) / 1024
        line = f"{k:62s} n={n:4d} gui/launch {gui:12.0f}  mfma_busy {busy:5.3f}  lds_conflict/active {lds:5.3f}  wait/active {wait:6.2f}  MB/launch {mb:8.1f}"
        print(line); fh.write(line + "\n")
PY
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
