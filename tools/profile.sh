#!/bin/bash
# rocprofv3 passes for profiles/rNN: kernel stats of the default bench, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters cannot share a pass; see MI355X_MICROARCH.md).
# usage (on the GPU box): bash tools/profile.sh gpurun_out/prof_rNN
set -u
OUT=${1:-gpurun_out/prof}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -o trace -- $BENCH > "$REPO/$OUT/bench_under_rocprof.json" 2> "$REPO/$OUT/stats.err"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$REPO/$OUT/pmc_$name" -o pmc -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$REPO/$OUT/pmc_$name.json" 2> "$REPO/$OUT/pmc_$name.err"
done
# the roofline workload alone (8 maps of 256x256 through the fp32 forward; 8 of 512x512 through the bf16 one):
# every launch of a kernel in these runs belongs to the same forward, so the per-kernel means are per-launch
# figures on exactly the basis of bench.py's roofline.achieved
for kind in f32 bf16; do
  if [ $kind = f32 ]; then FW="python $REPO/tools/bench_forward.py --batch 8 --size 256 --reps 2 --quiet"; else FW="python $REPO/tools/bench_forward.py --bf16 --batch 8 --size 512 --reps 2 --quiet"; fi
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$REPO/$OUT/fwd_${kind}/pmc_$name" -o pmc -- $FW > "$REPO/$OUT/fwd_${kind}_$name.log" 2>&1
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/fwd_${kind}/stats" -o trace -- $FW > "$REPO/$OUT/fwd_${kind}_stats.log" 2>&1
done
cd "$REPO"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.log" 2>&1
for kind in f32 bf16; do python tools/summarize_prof.py "$OUT/fwd_${kind}" >> "$OUT/summary.log" 2>&1; done
