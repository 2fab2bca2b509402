#!/bin/bash
# rocprofv3 passes for profiles/rNN: kernel stats of the default bench, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters cannot share a pass; see MI355X_MICROARCH.md).
# usage (on the GPU box): bash tools/profile.sh gpurun_out/prof_rNN
set -u
OUT=${1:-gpurun_out/prof}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
# the bench's own live PMC passes and extra stages are switched off here: rocprofv3 does not nest
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -o trace -- $BENCH > "$REPO/$OUT/bench_under_rocprof.json" 2> "$REPO/$OUT/stats.err"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d "$REPO/$OUT/pmc_$name" -o pmc -- python $REPO/bench.py --steps 6 --warmup 2 --advance 10 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong > "$REPO/$OUT/pmc_$name.json" 2> "$REPO/$OUT/pmc_$name.err"
done
# the roofline workload alone (tools/pmc_workload.py: 24 maps of 256x256 (one pipeline group of the default 48 rollouts per GPU) through the forward + the map accumulation;
# 8 of 512x512 through the bf16 forward; fwd_split = the default fp32_split path, fwd_f32 = the fp32 MFMA pipe): every launch of a kernel in these runs belongs to the same forward, so the
# per-kernel means are per-launch figures on exactly the basis of bench.py's roofline.achieved
for kind in split f32 bf16; do
  if [ $kind = split ]; then FW="python $REPO/tools/pmc_workload.py --precision fp32_split --batch 24 --size 256 --points 1900000";
  elif [ $kind = f32 ]; then FW="python $REPO/tools/pmc_workload.py --precision fp32 --batch 24 --size 256 --points 0";
  else FW="python $REPO/tools/pmc_workload.py --bf16 --batch 8 --size 512 --points 0"; fi
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $pass --output-format csv -d "$REPO/$OUT/fwd_${kind}/pmc_$name" -o pmc -- $FW > "$REPO/$OUT/fwd_${kind}_$name.log" 2>&1
  done
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/fwd_${kind}/stats" -o trace -- $FW > "$REPO/$OUT/fwd_${kind}_stats.log" 2>&1
done
# configs[2]: the training step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/train/stats" -o trace -- python $REPO/tools/bench_train.py --steps 3 --warmup 1 --cpu-batch 0 > "$REPO/$OUT/train.log" 2>&1
cd "$REPO"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.log" 2>&1
for kind in split f32 bf16; do python tools/summarize_prof.py "$OUT/fwd_${kind}" >> "$OUT/summary.log" 2>&1; done
python tools/summarize_prof.py "$OUT/train" >> "$OUT/summary.log" 2>&1
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# round 4: board power / clock under the B = 24 forward (eager and as a replayed graph), the graph A/B, the bf16 per-layer table,
# the binned map build against the append-order kernel
timeout 200 python tools/diag/power_trace.py --batch 24 --seconds 3 > "$OUT/power_trace_b24.txt" 2>&1
timeout 200 python tools/diag/power_trace.py --batch 8 --seconds 3 --precision bf16 > "$OUT/power_trace_bf16_512_b8.txt" 2>&1
timeout 300 python tools/diag/fwd_graph_ab.py --batches 1 2 4 24 > "$OUT/fwd_graph_ab.txt" 2>&1
timeout 300 python tools/diag/bf16_layer_table.py > "$OUT/bf16_layer_table.txt" 2>&1
timeout 200 python tools/diag/map_bins_ab.py > "$OUT/map_bins_ab.txt" 2>&1
# round 5: the training step per dispatch (which layer sizes a kernel's time sits in), its counters, its enqueue time, the BatchNorm passes
bash tools/diag/train_trace.sh "$OUT/train_trace.txt" > /dev/null 2>&1
bash tools/diag/train_pmc.sh "$OUT/train_pmc" > /dev/null 2>&1; cp "$OUT/train_pmc/summary.txt" "$OUT/train_pmc_summary.txt" 2>/dev/null
timeout 200 python tools/diag/train_host_time.py > "$OUT/train_host_time.txt" 2>&1
timeout 200 python tools/bench_bn.py > "$OUT/bench_bn.txt" 2>&1
timeout 300 python tools/bench_train.py --cpu-batch 1 > "$OUT/bench_train.json" 2> /dev/null
timeout 300 python tools/bench_forward.py --split --batch 24 --size 256 --reps 30 > "$OUT/forward_layers_b24.txt" 2>&1
timeout 300 python tools/bench_forward.py --split --batch 1 --size 256 --reps 200 > "$OUT/forward_layers_b1.txt" 2>&1
# gpurun merges at most 64 MiB back: the per-dispatch traces and counter rows are summarised above (kernel_stats.csv, pmc_summary.csv)
find "$OUT" \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
du -sh "$OUT" | tail -1
