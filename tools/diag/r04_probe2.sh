# round 4, probe 2: which runtime setting (if any) gives eager launches the graph's kernel-to-kernel behaviour
set -u
OUT=gpurun_out/r04_probe2; mkdir -p $OUT
run() { echo "== $1" >> $OUT/env_ab.txt; env $1 python tools/diag/fwd_graph_ab.py --batches 24 >> $OUT/env_ab.txt 2>&1; }
run "X=1"
run "AMD_DIRECT_DISPATCH=0"
run "ROC_SYSTEM_SCOPE_SIGNAL=0"
run "AMD_OPT_FLUSH=0"
run "AMD_OPT_FLUSH=1"
run "GPU_FLUSH_ON_EXECUTION=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "AMD_SERIALIZE_KERNEL=0"
cat $OUT/env_ab.txt
