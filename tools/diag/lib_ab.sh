# steps/s of the default lock-step with TWO builds of libnbp_hip.so swapped on the same box, alternating (for changes that have no switch):
#   cp nextbestpath_amd/libnbp_hip.so tools/diag/_tmp/lib_before.so   (before the change; *.so travels with gpurun)
#   ... change, rebuild ...;  gpurun -- 'bash tools/diag/lib_ab.sh tools/diag/_tmp/lib_before.so'
set -u
OUT=gpurun_out/lib_ab; mkdir -p $OUT
OTHER=$1
cp nextbestpath_amd/libnbp_hip.so /tmp/lib_new.so
B="python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong"
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], 'steps/s', d['value'], 'ms/lockstep', d['ms_per_step'], 'fwd ms', d['stages']['nbp_forward']['ms'])" $1 "$2"; }
for rep in $(seq 1 ${REPS:-2}); do
  for v in new before; do
    if [ $v = new ]; then cp /tmp/lib_new.so nextbestpath_amd/libnbp_hip.so; else cp $OTHER nextbestpath_amd/libnbp_hip.so; fi
    timeout 400 $B > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err; val $OUT/${v}_$rep.json "$v"
  done
done
cp /tmp/lib_new.so nextbestpath_amd/libnbp_hip.so
