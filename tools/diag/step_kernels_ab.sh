# per-kernel GPU time of the lock-step (rocprofv3 --kernel-trace --stats over a short bench) for the default and for each variant
# usage: bash tools/diag/step_kernels_ab.sh "NBP_MAP_BINS=0" ...
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/step_kernels_ab; mkdir -p $OUT; export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong"
cd /tmp
run() { tag=$1; shift; rm -rf $OUT/$tag; timeout 600 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o trace -- $B > $OUT/$tag.json 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); echo "== $tag"; head -14 "$f" | cut -c1-150; cp "$f" $OUT/${tag}_kernel_stats.csv; rm -rf $OUT/$tag; }
run default X=1
i=0
for v in "$@"; do i=$((i+1)); run v$i NBP_TUNING=1 $v; done
