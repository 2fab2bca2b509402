# round 4, probe 1: (a) forward as a hipGraph, (b) parity kernels' XCD-sibling order: time + memory-side traffic
set -u
REPO=$(pwd); export TMPDIR=/tmp
OUT=$REPO/gpurun_out/r04_probe1; mkdir -p $OUT
python tools/diag/fwd_graph_ab.py --batches 1 2 4 24 > $OUT/fwd_graph_ab.txt 2>&1
for sib in 1 0; do
  NBP_TUNING=1 NBP_SPLIT_PH_SIB=$sib python tools/bench_forward.py --split --batch 24 --reps 10 > $OUT/fwd_b24_sib$sib.txt 2>&1
  NBP_TUNING=1 NBP_SPLIT_PH_SIB=$sib python tools/bench_forward.py --split --batch 1 --reps 20 > $OUT/fwd_b1_sib$sib.txt 2>&1
done
cd /tmp
for sib in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    NBP_TUNING=1 NBP_SPLIT_PH_SIB=$sib rocprofv3 --pmc $c --output-format csv -d $OUT/sib$sib/pmc_$c -o pmc -- python $REPO/tools/pmc_workload.py --precision fp32_split --batch 24 --size 256 --points 0 > /dev/null 2>&1
  done
  python $REPO/tools/summarize_prof.py $OUT/sib$sib > /dev/null 2>&1
  echo "sib $sib" >> $OUT/traffic.txt; grep "h2_kernel" $OUT/sib$sib/pmc_summary.csv | sed 's/(anonymous namespace):://g' >> $OUT/traffic.txt
done
cd $REPO
rm -rf $OUT/sib*/pmc_*
