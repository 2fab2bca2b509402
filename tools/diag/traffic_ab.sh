# memory-side traffic of the split forward (B = 8, 256 x 256) under the XCD mapping modes: 2 x FETCH_SIZE + WRITE_SIZE per launch
set -u
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for mode in auto 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    export NBP_TUNING=1      # the A/B switches are ignored without the opt-in (csrc/nbp_tuning.cpp)
    if [ $mode = auto ]; then unset NBP_XCD_REMAP; else export NBP_XCD_REMAP=$mode; fi
    rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/tab/$mode/pmc_$c -o pmc -- python $REPO/tools/pmc_workload.py --precision fp32_split --batch 8 --size 256 --points 0 > /dev/null 2>&1
  done
  python $REPO/tools/summarize_prof.py $REPO/gpurun_out/tab/$mode > /dev/null
  echo "mode $mode"; grep "h2_kernel" $REPO/gpurun_out/tab/$mode/pmc_summary.csv | sed 's/(anonymous namespace):://g' | cut -c1-50,80-
done
