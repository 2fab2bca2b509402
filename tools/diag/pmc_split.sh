set -u
OUT=gpurun_out/pmc_split
REPO=$(pwd)
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
FW="python $REPO/tools/pmc_workload.py --precision fp32_split --batch 4 --size 256 --points 0"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$REPO/$OUT/pmc_$name" -o pmc -- $FW > "$REPO/$OUT/$name.log" 2>&1
done
cd $REPO
python tools/summarize_prof.py $OUT
grep "split_kernel" $OUT/pmc_summary.csv
