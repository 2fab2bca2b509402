set -x
mkdir -p gpurun_out/r06
export NBP_EXTRA_FLAGS=-DNBP_DBG_TS
python -c "from nextbestpath_amd import build; build.build(verbose=False)"
for v in 0 256; do
for shape in "64 64 256 1" "128 128 128 1" "512 512 32 1"; do
  echo "=== ring $v shape $shape"
  NBP_SPLIT_DEEP_RING_BLOCKS=$v python tools/diag/conv_timeline.py $shape
done; done > gpurun_out/r06/conv_timeline_b1.txt 2>&1
cat gpurun_out/r06/conv_timeline_b1.txt
