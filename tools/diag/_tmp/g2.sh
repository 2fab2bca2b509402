set -x
mkdir -p gpurun_out/r06
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_tile_probe tools/probes/mfma_tile_probe.hip && /tmp/mfma_tile_probe > gpurun_out/r06/mfma_tile_probe.txt 2>&1
tail -14 gpurun_out/r06/mfma_tile_probe.txt
for v in 0 256 0 256; do
  NBP_SPLIT_DEEP_RING_BLOCKS=$v python tools/bench_forward.py --split --batch 1 --reps 30 > gpurun_out/r06/fwd_b1_ring_$v.txt 2>&1
  tail -1 gpurun_out/r06/fwd_b1_ring_$v.txt
done
for b in 2 4; do for v in 0 256; do
  NBP_SPLIT_DEEP_RING_BLOCKS=$v python tools/bench_forward.py --split --batch $b --reps 20 2>&1 | tail -1
done; done
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py -m gpu -x -q 2>&1 | tail -5
