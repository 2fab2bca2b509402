set -x
mkdir -p gpurun_out/r06
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_tile_probe tools/probes/mfma_tile_probe.hip && /tmp/mfma_tile_probe > gpurun_out/r06/mfma_tile_probe.txt 2>&1
cat gpurun_out/r06/mfma_tile_probe.txt
python tools/bench_forward.py --split --batch 24 > gpurun_out/r06/forward_layers_b24_base.txt 2>&1
python tools/bench_forward.py --split --batch 1 > gpurun_out/r06/forward_layers_b1_base.txt 2>&1
tail -3 gpurun_out/r06/forward_layers_b24_base.txt gpurun_out/r06/forward_layers_b1_base.txt
python bench.py > gpurun_out/r06/bench_base.json 2> gpurun_out/r06/bench_base.err
tail -c 3000 gpurun_out/r06/bench_base.json
