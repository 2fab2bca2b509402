mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q -k "1x1 or small_functions or golden or block" 2>&1 | tail -5
export NBP_TUNING=1
for v in 0 1 0 1; do
  NBP_GATE_DMA=$v python tools/bench_forward.py --split --batch 24 --reps 10 > gpurun_out/r06/fwd_b24_gate_dma_$v.txt 2>&1
  grep -E "Att|fp32_split" gpurun_out/r06/fwd_b24_gate_dma_$v.txt | cut -c1-130
done
for v in 0 1; do NBP_GATE_DMA=$v python tools/bench_forward.py --split --batch 1 --reps 20 2>&1 | grep -E "Att._..W|fp32_split" | cut -c1-130; done
