timeout 2400 python -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -15
