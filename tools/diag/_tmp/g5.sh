timeout 1200 python -m pytest tests/test_gpu_training_data.py tests/test_gpu_sim_planner.py tests/test_blocks_golden.py -m gpu -x -q 2>&1 | tail -8
