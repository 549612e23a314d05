mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_golden.py tests/test_ops_gpu.py tests/test_dropin_gpu.py tests/test_dist_multi_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -25
