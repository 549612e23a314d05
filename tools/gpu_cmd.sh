timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -12
