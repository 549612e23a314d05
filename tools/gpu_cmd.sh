mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
