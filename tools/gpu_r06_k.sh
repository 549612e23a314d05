#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_repro_gpu.py tests/test_door_gpu.py tests/test_ops_gpu.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r06k_tests.txt 2>&1; tail -30 gpurun_out/r06k_tests.txt | cut -c1-400
