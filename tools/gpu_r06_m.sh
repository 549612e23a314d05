#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_repro_gpu.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r06m_tests.txt 2>&1; tail -30 gpurun_out/r06m_tests.txt | cut -c1-600
