#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_ops_gpu.py tests/test_dropin_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r06j_tests.txt 2>&1; tail -6 gpurun_out/r06j_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-also --no-predict > gpurun_out/r06j_bench.json 2> gpurun_out/r06j_bench.err; tail -1 gpurun_out/r06j_bench.err | cut -c1-900
