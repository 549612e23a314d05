#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
RFX_DETERMINISTIC=2 timeout 2400 python -m pytest tests/test_door_gpu.py tests/test_ops_gpu.py tests/test_dropin_gpu.py tests/test_sharded_gpu.py tests/test_fuzz_tools_gpu.py tests/test_baseline_configs_gpu.py tests/test_io_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r06n_det_env.txt 2>&1; tail -40 gpurun_out/r06n_det_env.txt | cut -c1-300
