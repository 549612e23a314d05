#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_repro_gpu.py tests/test_door_gpu.py tests/test_ops_gpu.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r06k_tests.txt 2>&1; tail -30 gpurun_out/r06k_tests.txt | cut -c1-400
RFX_DETERMINISTIC=1 timeout 900 python bench.py --workload c3w --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06k_det_c3w.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3w through rfx_select, RFX_DETERMINISTIC=1:', round(d['median_ms'],3), 'ms (median)')" | tee gpurun_out/r06k_det.txt
RFX_DETERMINISTIC=1 timeout 900 python bench.py --workload c3 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06k_det_c3.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 through rfx_select, RFX_DETERMINISTIC=1:', round(d['median_ms'],3), 'ms (median)')" | tee -a gpurun_out/r06k_det.txt
