#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_repro_gpu.py tests/test_dist_gpu.py tests/test_door_gpu.py tests/test_ops_gpu.py tests/test_sharded_gpu.py tests/test_fuzz_tools_gpu.py tests/test_dropin_gpu.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r06p_tests.txt 2>&1; tail -30 gpurun_out/r06p_tests.txt | cut -c1-600
timeout 900 python bench.py --workload c3w --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06p_c3w.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3w default', round(d['median_ms'],3), 'ms; reproducible modes:', json.dumps(d['door']['deterministic_mode']))" | tee gpurun_out/r06p_det.txt
timeout 900 python bench.py --workload c3 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06p_c3.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 default', round(d['median_ms'],3), 'ms; reproducible modes:', json.dumps(d['door']['deterministic_mode']))" | tee -a gpurun_out/r06p_det.txt
