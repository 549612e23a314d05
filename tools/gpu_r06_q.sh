#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_mapgroup_gpu.py tests/test_gpu_fuzz.py tests/test_door_gpu.py tests/test_ops_gpu.py tests/test_sharded_gpu.py tests/test_fuzz_tools_gpu.py tests/test_c_host_gpu.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r06q_tests.txt 2>&1; tail -8 gpurun_out/r06q_tests.txt | cut -c1-600
for s in 0 1; do RFX_NO_INSERT_SLOTS=$([ $s = 0 ] && echo 1) ; export RFX_NO_INSERT_SLOTS; [ $s = 1 ] && unset RFX_NO_INSERT_SLOTS
timeout 900 python bench.py --workload q7 --engine-door --steps 7 --warmup 2 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06q_q7_$s.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q7 slots-from-insert=$s', round(d['ms_per_step'],3), 'ms', d['config']['verified'][:120])" | tee -a gpurun_out/r06q_q7.txt
done
for seeds in "0 60"; do timeout 900 python tools/fuzz_new_paths.py $seeds 2>&1 | tail -2; timeout 900 python tools/fuzz_null_tuples.py $seeds 2>&1 | tail -2; done | tee gpurun_out/r06q_fuzz.txt
