#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r06s_q7.txt
for v in 1 0; do
  if [ $v = 0 ]; then export RFX_NO_PACKED_TABLE=1; else unset RFX_NO_PACKED_TABLE; fi
  timeout 900 python bench.py --workload q7 --engine-door --steps 7 --warmup 2 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06s_q7_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q7 packed=$v', round(d['ms_per_step'],3), 'ms', d['config']['verified'][:120])" | tee -a gpurun_out/r06s_q7.txt
done
unset RFX_NO_PACKED_TABLE
RFX_EMIT_BY_ROWS=2 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_mapgroup_gpu.py tests/test_gpu_fuzz.py tests/test_door_gpu.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r06s_tests_byrows.txt 2>&1; tail -5 gpurun_out/r06s_tests_byrows.txt | cut -c1-400
out=gpurun_out/r06s_fuzz.txt; : > $out
run() { echo "## $*" >> $out; ( env "$@" 2>&1 | grep -a "^done" | tail -1 ) >> $out; }
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_null_tuples.py 7000 7300
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_new_paths.py 7000 7300
run RFX_EMIT_BY_ROWS=2 RFX_DETERMINISTIC=2 python tools/fuzz_null_tuples.py 7300 7500
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_select_extremes.py 7000 7400
cat $out
