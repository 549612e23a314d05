#!/bin/bash
# round 6: extra fuzz seeds beyond the suite's slices, over the paths this round touched (residency by ownership everywhere; joins / update over shards; the
# row-hash tail by rows; four-key buckets in the sparse-key aggregate)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06h_fuzz_extra.txt; : > $O
timeout 900 python bench.py --workload q7 --steps 5 --warmup 2 --no-also --no-cpu-baseline --no-predict --engine-door 2> gpurun_out/r06h_q7.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q7 through the planner', round(d['ms_per_step'],3), d['config']['verified'])" | tee gpurun_out/r06h_q7.txt
RFX_EMIT_BY_ROWS=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "row_hash or key_tuples or several_keys or join" -p no:cacheprovider 2>&1 | tail -2
run() { echo "## $*" >> $O; ( timeout 420 env "$@" 2>&1 | grep -v amdgpu.ids | tail -4 ) >> $O; }
run python tools/fuzz_ops.py 0 600
run RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_ops.py 0 600
run RFX_SHARDS=4 python tools/fuzz_ops.py 600 1000
run python tools/fuzz_update_group.py 0 300
run RFX_SHARDS=3 python tools/fuzz_update_group.py 0 300
run RFX_SHARDS=4 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_update_group.py 300 500
run python tools/fuzz_null_tuples.py 0 300
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_null_tuples.py 300 600
run RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_null_tuples.py 0 300
run python tools/fuzz_new_paths.py 0 300
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_new_paths.py 300 500
run python tools/fuzz_operators.py 0 300
run RFX_SHARDS=3 python tools/fuzz_operators.py 300 500
run python tools/fuzz_select_extremes.py 0 800
run RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_select_extremes.py 800 1400
run python tools/fuzz_round3.py 0 600
run RFX_PLH_VAR=0 python tools/fuzz_round3.py 600 900
run FUZZ_SHARDS=1 python tools/fuzz_large.py 300 330
run RFX_VALIDATE=checksum python tools/fuzz_null_tuples.py 600 800
cat $O
