#!/bin/bash
# round 6: a long fuzz campaign over the last tree -- every fuzzer, fresh seed ranges, the shard / mode variants
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06u_fuzz.txt; : > $out
run() { echo "## $*" >> $out; ( timeout 1500 env "$@" 2>&1 | grep -a "^done\|^SEED\|Traceback\|Error" | tail -4 ) >> $out; }
run python tools/fuzz_ops.py 20000 21500
run RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_ops.py 21500 22500
run RFX_SHARDS=4 python tools/fuzz_ops.py 22500 23300
run RFX_DETERMINISTIC=1 python tools/fuzz_ops.py 23300 23600
run python tools/fuzz_select_extremes.py 20000 22000
run RFX_SHARDS=4 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_select_extremes.py 22000 23000
run python tools/fuzz_null_tuples.py 20000 21000
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_null_tuples.py 21000 22000
run RFX_SHARDS=3 python tools/fuzz_null_tuples.py 22000 22600
run python tools/fuzz_new_paths.py 20000 20800
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_new_paths.py 20800 21600
run python tools/fuzz_operators.py 20000 20600
run RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_operators.py 20600 21000
run python tools/fuzz_update_group.py 20000 20600
run RFX_SHARDS=3 python tools/fuzz_update_group.py 20600 21000
run python tools/fuzz_large.py 20000 20060
run python tools/fuzz_more.py 20000 20400
run python tools/fuzz_round3.py 20000 21000
cat $out
