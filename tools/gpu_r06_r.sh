#!/bin/bash
# round 6, the last commits (insert-recorded slots, reproducible sums with one / two limbs, rank slices): more seeds than the suite's slices, fresh ranges
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06r_fuzz.txt; : > $out
run() { echo "## $*" >> $out; ( env "$@" 2>&1 | grep -a "^done" | tail -1 ) >> $out; }
run python tools/fuzz_null_tuples.py 5000 5400
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_null_tuples.py 5400 5800
run RFX_NO_INSERT_SLOTS=1 RFX_EMIT_BY_ROWS=2 python tools/fuzz_null_tuples.py 5800 6000
run RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_null_tuples.py 6000 6300
run python tools/fuzz_new_paths.py 5000 5300
run RFX_EMIT_BY_ROWS=2 python tools/fuzz_new_paths.py 5300 5600
run RFX_DETERMINISTIC=2 python tools/fuzz_select_extremes.py 5000 5800
run RFX_DETERMINISTIC=2 RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_select_extremes.py 5800 6400
run RFX_DETERMINISTIC=2 RFX_SHARDS=4 python tools/fuzz_null_tuples.py 6300 6600
run RFX_DETERMINISTIC=2 python tools/fuzz_ops.py 5000 5400
run RFX_DETERMINISTIC=2 RFX_SHARDS=3 RFX_EXEC_SLICE_SHARDS=1 python tools/fuzz_ops.py 5400 5800
run RFX_DETERMINISTIC=2 RFX_VALIDATE=checksum python tools/fuzz_ops.py 5800 6000
cat $out
