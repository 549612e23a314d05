#!/bin/bash
# round 6, third GPU call: K9 with the no-return first-row update as the default (+ VAR 1 = cold spill loop, DBG 5 = round 5's read-first form), q7 with the
# emit by rows / fused tuple check against the slot-ranking form, q1 with two row pairs per lane, column files by pread, the touched tests.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for cfg in "0 0" "1 0" "0 5"; do
  set -- $cfg
  RFX_PLH_VAR=$1 RFX_PLH_DBG=$2 timeout 300 python tools/k9_ablate.py 2>&1 | grep RFX_PLH | sed "s/^/VAR=$1 /"
done
for v in 0 1; do
  RFX_PLH_VAR=$v timeout 600 python bench.py --workload k9 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06c_k9_var$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k9 bench VAR=$v', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['verified'])"
done
} 2>&1 | tee gpurun_out/r06c_k9.txt
{
for m in 0 1; do
  RFX_EMIT_BY_ROWS=$m timeout 900 python bench.py --workload q7 --steps 5 --warmup 2 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06c_q7_rows$m.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q7 RFX_EMIT_BY_ROWS=$m', round(d['ms_per_step'],3), d['config']['verified'])"
done
} 2>&1 | tee gpurun_out/r06c_q7.txt
{
for u in 1 2; do
  RFX_FEW_U=$u timeout 600 python bench.py --workload q1 --steps 5 --warmup 2 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06c_q1_u$u.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q1 RFX_FEW_U=$u', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['verified'])"
done
} 2>&1 | tee gpurun_out/r06c_q1.txt
timeout 300 python tools/h2d_bench.py 1000000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c_h2d.txt
echo "== tests: hashed / row-hash paths, default"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_io_gpu.py tests/test_mapgroup_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
echo "== tests: RFX_EMIT_BY_ROWS=2 (emit by rows wherever the probe exists)"; RFX_EMIT_BY_ROWS=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_ops_gpu.py -x -q -m gpu -k "rowhash or row_hash or tuple or multikey or key or join or q7 or wide" -p no:cacheprovider 2>&1 | tail -3
RFX_EMIT_BY_ROWS=2 timeout 600 python tools/fuzz_new_paths.py 0 120 2>&1 | tail -2
RFX_EMIT_BY_ROWS=2 timeout 600 python tools/fuzz_null_tuples.py 3000 3120 2>&1 | tail -2
RFX_PLH_VAR=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider 2>&1 | tail -2
echo "== tests: sharded door (joins + update over shards), drop-in"; timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_dropin_gpu.py tests/test_fuzz_tools_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15
