#!/bin/bash
# round 6, fourth GPU call: K9 aggregate variants (0 slot-by-slot, 1 + cold spill loop, 2 four-key buckets), the dense aggregates' no-return first-row update
# against the read-first form (c3 / c3w / q2), the row-hash route after the tuple-check fix, the sharded door (joins + update), the drop-in script.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in 0 1 2; do RFX_PLH_VAR=$v timeout 300 python tools/k9_ablate.py 2>&1 | grep RFX_PLH | sed "s/^/VAR=$v /"; done
for v in 0 2; do
  RFX_PLH_VAR=$v timeout 600 python bench.py --workload k9 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06d_k9_var$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k9 bench VAR=$v', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['verified'])"
done
} 2>&1 | tee gpurun_out/r06d_k9.txt
{
for w in c3 c3w q2; do for fr in 0 1; do
  RFX_PL_FIRST_READ=$fr timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict --engine-door 2> gpurun_out/r06d_${w}_fr$fr.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w RFX_PL_FIRST_READ=$fr', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done; done
} 2>&1 | tee gpurun_out/r06d_first_read.txt
echo "== tests: hashed / row-hash paths, default"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
echo "== tests: RFX_EMIT_BY_ROWS=2"; RFX_EMIT_BY_ROWS=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_ops_gpu.py -x -q -m gpu -k "rowhash or row_hash or tuple or multikey or key or join or q7 or wide" -p no:cacheprovider 2>&1 | tail -3
RFX_EMIT_BY_ROWS=2 timeout 600 python tools/fuzz_new_paths.py 0 120 2>&1 | tail -2
echo "== tests: RFX_PLH_VAR=2"; RFX_PLH_VAR=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider 2>&1 | tail -2
RFX_PLH_VAR=2 RFX_PLANE_HASH_PARTS=128 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider 2>&1 | tail -2
echo "== tests: sharded door (joins + update over shards), drop-in, fuzz slices"; timeout 1800 python -m pytest tests/test_sharded_gpu.py tests/test_dropin_gpu.py tests/test_fuzz_tools_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -25
