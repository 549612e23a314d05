#!/bin/bash
# round 6, second GPU call: K9 A/B (RFX_PLH_VAR 0 = round 5's record-by-record probing, 1 = lockstep probing + no first-row read + cold spill loop), the
# ablations of both, the sparse-key tests under both variants, the mmap H2D rate with / without MADV_POPULATE_READ, the failing test of call one.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1; do
  for d in 0 3 2 1 4; do
    [ "$v" = 1 ] && [ "$d" != 0 ] && continue
    RFX_PLH_VAR=$v RFX_PLH_DBG=$d timeout 300 python tools/k9_ablate.py 2>&1 | grep RFX_PLH | sed "s/^/VAR=$v /"
  done
done > gpurun_out/r06b_k9_ab.txt 2>&1
cat gpurun_out/r06b_k9_ab.txt
for v in 0 1; do
  RFX_PLH_VAR=$v timeout 600 python bench.py --workload k9 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06b_k9_var$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k9 VAR=$v', d['ms_per_step'], d['roofline']['frac'], d['config']['verified'], d['config']['paths'])"
done 2>&1 | tee gpurun_out/r06b_k9_bench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider 2>&1 | tail -3
RFX_PLANE_HASH_PARTS=128 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "sampled_scope or two_threads or residency" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/h2d_bench.py 1000000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_h2d_populate.txt
RFX_IO_POPULATE=0 timeout 300 python tools/h2d_bench.py 1000000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_h2d_nopopulate.txt
