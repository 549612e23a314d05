#!/bin/bash
# round 6, fifth GPU call (the fourth ran a torn object: the device half of rfx_group_plane.o compiled before an edit, the host half after): K9 variants,
# row-hash route, sharded door, drop-in -- every test leg's full output kept in gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for cfg in "0 0" "0 5" "1 0" "2 0"; do set -- $cfg; RFX_PLH_VAR=$1 RFX_PLH_DBG=$2 timeout 300 python tools/k9_ablate.py 2>&1 | grep RFX_PLH | sed "s/^/VAR=$1 /"; done
for v in 0 2; do
  RFX_PLH_VAR=$v timeout 600 python bench.py --workload k9 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06e_k9_var$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k9 bench VAR=$v', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['verified'])"
done
} 2>&1 | tee gpurun_out/r06e_k9.txt
echo "== tests: parity + golden, default"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r06e_t_default.txt 2>&1; tail -3 gpurun_out/r06e_t_default.txt
echo "== tests: RFX_EMIT_BY_ROWS=2"; RFX_EMIT_BY_ROWS=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_ops_gpu.py -x -q -m gpu -k "rowhash or row_hash or tuple or multikey or key or join or q7 or wide" -p no:cacheprovider > gpurun_out/r06e_t_byrows.txt 2>&1; tail -3 gpurun_out/r06e_t_byrows.txt
RFX_EMIT_BY_ROWS=2 timeout 600 python tools/fuzz_new_paths.py 0 120 2>&1 | tail -2
echo "== tests: RFX_PLH_VAR=2"; RFX_PLH_VAR=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider > gpurun_out/r06e_t_var2.txt 2>&1; tail -2 gpurun_out/r06e_t_var2.txt
RFX_PLH_VAR=2 RFX_PLANE_HASH_PARTS=128 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "sparse or hash or k9 or plane" -p no:cacheprovider > gpurun_out/r06e_t_var2_128.txt 2>&1; tail -2 gpurun_out/r06e_t_var2_128.txt
echo "== tests: sharded door, drop-in, fuzz slices, mapgroup, ops"; timeout 2400 python -m pytest tests/test_sharded_gpu.py tests/test_dropin_gpu.py tests/test_fuzz_tools_gpu.py tests/test_mapgroup_gpu.py tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r06e_t_door.txt 2>&1; tail -12 gpurun_out/r06e_t_door.txt
mkdir -p gpurun_out/rtc_dump; rm -rf /tmp/rtc_nocache; RFX_RTC_CACHE=/tmp/rtc_nocache RFX_RTC_DUMP=gpurun_out/rtc_dump timeout 600 python bench.py --workload q1 --steps 3 --warmup 2 --no-also --no-cpu-baseline --no-predict > /dev/null 2> gpurun_out/r06e_q1_dump.err; ls -la gpurun_out/rtc_dump | head
