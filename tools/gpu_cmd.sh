#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
ls rayforce_amd/rtc_cache | head
for w in c2b c5; do
  RFX_TRACE=1 timeout 600 python bench.py --workload $w --steps 10 2> gpurun_out/rtc_$w.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:30], d['ms_per_step'], d['roofline']['frac'], d['rtc'])"
  grep "rtc" gpurun_out/rtc_$w.err | head -8
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rtc or run_time or plan_kernels" 2>&1 | tail -3
ls rayforce_amd/rtc_cache | wc -l
