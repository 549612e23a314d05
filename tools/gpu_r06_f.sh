#!/bin/bash
# round 6, the record run: the whole GPU suite, smoke, the default bench line, the two-aggregate sparse-key case, q1 with / without the next-tile prefetch, the upload
# overlap of four shards under rocprofv3 --memory-copy-trace, the N > 1 one-process record on one device.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r06f_suite.txt 2>&1; tail -8 gpurun_out/r06f_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err; tail -1 gpurun_out/r06f_bench.err | cut -c1-1800
for v in 2 0; do RFX_PLH_VAR=$v timeout 300 python tools/k9_two.py 2>&1 | grep "k9 sum"; done | tee gpurun_out/r06f_k9_two.txt
{ for pf in 1 0; do
  rm -rf /tmp/rtcq1; RFX_RTC_CACHE=/tmp/rtcq1 RFX_FEW_PREFETCH=$pf timeout 600 python bench.py --workload q1 --steps 5 --warmup 2 --no-also --no-cpu-baseline --no-predict 2> gpurun_out/r06f_q1_pf$pf.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q1 FEW_PREFETCH=$pf', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['verified'])"
done; } 2>&1 | tee gpurun_out/r06f_q1.txt
cd /tmp; rm -rf /tmp/rp_pin; RFX_SHARDS=4 timeout 600 rocprofv3 --memory-copy-trace --output-format csv -d /tmp/rp_pin -o pin -- python $OLDPWD/tools/pin_overlap.py run 200000000 > $OLDPWD/gpurun_out/r06f_pin_overlap.txt 2>&1; cd $OLDPWD
python tools/pin_overlap.py report /tmp/rp_pin >> gpurun_out/r06f_pin_overlap.txt 2>&1; RFX_SHARDS=1 timeout 300 python tools/pin_overlap.py run 200000000 2>&1 | grep rfx_pin >> gpurun_out/r06f_pin_overlap.txt; grep -v amdgpu.ids gpurun_out/r06f_pin_overlap.txt | tail -12
RFX_BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r06f_bench_2shards_one_device.json 2> gpurun_out/r06f_bench_2shards.err; python -c "
import json; d=json.load(open('gpurun_out/r06f_bench_2shards_one_device.json')); print('one process, 2 shards on one device:', d['ms_per_step'], 'ranks_seen', d['config']['ranks_seen'], d['config']['communicators'], d['config']['planner'], 'cpu_baseline' , (d['cpu_baseline'] or {}).get('value'))"
RFX_BENCH_SAME_DEVICE=1 RFX_EXEC_FORCE_RCCL=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2> /dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('... with a forced one-rank RCCL world:', d['ms_per_step'], 'ranks_seen', d['config']['ranks_seen'], d['config']['communicators'], d['config']['planner'])"
