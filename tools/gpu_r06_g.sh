#!/bin/bash
# round 6: the rocprofv3 summaries (kernel stats + FETCH_SIZE / WRITE_SIZE passes) of every workload, and the upload overlap once more with a warmed-up layer
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile_round.sh r06 > gpurun_out/r06g_profile_round.log 2>&1; tail -3 gpurun_out/r06g_profile_round.log
cd /tmp; rm -rf /tmp/rp_pin; RFX_SHARDS=4 timeout 600 rocprofv3 --memory-copy-trace --output-format csv -d /tmp/rp_pin -o pin -- python $R/tools/pin_overlap.py run 400000000 2>&1 | grep -v "rocprofv3\|amdgpu.ids" > $R/gpurun_out/r06g_pin_overlap.txt; cd $R
python tools/pin_overlap.py report /tmp/rp_pin >> gpurun_out/r06g_pin_overlap.txt 2>&1
for s in 4 1; do RFX_SHARDS=$s timeout 300 python tools/pin_overlap.py run 400000000 2>&1 | grep rfx_pin | sed "s/^/(no profiler) /" >> gpurun_out/r06g_pin_overlap.txt; done
cat gpurun_out/r06g_pin_overlap.txt
