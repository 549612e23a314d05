#!/bin/bash
# tools/gpu_ab.sh "<tests -k expr>" "<workload:flagA,flagB ...>"   -- in-process A/B of tune flags per workload (same box, same clocks)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$1" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" > gpurun_out/step_tests.log 2>&1
  echo "tests rc=$?" >> gpurun_out/step_tests.log; tail -3 gpurun_out/step_tests.log
fi
for item in $2; do
  w=${item%%:*}; fl=${item#*:}
  timeout 600 python bench.py --workload $w --ab $fl --steps 8 > gpurun_out/ab_$w.log 2>&1
  grep "\[ab\] rep [12]" gpurun_out/ab_$w.log
done
