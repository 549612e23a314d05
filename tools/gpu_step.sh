#!/bin/bash
# one gpurun call of the build->measure loop: targeted tests, then in-process A/B and per-kernel times of the group-by workloads
#   tools/gpu_step.sh "<pytest -k expr>" "<workloads>" "<ab flag list>"
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
T="${1:-chunk or plane}"
WL="${2:-c3 c3w}"
AB="${3:-0,1048576}"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$T" > gpurun_out/step_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/step_tests.log
tail -5 gpurun_out/step_tests.log
for w in $WL; do
  timeout 600 python bench.py --workload $w --ab $AB --steps 5 > gpurun_out/step_ab_$w.log 2>&1
  grep "\[ab\]" gpurun_out/step_ab_$w.log | tail -4
  timeout 300 bash tools/kstats.sh $w 0 2>&1 | head -8
done
