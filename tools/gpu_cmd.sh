#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "two_level" > gpurun_out/t_tree.log 2>&1; echo "tree rc=$?"; tail -15 gpurun_out/t_tree.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dropin_gpu.py tests/test_c_host_gpu.py -x -q -m gpu > gpurun_out/t_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/t_ops.log
for w in c2 c2b c5 c3w; do
  timeout 600 python bench.py --workload $w --engine-door --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['roofline']['frac'])"
done
