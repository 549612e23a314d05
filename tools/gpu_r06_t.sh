#!/bin/bash
# round 6, the last tree: the whole GPU suite, smoke, the default bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider ) > gpurun_out/r06t_suite.txt 2>&1; tail -8 gpurun_out/r06t_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06t_bench.json 2> gpurun_out/r06t_bench.err; tail -1 gpurun_out/r06t_bench.err | cut -c1-1800
