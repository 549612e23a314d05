#!/bin/bash
# round 6, first GPU call: the tests touched by residency-by-ownership / visibility / parallel uploads / sharded joins / sparse group, then the default bench,
# the unpinned-vs-pinned measurement and the H2D rates.  Everything it keeps goes to gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_abi_cpu.py tests/test_ops_gpu.py tests/test_mapgroup_gpu.py tests/test_dropin_gpu.py tests/test_sharded_gpu.py tests/test_fuzz_tools_gpu.py tests/test_io_gpu.py -x -q -m "gpu or not gpu" -p no:cacheprovider > gpurun_out/r06a_tests.txt 2>&1
tail -15 gpurun_out/r06a_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -1 gpurun_out/r06a_bench.err | cut -c1-1500
timeout 600 python tools/unpinned.py 1e9 10 > gpurun_out/r06a_unpinned.txt 2>&1
cat gpurun_out/r06a_unpinned.txt | grep -v amdgpu.ids
timeout 300 python tools/h2d_bench.py 1000000000 > gpurun_out/r06a_h2d.txt 2>&1
cat gpurun_out/r06a_h2d.txt | grep -v amdgpu.ids
