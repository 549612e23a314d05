bash tools/gpu_step.sh "chunk or plane" "c3 c3w" "0,1048576"
echo GROUPS; timeout 600 python tools/bench_groups.py 2>&1 | tail -12
