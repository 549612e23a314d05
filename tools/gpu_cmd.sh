bash tools/gpu_step.sh "chunk or plane" "c3 c3w" "0,1048576"
echo PBITS8; RFX_PLANE_PBITS=8 timeout 300 python bench.py --workload c3 --ab 0 --steps 5 2>&1 | grep "\[ab\]" | tail -1
RFX_PLANE_PBITS=8 timeout 300 bash tools/kstats.sh c3 0 2>&1 | head -3
echo PBITS8 c3w; RFX_PLANE_PBITS=8 timeout 300 python bench.py --workload c3w --ab 0 --steps 5 2>&1 | grep "\[ab\]" | tail -1
RFX_PLANE_PBITS=8 timeout 300 bash tools/kstats.sh c3w 0 2>&1 | head -3
