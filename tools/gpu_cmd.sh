mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs_gpu.py -x -q -m gpu -k "chunk or plane or baseline or c3 or c5" 2>&1 | tail -4
for w in c3w; do
  timeout 600 python bench.py --workload $w --engine-door --ab 0 --steps 5 2>&1 | grep "\[ab\]" | tail -2
  echo LOAD_ALL; RFX_PLANE_LOAD_ALL=1 timeout 600 python bench.py --workload $w --engine-door --ab 0 --steps 5 2>&1 | grep "\[ab\]" | tail -2
  timeout 300 bash tools/kstats.sh $w 0 2>&1 | head -4
done
