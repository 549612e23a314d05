mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python bench.py > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err; echo bench rc=$?
tail -5 gpurun_out/bench_r03a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r03a.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['roofline']['frac'], d.get('door',{}).get('ms_per_step'), d.get('engine'))
for k,v in d.get('also',{}).items(): print(k, {x:v.get(x) for x in ('ms_per_step','frac','verified','paths','error')})
print(d.get('boundary')); print(d.get('cpu_baseline'))
PY
