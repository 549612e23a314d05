#!/bin/bash
# tools/atomics_pmc.sh -- WHERE do device-memory atomics execute?  Runs tools/atomics_probe under rocprofv3 --pmc (own run, kernel-trace only)
# and prints, per kernel instantiation, the mean L2 counters per issued atomic.  Output: gpurun_out/atomics_pmc_r02.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LG=${1:-24}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/apmc
for set in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_ATOMIC_SECTORS_sum"; do
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/apmc/$(echo $set | cut -c1-12) -o a -- $REPO/tools/atomics_probe $LG > /tmp/apmc.log 2>&1
done
python - "$LG" > $REPO/gpurun_out/atomics_pmc_r02.json <<'PY'
import csv, glob, json, sys, collections
rows = 1 << int(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/apmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_atomics' in r['Kernel_Name']:
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
names = {0: 'f64add', 1: 'u64add', 2: 'u64min', 3: 'f32add', 4: 'u32add'}
for k, d in agg.items():
    t = k[k.index('<') + 1:k.index('>')].replace(' ', '').split(',')
    label = f"{names[int(t[0])]}/{'workgroup' if int(t[1]) else 'agent'}-scope/{'per-XCD tables' if int(t[2]) else 'one table'}{'/sorted-in-wave' if int(t[3]) else ''}"
    out[label] = {c: round(sum(v) / len(v) / rows, 4) for c, v in d.items()}
    out[label]['dispatches'] = len(next(iter(d.values())))
print(json.dumps({"unit": "counter per issued atomic (mean over all table sizes 64 KB .. 64 MB and allocation kinds)", "rows_per_dispatch": rows, "kernels": out}, indent=1))
PY
cat $REPO/gpurun_out/atomics_pmc_r02.json | head -60
