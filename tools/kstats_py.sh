#!/bin/bash
# tools/kstats_py.sh <python script> -- per-kernel average durations and call counts of an arbitrary script (rocprofv3 --kernel-trace --stats)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ksp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksp -o k -- python $REPO/$1 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ksp/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us  x{r['Calls']:>4}  {r['Name'][:80]}")
PY
