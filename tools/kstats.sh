#!/bin/bash
# tools/kstats.sh <workload> <tune-flags> -- per-kernel average durations (rocprofv3 --kernel-trace --stats)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python $REPO/bench.py --no-cpu-baseline --no-also --engine-door --workload $1 --steps 5 --tune-flags $2 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(f"{float(r['AverageNs'])/1e6:8.3f} ms  x{r['Calls']:>3}  {r['Name'][:70]}")
PY
