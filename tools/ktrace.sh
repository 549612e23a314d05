#!/bin/bash
# tools/ktrace.sh <workload> <tune-flags> <kernel-substring> -- every dispatch's duration (ms) of the kernels matching the substring, in launch order
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --no-cpu-baseline --no-also --engine-door --workload $1 --steps 2 --warmup 1 --tune-flags $2 > /dev/null 2>&1
python - "$3" <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/kt/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if sys.argv[1] in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(' '.join(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:.2f}" for r in rows))
PY
