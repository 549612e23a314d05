#!/bin/bash
# tools/timeline.sh <workload> -- every kernel / memory operation of the LAST timed step in launch order: start offset (us), duration (us), name
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o t -- python $REPO/bench.py --no-cpu-baseline --no-also --engine-door --workload $1 --steps 3 --warmup 2 > /dev/null 2>&1
python - <<'PY'
import csv,glob
rows=[]
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]))
for f in glob.glob('/tmp/tl/**/*memory_copy_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')+' '+r.get('Bytes','')))
rows.sort()
# last occurrence of the scope sample kernel = start of the last step
idx=[i for i,r in enumerate(rows) if 'k_scope_sample' in r[2]]
s=idx[-1] if idx else max(0,len(rows)-30)
t0=rows[s][0]
for a,b,n in rows[s:s+40]: print(f"{(a-t0)/1e3:9.1f} us  {(b-a)/1e3:8.1f} us  {n}")
PY
