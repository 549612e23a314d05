#!/bin/bash
# tools/pmc_any.sh <workload> <kernel-substring> "<counters set 1>" "<counters set 2>" ...
W=$1; K=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "$@"; do
  rm -rf /tmp/pa
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pa -o x -- python $REPO/bench.py --no-cpu-baseline --no-also --engine-door --workload $W --steps 2 --warmup 1 > /tmp/pa.log 2>&1
  f=$(find /tmp/pa -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no output for: $set"; tail -3 /tmp/pa.log; continue; }
  python - "$f" "$K" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if sys.argv[2] in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print({c: round(sum(v)/len(v)) for c, v in agg.items()})
PY
done
