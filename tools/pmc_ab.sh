#!/bin/bash
# tools/pmc_ab.sh <workload> <kernel-substring> <flags> "<counter set>" ...
W=$1; K=$2; F=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "$@"; do
  rm -rf /tmp/pa
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pa -o x -- python $REPO/bench.py --no-cpu-baseline --no-also --workload $W --steps 2 --warmup 1 --tune-flags $F > /tmp/pa.log 2>&1
  f=$(find /tmp/pa -name "*counter_collection.csv" | head -1)
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
