#!/bin/bash
# tools/profile_round.sh <round-tag>  -- run on the GPU box (via gpurun).  Writes rocprofv3 summaries to gpurun_out/prof_<tag>/.
# Pass 1: --kernel-trace --stats per workload (per-kernel durations).  Pass 2/3: --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# their own runs (never combined with trace domains other than kernel-trace, per the pool rules).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
WL=${2:-"c2 c2b c3 c3w q2 x6 q1 k9 c5 w2 q7"}   # optional second argument: only these workloads
for w in $WL; do
  DOOR="--engine-door"; [ "$w" = "c3w" ] && DOOR=""   # the headline workload: the same command as the bench line (rfx_select leg included)
  rm -rf /tmp/rp_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$w -o $w -- python $REPO/bench.py --no-cpu-baseline --no-also --no-predict --workload $w $DOOR --steps 10 > $OUT/${w}_bench.log 2>&1
  f=$(find /tmp/rp_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${w}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rpc_${w}_$c
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rpc_${w}_$c -o $w -- python $REPO/bench.py --no-cpu-baseline --no-also --workload $w --engine-door --steps 3 --warmup 1 > $OUT/${w}_pmc_$c.log 2>&1
    f=$(find /tmp/rpc_${w}_$c -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" "$c" > $OUT/${w}_pmc_$c.txt <<'PY'
import csv, sys, collections
path, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
with open(path) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") == ctr:
            agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
print(f"# {ctr}: per-dispatch mean of the raw counter value, by kernel (dispatch count)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/len(v):.1f}\t{len(v)}\t{k[:110]}")
PY
    fi
  done
done
ls -la $OUT
