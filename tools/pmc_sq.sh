#!/bin/bash
# tools/pmc_sq.sh <workload> -- SQ counters of every kernel of one bench workload (own rocprofv3 run, kernel-trace only)
set -u
W=${1:-c3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/sq_$W
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$W -o $W -- python $REPO/bench.py --no-cpu-baseline --no-also --engine-door --workload $W --steps 2 --warmup 1 > /tmp/sq_$W.log 2>&1
  f=$(find /tmp/sq_$W -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        agg[row["Kernel_Name"].split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if not any(x in k for x in ("k_part", "k_filter_aggr<", "k_group", "k_sel", "k_emit")): continue
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
done
