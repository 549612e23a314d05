"""dev: rank + emit as one persistent launch against the ten launches of rounds 1-4 (the default; RFX_FUSED_RANK=1 takes the one launch), the c3w shape through rfx_select
with the planner's phase timers; and the six-key row-hash result (6.4 GB of host columns) through the door: python tools/rank_ab.py [q7]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
eng = Engine(0)
ops = H.lib(); ops.rfx_host_bind()
if len(sys.argv) > 1 and sys.argv[1] == "q7":
    r = bench.c_door("q7", eng, bench.WORKLOADS["q7"]["rows"], 4, 2, device_columns=True)
    print("q7 through rfx_select:", r["ms_per_step"], "ms", r["steps_ms"], file=sys.stderr)
    sys.exit(0)
rows = int(float(os.environ.get("DOOR_ROWS", "1e9")))
spec, q = bench.C_DOOR["c3w"]
cols = bench.door_columns(eng, spec, rows); eng.sync()
tab = H.device_table(cols); d = H.select_dict(q, tab)
dt, got = bench.door_run(ops, H, d, 20, 5)
print(f"fused={'yes' if os.environ.get('RFX_FUSED_RANK') else 'no'} rows {rows}: {dt * 1e3 / 20:.3f} ms/query {bench.door_run.last_steps_ms}", file=sys.stderr)
print({k: round(v, 4) for k, v in bench.door_phases(ops, H, d, 10).items() if not isinstance(v, list)}, file=sys.stderr)
