"""dev: the planner's phase timers for the q7 shape through Engine (rfx_exec_timing), packed table on / off (RFX_NO_PACKED_TABLE=1).  tools/q7_phases.py [steps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rayforce_amd.engine import Engine
from rayforce_amd import _lib as L
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
eng = Engine(0)
job = bench.Job("q7", eng, None, bench.WORKLOADS["q7"]["rows"], 0)
for _ in range(2):
    job.step()
eng.sync()
eng.lib.rfx_exec_timing(eng._x, 1)
names = dict(L.RFX_XSTAT_PHASES)
b = {k: eng.xstat(v) for k, v in names.items()}
t0 = time.perf_counter()
for _ in range(steps):
    job.step()
eng.sync()
wall = (time.perf_counter() - t0) / steps * 1e3
print("packed" if not os.environ.get("RFX_NO_PACKED_TABLE") else "field by field", "wall ms/step", round(wall, 2), {k: round((eng.xstat(v) - b[k]) / steps / 1e6, 2) for k, v in names.items()}, flush=True)
