"""dev: the c3w shape through rfx_select on device handles at a given row count (default 1.25e8 = one of eight devices' share), 30 queries:
for a rocprofv3 --kernel-trace --stats run (tools/kstats_py.sh tools/door_small.py) -- which kernels a per-device pass is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
rows = int(float(os.environ.get("DOOR_ROWS", "125000000")))
eng = Engine(0)
ops = H.lib(); ops.rfx_host_bind()
q = {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}
cols = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5), "a": eng.gen_i64(rows, 2, 1_000_000)}
eng.sync()
dtab = H.device_table(cols); dd = H.select_dict(q, dtab)
for _ in range(3): ops.rfx_host_drop(ops.rfx_select(dd))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): ops.rfx_host_drop(ops.rfx_select(dd))
torch.cuda.synchronize()
print(f"## rows {rows}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/query", file=sys.stderr, flush=True)
