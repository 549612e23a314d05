"""dev: where a select's wall time goes at the C door (RFX_TRACE=2 prints microseconds between marks per query): the c3w shape on pinned host columns."""
import os, sys, time
os.environ["RFX_TRACE"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
eng = Engine(0)
ops = H.lib(); ops.rfx_host_bind()
host = {"k": eng.gen_i64(rows, 4, 1_000_000).cpu().numpy(), "v": eng.gen_f64(rows, 5).cpu().numpy(), "a": eng.gen_i64(rows, 2, 1_000_000).cpu().numpy()}
tab = H.table(host); pin = ops.rfx_pin(tab)
d = H.select_dict({"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}, tab)
for i in range(8):
    t0 = time.perf_counter(); r = ops.rfx_select(d); dt = time.perf_counter() - t0
    print(f"query {i}: {dt * 1e3:.3f} ms", file=sys.stderr, flush=True); ops.rfx_host_drop(r)
