"""dev: where a select's wall time goes at the C door (RFX_TRACE=2 prints microseconds between marks per query): the c3w shape on pinned host
columns, on device column handles over torch-owned memory, and through Engine, in one process."""
import os, sys, time
os.environ["RFX_TRACE"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
eng = Engine(0)
ops = H.lib(); ops.rfx_host_bind()
q = {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}
cols = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5), "a": eng.gen_i64(rows, 2, 1_000_000)}
def timed(label, fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"## {label}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/query", file=sys.stderr, flush=True)
timed("Engine.group_by (torch-owned columns)", lambda: eng.group_by("k", [("sum", "v")], ("<", "a", 100_000), cols))
dtab = H.device_table(cols); dd = H.select_dict(q, dtab)
timed("rfx_select on device handles (torch-owned columns)", lambda: ops.rfx_host_drop(ops.rfx_select(dd)))
host = {c: t.cpu().numpy() for c, t in cols.items()}
del cols, dtab; torch.cuda.empty_cache()
tab = H.table(host); pin = ops.rfx_pin(tab)
d = H.select_dict(q, tab)
timed("rfx_select on pinned host columns (library-owned device copies)", lambda: ops.rfx_host_drop(ops.rfx_select(d)))
