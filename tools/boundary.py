"""dev: the C3w shape through rfx_select on pinned host columns (RFX_TRACE=2 prints where each call's time goes).  tools/boundary.py [rows=1e8] [reps=5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = Engine(0)
ops = H.lib()
ops.rfx_host_bind()
cols = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5), "a": eng.gen_i64(rows, 2, 1_000_000)}
q = {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}
for _ in range(2):
    eng.select({"from": cols, **q})
eng.sync()
t0 = time.perf_counter()
for _ in range(reps):
    eng.select({"from": cols, **q})
eng.sync()
print(f"engine {1e3 * (time.perf_counter() - t0) / reps:.3f} ms")
host = {k: v.cpu().numpy() for k, v in cols.items()}
del cols
tab = H.table(host)
p = ops.rfx_pin(tab)
for name, qq in (("c3w", q), ("c3", {"by": "k", "s": ("sum", "v")})):
    d = H.select_dict(qq, tab)
    for _ in range(2):
        ops.rfx_host_drop(ops.rfx_select(d))
    t0 = time.perf_counter()
    for _ in range(reps):
        r = ops.rfx_select(d)
        assert r and not H.is_error(r), H.error_text(r)
        ops.rfx_host_drop(r)
    print(f"rfx_select {name} {1e3 * (time.perf_counter() - t0) / reps:.3f} ms  on_gpu {int(ops.rfx_last_select_on_gpu())}")
