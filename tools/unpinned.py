"""dev: what an UNPINNED column costs per query at the operator boundary -- the c3w shape through rfx_select on host columns that were never
pinned, so every use must prove the cached copy current: by soft-dirty page bits where the kernel has them, else by a checksum of the whole
payload (RFX_SOFT_DIRTY=0 forces that).  tools/unpinned.py [rows=1e9] [reps=5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = Engine(0)
ops = H.lib()
ops.rfx_host_bind()
host = {"k": eng.gen_i64(rows, 4, 1_000_000).cpu().numpy(), "v": eng.gen_f64(rows, 5).cpu().numpy(), "a": eng.gen_i64(rows, 2, 1_000_000).cpu().numpy()}
tab = H.table(host)
d = H.select_dict({"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}, tab)
t0 = time.perf_counter()
ops.rfx_host_drop(ops.rfx_select(d))
print(f"first query (uploads {3 * rows * 8 / 1e9:.0f} GB): {time.perf_counter() - t0:.3f} s")
for _ in range(4):  # two uses prove the columns stable, the next one starts the page tracking (one clear_refs)
    t1 = time.perf_counter()
    ops.rfx_host_drop(ops.rfx_select(d))
    print(f"  warm-up query {time.perf_counter() - t1:.3f} s")
s0 = H.to_numpy(ops.rfx_stats(0))
t0 = time.perf_counter()
for _ in range(reps):
    r = ops.rfx_select(d)
    assert r and not H.is_error(r), H.error_text(r)
    ops.rfx_host_drop(r)
dt = (time.perf_counter() - t0) / reps
s1 = H.to_numpy(ops.rfx_stats(0))
print(f"rows {rows} unpinned rfx_select {1e3 * dt:.2f} ms/query  cache hits {s1[5] - s0[5]}  of them by page bits {s1[11] - s0[11]}  on_gpu {int(ops.rfx_last_select_on_gpu())}  RFX_SOFT_DIRTY={os.environ.get('RFX_SOFT_DIRTY', '(default)')}")
