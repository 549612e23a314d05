"""dev: what an UNPINNED column costs per query at the operator boundary -- the c3w shape through rfx_select on host columns that were never
pinned, so every use must prove the cached copy current: by soft-dirty page bits where the kernel has them, else by a checksum of the whole
payload (the default; RFX_SOFT_DIRTY=1 opts into the page bits).  Also prints what the page tracking costs the HOST: the time to write one cell
in every page of the three columns before anything was clean-marked and again right after the call that started the tracking (each first write
to a write-protected page is a minor fault), and the duration of that call (it contains the clear_refs walk over every page of the process).
tools/unpinned.py [rows=1e9] [reps=5]"""
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
def touch_every_page():
    """the host writing into its own columns: one cell per 4 KB page, same value back (the payload, and so the checksum, does not change)"""
    t = time.perf_counter()
    for a in host.values():
        a[::512] += 0
    return time.perf_counter() - t
print(f"host write pass over {3 * rows * 8 / 4096:.0f} pages before any tracking: {touch_every_page():.3f} s (again: {touch_every_page():.3f} s)")
t0 = time.perf_counter()
ops.rfx_host_drop(ops.rfx_select(d))
print(f"first query (uploads {3 * rows * 8 / 1e9:.0f} GB): {time.perf_counter() - t0:.3f} s")
for _ in range(4):  # two uses prove the columns stable, the next one starts the page tracking (one clear_refs)
    t1 = time.perf_counter()
    ops.rfx_host_drop(ops.rfx_select(d))
    print(f"  warm-up query {time.perf_counter() - t1:.3f} s")
print(f"host write pass right after the warm-up calls (pages write-protected by clear_refs when RFX_SOFT_DIRTY=1): {touch_every_page():.3f} s (again: {touch_every_page():.3f} s)")
for _ in range(4):
    ops.rfx_host_drop(ops.rfx_select(d))
s0 = H.to_numpy(ops.rfx_stats(0))
t0 = time.perf_counter()
for _ in range(reps):
    r = ops.rfx_select(d)
    assert r and not H.is_error(r), H.error_text(r)
    ops.rfx_host_drop(r)
dt = (time.perf_counter() - t0) / reps
s1 = H.to_numpy(ops.rfx_stats(0))
print(f"rows {rows} unpinned rfx_select {1e3 * dt:.2f} ms/query  cache hits {s1[5] - s0[5]}  of them by page bits {s1[11] - s0[11]}  on_gpu {int(ops.rfx_last_select_on_gpu())}  RFX_SOFT_DIRTY={os.environ.get('RFX_SOFT_DIRTY', '(default)')}")
