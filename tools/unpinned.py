"""dev: what an UNPINNED column costs per query at the operator boundary -- the c3w shape through rfx_select on host columns, three ways in one
process: (1) never pinned, validated BY OWNERSHIP (round 6, the default: the cache holds clone_obj on the column, a use is one pointer
compare), (2) the same columns after rfx_pin, (3) RFX_VALIDATE=checksum semantics (rfx_ops_set_validation(1): a checksum of the whole payload per
use, what rounds 1-5 did for unpinned columns; RFX_SOFT_DIRTY=1 opts into the page bits there).  Prints ms per query of each and the counters
that say which validation ran (rfx_stats[12] checksums, [13] pointer compares, [11] page bits).
tools/unpinned.py [rows=1e9] [reps=10]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rayforce_amd.engine import Engine
from rayforce_amd import hostobj as H
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = Engine(0)
ops = H.lib()
ops.rfx_host_bind()
host = {"k": eng.gen_i64(rows, 4, 1_000_000).cpu().numpy(), "v": eng.gen_f64(rows, 5).cpu().numpy(), "a": eng.gen_i64(rows, 2, 1_000_000).cpu().numpy()}
tab = H.table(host)
d = H.select_dict({"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}, tab)


def timed(label):
    t0 = time.perf_counter()
    ops.rfx_host_drop(ops.rfx_select(d))
    first = time.perf_counter() - t0
    for _ in range(4):
        ops.rfx_host_drop(ops.rfx_select(d))
    s0 = H.to_numpy(ops.rfx_stats(0))
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = ops.rfx_select(d)
        ts.append(time.perf_counter() - t0)
        assert r and not H.is_error(r), H.error_text(r)
        ops.rfx_host_drop(r)
    s1 = H.to_numpy(ops.rfx_stats(0))
    ts = np.array(ts) * 1e3
    print(f"{label:34s} rows {rows}  first call {first:.3f} s  then {np.median(ts):.2f} ms/query (median of {reps}; mean {ts.mean():.2f}, min {ts.min():.2f})  "
          f"cache hits {s1[5] - s0[5]}: by ownership {s1[13] - s0[13]}, checksums {s1[12] - s0[12]}, page bits {s1[11] - s0[11]}  uploads {s1[4] - s0[4]}  on_gpu {int(ops.rfx_last_select_on_gpu())}", flush=True)
    return float(np.median(ts))


own = timed("unpinned, by ownership (default)")
p = ops.rfx_pin(tab)
pinned = timed("pinned (rfx_pin)")
ops.rfx_host_drop(ops.rfx_unpin(tab))
ops.rfx_host_drop(p)
print(f"unpinned / pinned = {own / pinned:.3f}")
if os.environ.get("RFX_UNPINNED_SKIP_CHECKSUM") != "1":
    assert ops.rfx_ops_set_validation(1) == 0
    timed(f"unpinned, by checksum{' + RFX_SOFT_DIRTY=1' if os.environ.get('RFX_SOFT_DIRTY') == '1' else ''}")
    assert ops.rfx_ops_set_validation(0) == 0
