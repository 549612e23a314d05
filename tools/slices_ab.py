"""Experiment (round 5): does the aggregate pass hide behind the scatter when the table is processed in ROW SLICES on two
streams?  No library change: two contexts (two streams) of the flat ABI alternate over the slices -- context i % 2 scatters slice i
(rfx_hip_group_scope: blocks until that scatter is done) while the OTHER context's aggregate of slice i - 1 (rfx_hip_group_dense_accumulate:
enqueued just before, asynchronous) runs beside it; every context folds into its own tables under a known key scope, one merge at the end.
    python tools/slices_ab.py [c3w|c3] [rows] [slice counts ...]
Prints ms per query for the unsliced form (one context, one slice) and for every slice count, plus the kernel-free floor (scatter only)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rayforce_amd import _lib as L  # noqa: E402
from rayforce_amd.engine import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3w"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
counts = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 16, 32]
eng = Engine(0)
lib = eng.lib
k, v = eng.gen_i64(rows, 4, 1_000_000), eng.gen_f64(rows, 5)
a = eng.gen_i64(rows, 2, 1_000_000) if name == "c3w" else None
eng.sync()
ctxs = []
for _ in range(2):
    c = C.c_void_p()
    L.check(lib.rfx_hip_ctx_create(0, None, C.byref(c)), "ctx_create")
    ctxs.append(c)
KMIN, RANGE = 0, 1_000_000
tables = []
for c in ctxs:
    first = torch.empty(RANGE, dtype=torch.int64, device="cuda")
    acc = torch.empty(RANGE, dtype=torch.float64, device="cuda")
    gt = L.GroupTables()
    gt.kmin, gt.range, gt.nagg = KMIN, RANGE, 1
    gt.d_first, gt.d_acc[0] = first.data_ptr(), acc.data_ptr()
    tables.append((gt, first, acc))


def descr(r0, n):
    ag = (L.Agg * 1)()
    ag[0].kind, ag[0].d_col, ag[0].col_type = L.AGGS["sum"], v.data_ptr() + r0 * 8, L.RFX_F64
    pr = (L.Pred * 1)()
    npred = 0
    if a is not None:
        pr[0].d_col, pr[0].col_type, pr[0].op, pr[0].rhs_type, pr[0].rhs_i = a.data_ptr() + r0 * 8, L.RFX_I64, L.OPS["<"], L.RFX_I64, 100_000
        npred = 1
    return ag, pr, npred


def query(nsl, aggregate=True):
    span = ((rows + nsl - 1) // nsl + 511) & ~511
    used = set()
    pending = None
    mn, mx, seen = C.c_int64(), C.c_int64(), C.c_int64()
    for i in range(nsl):
        r0 = i * span
        n = min(span, rows - r0)
        if n <= 0:
            break
        j = i % 2 if nsl > 1 else 0
        c, (gt, _, _) = ctxs[j], tables[j]
        ag, pr, npred = descr(r0, n)
        if j not in used:
            L.check(lib.rfx_hip_group_tables_init(c, ag, C.byref(gt)), "tables_init")
            used.add(j)
        if pending is not None and aggregate:  # the previous slice's aggregate, asynchronous, on the OTHER stream
            pc, pgt, pag, ppr, pnp, pr0, pn = pending
            L.check(lib.rfx_hip_group_dense_accumulate(pc, C.c_void_p(k.data_ptr() + pr0 * 8), ppr, pnp, L.RFX_AND, pag, pn, pr0, C.byref(pgt)), "accumulate")
        L.check(lib.rfx_hip_group_scope(c, C.c_void_p(k.data_ptr() + r0 * 8), pr, npred, L.RFX_AND, ag, 1, n, C.byref(mn), C.byref(mx), C.byref(seen)), "scope")
        pending = (c, gt, ag, pr, npred, r0, n)
    if pending is not None and aggregate:
        pc, pgt, pag, ppr, pnp, pr0, pn = pending
        L.check(lib.rfx_hip_group_dense_accumulate(pc, C.c_void_p(k.data_ptr() + pr0 * 8), ppr, pnp, L.RFX_AND, pag, pn, pr0, C.byref(pgt)), "accumulate")
    for c in ctxs:
        L.check(lib.rfx_hip_ctx_sync(c), "sync")
    if len(used) == 2 and aggregate:
        ag, _, _ = descr(0, rows)
        L.check(lib.rfx_hip_group_tables_merge(ctxs[0], ag, C.byref(tables[0][0]), C.byref(tables[1][0])), "merge")
        L.check(lib.rfx_hip_ctx_sync(ctxs[0]), "sync")


def timed(nsl, aggregate=True, reps=8):
    for _ in range(2):
        query(nsl, aggregate)
    t0 = time.perf_counter()
    for _ in range(reps):
        query(nsl, aggregate)
    return (time.perf_counter() - t0) * 1e3 / reps


ref = None
for nsl in counts:
    ms = timed(nsl)
    if nsl == 1:
        ref = tables[0][2].clone()
    elif ref is not None:
        err = float((tables[0][2] - ref).abs().max())
        assert err <= 1e-6, err
    print(f"{name} rows {rows} slices {nsl:3d}: {ms:7.3f} ms per query (scatter only: {timed(nsl, False):7.3f})"
          f"   plane_scatter launches so far {sum(int(lib.rfx_hip_ctx_stat(c, 0)) for c in ctxs)}", flush=True)
