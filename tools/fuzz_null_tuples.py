"""Key tuples with nulls among the keys through rfx_select (standalone host model) against the oracle -- meant for RFX_SHARDS=k
[RFX_EXEC_SLICE_SHARDS=1], where the tuple proof is a (min, max) pair per key column and a null key rides as max + 1:
python tools/fuzz_null_tuples.py <first seed> <last seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ctypes as C
from oracle import rfo
from rayforce_amd import hostobj as H
from test_gpu_parity import same_f64
sys.path.insert(0, os.path.join(ROOT, "tools"))

NULL = -(2**63)
ops = H.lib()
ops.rfx_host_bind()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = handed = 0
import gen_case
for seed in range(lo, hi):
    n, t, by, kinds, q = gen_case.gen(seed)
    if os.environ.get("FUZZ_CLEAR") == "trim":
        x = ops.rfx_ops_exec()
        if x:
            ops.rfx_hip_ctx_trim(C.c_void_p(ops.rfx_exec_ctx(C.c_void_p(x), 0)))
    elif os.environ.get("FUZZ_CLEAR"):
        ops.rfx_cache_clear()
    try:
        want = rfo.select({"from": t, **q})
        tab = H.table(t)
        d = H.select_dict(q, tab)
        for rep in range(2):
            r = ops.rfx_select(d)
            if H.is_error(r):
                handed += 1
                print("HANDED", seed, n, kinds, sorted(k for k in q if k != "by"), H.error_text(r)[60:130], flush=True)
                ops.rfx_host_drop(r)
                break
            got = H.table_to_numpy(r)
            ops.rfx_host_drop(r)
            assert list(got) == list(want), (list(got), list(want))
            for name in want:
                g, w = got[name], want[name]
                assert g.dtype == w.dtype and g.shape == w.shape, (name, g.dtype, w.dtype, g.shape, w.shape)
                if w.dtype == np.float64 and name in q and q[name][0] in ("sum", "avg"):
                    same_f64(g, w)
                else:
                    assert np.array_equal(g, w, equal_nan=w.dtype == np.float64), name
        ops.rfx_host_drop(d)
        ops.rfx_host_drop(tab)
    except Exception as e:  # noqa: BLE001
        bad += 1
        if n <= 2:
            for k_ in t: print("   col", k_, t[k_])
            for k_ in want: print("   ", k_, "got", got.get(k_), "want", want[k_])
            print("   stats", [int(v) for v in H.stats(ops)] if hasattr(H, "stats") else "")
        print("SEED", seed, "rep", rep, "n", n, kinds, {k: v for k, v in q.items()}, "->", repr(e)[:300], flush=True)
print("done", hi - lo, "seeds,", handed, "handed back,", bad, "failures")
