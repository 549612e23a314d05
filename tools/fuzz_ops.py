"""One-off: the fixed-seed query generator of tests/test_gpu_fuzz.py through the C OPERATOR layer (rfx_select on host objects, standalone host model)
against the oracle -- every query asked TWICE over the same table object (the second time finds columns resident and the key column's scope
remembered).  Shapes the operator hands back (no host to hand them to here: an error object) are counted, not compared.
python tools/fuzz_ops.py <first seed> <last seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import rfo
from rayforce_amd import hostobj as H
import test_gpu_fuzz as F
from test_gpu_parity import same_f64, _abs_scale

ops = H.lib()
ops.rfx_host_bind()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = handed = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    t, q = F.make_case(rng)
    try:
        try:
            want = rfo.select({"from": t, **q})
        except rfo.NotPerfect:
            continue
        if isinstance(q.get("by"), dict) and len(q["by"]) > 1 and "where" in q:
            continue  # (where: with several by: columns is handed back by design: the reference's own result is defective)
        tab = H.table(t)
        d = H.select_dict(q, tab)
        for rep in range(2):
            r = ops.rfx_select(d)
            if H.is_error(r):
                handed += 1
                ops.rfx_host_drop(r)
                break
            got = H.table_to_numpy(r)
            ops.rfx_host_drop(r)
            assert list(got) == list(want), (list(got), list(want))
            for name in want:
                g, w = got[name], want[name]
                assert g.dtype == w.dtype, (name, g.dtype, w.dtype)
                if w.dtype == np.float64:
                    same_f64(g, w, scale=_abs_scale(t, q, name) if name in q else None)
                else:
                    assert np.array_equal(g, w), name
        ops.rfx_host_drop(d)
        ops.rfx_host_drop(tab)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "n", len(t["k"]), {k: v for k, v in q.items()}, "->", repr(e)[:300], flush=True)
print("done", hi - lo, "seeds,", handed, "handed back,", bad, "failures")
