"""rfx_update (atoms, element-wise mappings, per-group aggregates; where / by) and rfx_group + MAPGROUP folds over adversarial cells at a few
row counts, against the oracle (one shard: both are the host's under shards).  Shapes handed back are counted, with their reason.
python tools/fuzz_update_group.py <first seed> <last seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import numpy as np
from oracle import rfo
from rayforce_amd import hostobj as H
from fuzz_operators_cols import col_i64, col_f64

NULL = -(2**63)
ops = H.lib()
ops.rfx_host_bind()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = handed = 0
why = {}


def same(g, w, what, tol=False):
    assert g.dtype == w.dtype and g.shape == w.shape, (what, g.dtype, w.dtype, g.shape, w.shape)
    if w.dtype == np.float64:
        assert np.array_equal(np.isnan(g), np.isnan(w)), (what, "nan pattern")
        fin = np.isfinite(w)
        assert np.array_equal(g[~fin & ~np.isnan(w)], w[~fin & ~np.isnan(w)]), (what, "inf")
        if tol:
            assert np.allclose(g[fin], w[fin], rtol=1e-9, atol=1e-300), what
        else:
            assert np.array_equal(g[fin], w[fin]), what
    else:
        assert np.array_equal(g, w), what


for seed in range(lo, hi):
    rng = np.random.default_rng(33_000 + seed)
    n = int(rng.choice([1, 2, 64, 4099, 70_001, 300_007]))
    kk = int(rng.choice([1, 3, 50, 3000]))
    t = {"k": rng.integers(0, kk, n).astype(np.int64) + int(rng.integers(-5, 5)), "a": col_i64(rng, n), "b": col_i64(rng, n), "v": col_f64(rng, n), "w": col_f64(rng, n)}
    big = {c for c in ("v", "w") if int((np.abs(t[c]) >= 1e300).sum()) > 1} | {c for c in ("a", "b") if int((np.abs(t[c].astype(np.float64)) >= 2.0**61).sum()) > 1}
    what = None
    try:
        # ---- update
        kind = int(rng.integers(0, 6))
        ic, fc = str(rng.choice(["a", "b"])), str(rng.choice(["v", "w"]))
        if kind == 0:
            q = {fc: float(rng.choice([0.0, -0.0, 1.5, float("inf")]))}
        elif kind == 1:
            q = {ic: int(rng.choice([0, 7, NULL, 2**62]))}
        elif kind == 2:
            q = {fc: (str(rng.choice(["+", "-", "*"])), fc, float(rng.choice([1.5, -2.0, 0.0])))}
        elif kind == 3:
            q = {ic: (str(rng.choice(["+", "-", "*"])), ic, str(rng.choice(["k", "a", "b"])))}
        elif kind == 4:
            q = {"n1": (str(rng.choice(["sum", "min", "max", "count", "first"])), ic), "by": "k"}
        else:
            q = {"n2": (str(rng.choice(["sum", "avg", "min", "max", "first"])), fc), "by": "k"}
        if rng.random() < 0.6:
            c = str(rng.choice(["a", "b", "v", "w", "k"]))
            q["where"] = (str(rng.choice(["<", ">", "<=", ">=", "!=", "=="])), c, [0, NULL, 2**62, 0.0, -0.5, float("nan"), 2][int(rng.integers(0, 7))])
        what = ("update", q)
        tab = H.table(t)
        d = H.select_dict(q, tab)
        r = ops.rfx_update(d)
        if H.is_error(r):
            handed += 1
            msg = H.error_text(r)
            key = msg[msg.find("("):][:60]
            why[key] = why.get(key, 0) + 1
        else:
            got, want = H.table_to_numpy(r), rfo.update({"from": t, **q})
            assert list(got) == list(want), (what, list(got), list(want))
            for name in want:
                agg = name in q and isinstance(q[name], tuple) and q[name][0] in ("sum", "avg")
                if agg and q[name][1] in big:
                    continue  # (overflow on the way: order decides -- DESIGN.md deviation 2 / f64 +-1e308)
                same(got[name], want[name], (what, name), tol=agg)
        for o in (r, d, tab):
            ops.rfx_host_drop(o)
        # ---- group + MAPGROUP folds
        what = ("group",)
        kv = H.vector(t["k"])
        r = ops.rfx_group(kv)
        if H.is_error(r):
            handed += 1
            why["group"] = why.get("group", 0) + 1
            ops.rfx_host_drop(r)
        else:
            gids, firsts, groups, dense = rfo.group_index(t["k"], None)
            items = H.list_items(r)
            itype = C.c_int64.from_address(items[0] + 8).value
            assert C.c_int64.from_address(items[1] + 8).value == groups, (what, "groups")
            if dense:
                assert np.array_equal(H.to_numpy(items[6]), firsts), (what, "firsts")
            else:  # sparse keys (round 6): the IDS flavour of index_group_i64_unscoped carries no first rows (core/index.c:1959-1977)
                assert itype == 0 and H.header(items[6]).type == 126, (what, "sparse index: no first rows")
            if itype == 0:
                assert np.array_equal(H.to_numpy(items[2]), gids), (what, "ids")
            for cname in (ic, fc):
                for fn in ("sum", "min", "max", "avg", "count", "first"):
                    what = ("mapgroup", fn, cname)
                    pr = ops.rfx_host_list(2)
                    arr = (C.c_void_p * 2).from_address(H.payload(pr))
                    arr[0], arr[1] = H.vector(t[cname]), ops.rfx_host_clone(r)
                    H.header(pr).type = 72
                    g = getattr(ops, f"rfx_{fn}")(pr)
                    if H.is_error(g):
                        handed += 1
                        why["mapgroup " + fn] = why.get("mapgroup " + fn, 0) + 1
                    elif not (fn in ("sum", "avg") and cname in big):
                        want = rfo.select({"from": t, "by": "k", "o": (fn, cname)})["o"]
                        same(H.to_numpy(g), want, what, tol=fn in ("sum", "avg"))
                    ops.rfx_host_drop(g)
                    ops.rfx_host_drop(pr)
            ops.rfx_host_drop(r)
        ops.rfx_host_drop(kv)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "n", n, what, "->", repr(e)[:300], flush=True)
print("done", hi - lo, "seeds,", handed, "handed back", why, ",", bad, "failures")
