"""The operators beside rfx_select (folds of vectors and of lazy MAPFILTER pairs, comparisons, arithmetic, and / or, where, at) over columns of
adversarial cells -- nulls, NaN, +-inf, -0.0, the ends of the i64 range -- and a FEW row counts (the stand-in host recycles addresses: every
call meets the residency cache with a same-sized predecessor), against the oracle.  Meant for RFX_SHARDS=1 and k:
python tools/fuzz_operators.py <first seed> <last seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
from oracle import rfo
from rayforce_amd import hostobj as H

NULL = -(2**63)
ops = H.lib()
ops.rfx_host_bind()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = handed = calls = 0


sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_operators_cols import col_i64, col_f64  # noqa: E402


def atom_value(r):
    h = H.header(r)
    v = C.c_double.from_address(r + 8).value if h.type == -H.T_F64 else C.c_int64.from_address(r + 8).value
    return v


def same_scalar(got, want, fn, f64):
    if want is None:
        return got == NULL or got != got
    if isinstance(want, float):
        if want != want:
            return got != got
        if fn in ("sum", "avg") and np.isfinite(want):
            return abs(got - want) <= 1e-9 * max(abs(want), 1e-300) + 1e-12
        return got == want or (got != got and want != want)
    return got == want


for seed in range(lo, hi):
    rng = np.random.default_rng(55_000 + seed)
    n = int(rng.choice([1, 2, 7, 64, 4099, 70_001, 300_007]))
    a, b, v, w = col_i64(rng, n), col_i64(rng, n), col_f64(rng, n), col_f64(rng, n)
    cols = {"a": a, "b": b, "v": v, "w": w}
    vecs = {k: H.vector(x) for k, x in cols.items()}
    what = None
    try:
        # folds of vectors and of MAPFILTER pairs
        sel = np.nonzero(rng.random(n) < rng.choice([0.0, 0.02, 0.5, 1.0]))[0].astype(np.int64)
        idv = H.vector(sel)
        for cname in ("a", "v"):
            pair = H.list_of([H.vector(cols[cname]), idv]) if False else None
            for fn in ("sum", "min", "max", "avg", "count", "first"):
                what = (fn, cname, "vector")
                r = getattr(ops, f"rfx_{fn}")(vecs[cname]); calls += 1
                big = cname == "v" and fn in ("sum", "avg") and int((np.abs(cols[cname]) >= 1e300).sum()) > 1  # (a sum that overflows on the way depends on its order)
                if H.is_error(r):
                    handed += 1
                elif not big:
                    assert same_scalar(atom_value(r), rfo.fold(fn, cols[cname]), fn, cname == "v"), (what, atom_value(r), rfo.fold(fn, cols[cname]))
                ops.rfx_host_drop(r)
                what = (fn, cname, "mapfilter", len(sel))
                pr = ops.rfx_host_list(2)
                arr = (C.c_void_p * 2).from_address(H.payload(pr))
                arr[0], arr[1] = ops.rfx_host_clone(vecs[cname]), ops.rfx_host_clone(idv)
                H.header(pr).type = 71
                r = getattr(ops, f"rfx_{fn}")(pr); calls += 1
                if H.is_error(r):
                    handed += 1
                elif not big:
                    assert same_scalar(atom_value(r), len(sel) if fn == "count" else rfo.fold(fn, cols[cname][sel]), fn, cname == "v"), (what, atom_value(r), rfo.fold(fn, cols[cname][sel]))
                ops.rfx_host_drop(r)
                ops.rfx_host_drop(pr)
        # comparisons: vector (x) atom, vector (x) vector, mixed types
        for _ in range(6):
            op = str(rng.choice(["==", "!=", "<", ">", "<=", ">="]))
            l = str(rng.choice(["a", "b", "v", "w"]))
            if rng.random() < 0.5:
                rn = str(rng.choice(["a", "b", "v", "w"]))
                rv, ro = cols[rn], vecs[rn]
                own = False
            else:
                rv = [int(rng.integers(-3, 3)), NULL, 2**63 - 1, 0.0, float("nan"), -0.5, float("inf")][int(rng.integers(0, 7))]
                ro, own = H.atom(rv), True
            what = ("cmp", op, l, rv if own else rn)
            r = getattr(ops, {"==": "rfx_eq", "!=": "rfx_ne", "<": "rfx_lt", ">": "rfx_gt", "<=": "rfx_le", ">=": "rfx_ge"}[op])(vecs[l], ro); calls += 1
            if H.is_error(r):
                handed += 1
            else:
                got = H.to_numpy(r).astype(bool)
                want = rfo.cmp(op, cols[l], rv).astype(bool)
                assert np.array_equal(got, want), (what, int((got != want).sum()))
            ops.rfx_host_drop(r)
            if own:
                ops.rfx_host_drop(ro)
        # arithmetic
        for _ in range(4):
            op = str(rng.choice(["+", "-", "*", "div"]))
            l = str(rng.choice(["a", "b", "v", "w"]))
            if rng.random() < 0.5:
                rn = str(rng.choice(["a", "b", "v", "w"]))
                rv, ro, own = cols[rn], vecs[rn], False
            else:
                rv = [int(rng.integers(-3, 3)), 0, 2**62, 0.0, -2.5, 1e300][int(rng.integers(0, 6))]
                ro, own = H.atom(rv), True
            what = ("arith", op, l, rv if own else rn)
            r = getattr(ops, {"+": "rfx_add", "-": "rfx_sub", "*": "rfx_mul", "div": "rfx_div"}[op])(vecs[l], ro); calls += 1
            if H.is_error(r):
                handed += 1
            else:
                got, want = H.to_numpy(r), rfo.binop(op, cols[l], rv)
                assert got.dtype == want.dtype, (what, got.dtype, want.dtype)
                if op == "div" and own:  # f64 `div` by an atom: the reference's fast-math build multiplies by the reciprocal (<= 1 ulp from the division made here)
                    fin = np.isfinite(want)
                    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), what
                    assert np.all(np.abs(got[fin] - want[fin]) <= 2.3e-16 * np.abs(want[fin])), what
                else:
                    assert np.array_equal(got.view(np.int64), want.view(np.int64)) or np.array_equal(got, want, equal_nan=got.dtype == np.float64), (what, int((got != want).sum()))
            ops.rfx_host_drop(r)
            if own:
                ops.rfx_host_drop(ro)
        # and / or / where / at
        m1, m2 = rng.random(n) < rng.choice([0.0, 0.3, 1.0]), rng.random(n) < 0.5
        hm = (C.c_void_p * 2)(H.vector(m1), H.vector(m2))
        for fn, want in (("rfx_and", m1 & m2), ("rfx_or", m1 | m2)):
            what = (fn,)
            r = getattr(ops, fn)(hm, 2); calls += 1
            if H.is_error(r):
                handed += 1
            else:
                assert np.array_equal(H.to_numpy(r).astype(bool), want), what
            ops.rfx_host_drop(r)
        what = ("where",)
        r = ops.rfx_where(hm[0]); calls += 1
        if H.is_error(r):
            handed += 1
        else:
            assert np.array_equal(H.to_numpy(r), np.nonzero(m1)[0]), what
        ops.rfx_host_drop(r)
        for cname in ("a", "v"):
            what = ("at", cname, len(sel))
            r = ops.rfx_at(vecs[cname], idv); calls += 1
            if H.is_error(r):
                handed += 1
            else:
                got, want = H.to_numpy(r), cols[cname][sel]
                assert np.array_equal(got.view(np.int64), want.view(np.int64)), what
            ops.rfx_host_drop(r)
        ops.rfx_host_drop(hm[0]); ops.rfx_host_drop(hm[1]); ops.rfx_host_drop(idv)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "n", n, what, "->", repr(e)[:300], flush=True)
    for o in vecs.values():
        ops.rfx_host_drop(o)
print("done", hi - lo, "seeds,", calls, "calls,", handed, "handed back,", bad, "failures")
