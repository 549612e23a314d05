"""Operator-level C ABI (include/rfx_ops.h) driven through obj_p arguments laid out like RayforceDB objects, by the
standalone host object model (no reference process).  Checker: the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import rfo
from rayforce_amd import hostobj as H

pytestmark = pytest.mark.gpu
NULL = -(2**63)


@pytest.fixture(scope="module")
def ops(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    l = H.lib()
    assert l.rfx_host_bind() == 0  # standalone host inside the python process
    yield l
    l.rfx_cache_clear()


@pytest.fixture
def checksum_mode(ops):
    """Validation by checksum (rfx_ops_set_validation(1)): for the tests below that write a cached payload IN PLACE through a raw numpy view,
    whatever its reference count -- what the reference itself never does (cow_obj, core/rayforce.c:3003-3026).  The default, validation by
    ownership, is back afterwards."""
    assert ops.rfx_ops_set_validation(1) == 0
    yield
    assert ops.rfx_ops_set_validation(0) == 0


def host_table(n, keys=1000, seed=0):
    return {"k": rfo.gen_i64(n, 4 + seed, keys), "a": rfo.gen_i64(n, 2 + seed, 1_000_000), "v": rfo.gen_f64(n, 5 + seed)}


def run_select(ops, host, query):
    tab = H.table(host)
    d = H.select_dict(query, tab)
    r = ops.rfx_select(d)
    assert r, "null result"
    if H.is_error(r):
        msg = H.error_text(r)
        ops.rfx_host_drop(r)
        ops.rfx_host_drop(d)
        ops.rfx_host_drop(tab)
        raise RuntimeError(msg)
    out = H.table_to_numpy(r)
    for o in (r, d, tab):
        ops.rfx_host_drop(o)
    return out


def check(got, want):
    assert list(got) == list(want)
    for name in want:
        g, w = got[name], want[name]
        assert g.dtype == w.dtype, name
        if w.dtype == np.float64:
            assert np.array_equal(np.isnan(g), np.isnan(w))
            ok = ~np.isnan(w)
            assert np.allclose(g[ok], w[ok], rtol=1e-9, atol=0), name
        else:
            assert np.array_equal(g, w), name


@pytest.mark.parametrize("n", [1000, 300_007])
def test_select_where_aggregates(ops, n):
    host = host_table(n)
    q = {"s": ("sum", "a"), "f": ("sum", "v"), "c": ("count", "v"), "m": ("max", "a"), "x": ("avg", "v")}
    got = run_select(ops, host, {**q, "where": ("<", "a", 100_000)})
    assert ops.rfx_last_select_on_gpu() == 1
    check(got, rfo.select({"from": host, **q, "where": ("<", "a", 100_000)}))
    w3 = ("and", ("<", "a", 500_000), (">", "v", 0.25), ("!=", "k", 7))
    check(run_select(ops, host, {**q, "where": w3}), rfo.select({"from": host, **q, "where": w3}))


@pytest.mark.parametrize("keys", [10, 1000, 250_000])
def test_select_by(ops, keys):
    host = host_table(400_003, keys=keys)
    q = {"s": ("sum", "v"), "c": ("count", "a"), "mn": ("min", "a"), "av": ("avg", "v")}
    check(run_select(ops, host, {**q, "by": "k"}), rfo.select({"from": host, **q, "by": "k"}))
    check(run_select(ops, host, {**q, "by": "k", "where": (">", "v", 0.5)}), rfo.select({"from": host, **q, "by": "k", "where": (">", "v", 0.5)}))


def test_select_by_sparse_keys(ops):
    host = host_table(50_000, keys=500)
    host["k"] = host["k"] * 999_983 + 5
    q = {"s": ("sum", "v"), "c": ("count", "a")}
    check(run_select(ops, host, {**q, "by": "k"}), rfo.select({"from": host, **q, "by": "k"}))


def test_select_by_several_keys(ops):
    """by: {name: column ...} -- composite key (index_group_list_perfect); a one-entry dict renames the key column."""
    n = 300_007
    host = {"k1": rfo.gen_i64(n, 31, 100), "k2": rfo.gen_i64(n, 41, 90) + 10**9, "k3": rfo.gen_i64(n, 51, 3) - 1, "v": rfo.gen_f64(n, 36),
            "a": rfo.gen_i64(n, 37, 1_000_000)}
    q = {"s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", "a")}
    for by in ({"id1": "k1", "id2": "k2"}, {"z": "k3", "y": "k2", "x": "k1"}, {"only": "k2"}):
        check(run_select(ops, host, {**q, "by": by}), rfo.select({"from": host, **q, "by": by}))
        assert ops.rfx_last_select_on_gpu() == 1
    # where: + several keys is left to the host (the reference's own answer for it is defective); no host here -> loud error
    with pytest.raises(RuntimeError, match="several by: columns"):
        run_select(ops, host, {**q, "by": {"a1": "k1", "a2": "k2"}, "where": ("<", "a", 5)})
    # product of ranges beyond i64 (a null key always is) -> the reference's row-hash path: answered here too, key columns
    # and first-occurrence group order included
    host["k2"][3] = NULL
    host["k1"][::1013] = NULL
    ops.rfx_cache_clear()  # a rebuilt column may land on the freed one's address; one changed cell can escape the sampled checksum
    check(run_select(ops, host, {**q, "by": {"a1": "k1", "a2": "k2"}}), rfo.select({"from": host, **q, "by": {"a1": "k1", "a2": "k2"}}))
    assert ops.rfx_last_select_on_gpu() == 1
    wide = dict(host)
    wide["k1"] = rfo.gen_i64(n, 31, 100) * (1 << 50)
    wide["k2"] = rfo.gen_i64(n, 41, 90) * (1 << 44) - (1 << 52)
    ops.rfx_cache_clear()
    by3 = {"x": "k1", "y": "k2", "z": "k3"}
    check(run_select(ops, wide, {"s": ("sum", "v"), "c": ("count", "a"), "by": by3}), rfo.select({"from": wide, "s": ("sum", "v"), "c": ("count", "a"), "by": by3}))
    assert ops.rfx_last_select_on_gpu() == 1
    # (round 1 proved a group's tuple with a (min, max) aggregate pair per key column and ran out of aggregate slots here: the proof is a
    #  probe of the group-by's own table + one gather / compare per key column now, the outputs have the launch to themselves)
    check(run_select(ops, wide, {**q, "by": by3}), rfo.select({"from": wide, **q, "by": by3}))
    assert ops.rfx_last_select_on_gpu() == 1


def test_select_expression_aggregates(ops):
    """(sum (* a v)) & co through the operator ABI: scalar, grouped (LDS tables and partitioned), flat and nested where."""
    host = host_table(300_007, keys=4000)
    host["b"] = rfo.gen_i64(300_007, 9, 9) - 1
    q = {"s1": ("sum", ("*", "a", "v")), "s2": ("sum", ("*", "a", "b")), "av": ("avg", ("-", "a", "b")), "mn": ("min", ("-", 100, "a")),
         "s4": ("sum", ("div", "a", "b")), "mx": ("max", ("*", "v", 2.5)), "plain": ("sum", "v")}
    nested = ("and", ("or", ("<", "a", 1000), (">", "v", 0.9)), ("!=", "k", 3))
    for extra in ({}, {"where": ("<", "b", 5)}, {"where": nested}, {"by": "k"}, {"by": "k", "where": ("<", "b", 5)}, {"by": "k", "where": nested}):
        check(run_select(ops, host, {**q, **extra}), rfo.select({"from": host, **q, **extra}))
        assert ops.rfx_last_select_on_gpu() == 1
    host2 = host_table(300_007, keys=200_000)
    q2 = {"s": ("sum", ("*", "a", "v")), "m": ("max", ("+", "v", "a"))}
    check(run_select(ops, host2, {**q2, "by": "k"}), rfo.select({"from": host2, **q2, "by": "k"}))
    with pytest.raises(RuntimeError, match="not covered by the MI355X path"):
        run_select(ops, host, {"c": ("count", ("*", "a", "v"))})
    # nested expressions (TPC-H Q1 shape): up to four operations, folded on the fly
    q3 = {"dp": ("sum", ("*", "v", ("-", 1, "v"))), "ch": ("sum", ("*", ("*", "v", ("-", 1.0, "v")), ("+", 1, "b"))),
          "mx": ("max", ("div", ("*", "v", "a"), ("+", "b", 2))), "plain": ("avg", "a")}
    for extra in ({}, {"where": ("<", "b", 5)}, {"where": nested}, {"by": "k"}, {"by": "k", "where": nested}):
        check(run_select(ops, host, {**q3, **extra}), rfo.select({"from": host, **q3, **extra}))
        assert ops.rfx_last_select_on_gpu() == 1
    with pytest.raises(RuntimeError, match="deeper than"):
        run_select(ops, host, {"s": ("sum", ("+", ("+", ("+", ("+", ("+", "a", 1), 1), 1), 1), 1))})


def test_select_by_xbar(ops):
    host = host_table(300_007, keys=4000)
    q = {"s": ("sum", "v"), "c": ("count", "a")}
    for by in ({"b": ("xbar", "a", 1000)}, {"k": "k", "b": ("xbar", "a", 250_000)}):
        check(run_select(ops, host, {**q, "by": by}), rfo.select({"from": host, **q, "by": by}))
        assert ops.rfx_last_select_on_gpu() == 1
    check(run_select(ops, host, {**q, "by": {"b": ("xbar", "a", 1000)}, "where": (">", "v", 0.5)}),
          rfo.select({"from": host, **q, "by": {"b": ("xbar", "a", 1000)}, "where": (">", "v", 0.5)}))


def test_nested_tree_and_projection(ops):
    host = host_table(200_003)
    host["b"] = rfo.gen_i64(200_003, 77, 9) - 1
    nested = ("and", ("or", ("<", "a", 1000), (">", "v", 0.9)), ("!=", "k", 3))
    q = {"s": ("sum", "a"), "f": ("sum", "v"), "c": ("count", "a"), "fi": ("first", "a")}
    m0 = H.to_numpy(ops.rfx_stats(0))[10]
    check(run_select(ops, host, {**q, "where": nested}), rfo.select({"from": host, **q, "where": nested}))
    assert ops.rfx_last_select_on_gpu() == 1
    check(run_select(ops, host, {**q, "where": nested, "by": "k"}), rfo.select({"from": host, **q, "where": nested, "by": "k"}))
    q19 = ("or", ("==", ("div", "a", 1000), 7), ("and", (">", ("*", "v", 2.0), 1.5), ("!=", "k", 3)), ("and", ("<", "b", 2), ("<", "a", 50_000), (">=", "v", 0.5)))
    for extra in ({}, {"by": "k"}, {"by": {"t": ("xbar", "a", 1000)}}):  # (where: with several by: columns is handed to the host: reference defect)
        check(run_select(ops, host, {**q, "where": q19, **extra}), rfo.select({"from": host, **q, "where": q19, **extra}))
    assert H.to_numpy(ops.rfx_stats(0))[10] == m0, "two-level trees: one fused pass, no materialised comparison masks"
    deep = ("or", ("and", ("<", "a", 300_000), (">", "v", 0.2)), ("and", ("==", "b", 3), ("or", ("<", "v", 0.3), (">=", "a", 900_000))))
    check(run_select(ops, host, {**q, "where": deep}), rfo.select({"from": host, **q, "where": deep}))
    assert H.to_numpy(ops.rfx_stats(0))[10] == m0  # (round 4: three and four levels run in the fused pass too -- rfx_pred_t's tree form)
    wide = ("or", *[("and", ("<", "a", 100_000 * (i + 1)), (">", "v", 0.1 * i)) for i in range(5)])  # ten comparisons: more than one pass carries
    check(run_select(ops, host, {**q, "where": wide}), rfo.select({"from": host, **q, "where": wide}))
    check(run_select(ops, host, {**q, "where": wide, "by": "k"}), rfo.select({"from": host, **q, "where": wide, "by": "k"}))
    assert H.to_numpy(ops.rfx_stats(0))[10] > m0  # ... the reference's own plan -- masks, where -- on the device
    # projection = filter_collect of every column (core/filter.c:51-165), flat and nested predicates
    check(run_select(ops, host, {"where": ("<", "a", 1000)}), rfo.select({"from": host, "where": ("<", "a", 1000)}))
    check(run_select(ops, host, {"where": nested}), rfo.select({"from": host, "where": nested}))
    check(run_select(ops, host, {"where": ("<", "a", -1)}), rfo.select({"from": host, "where": ("<", "a", -1)}))


def test_unsupported_shape_fails_loudly_without_host(ops):
    host = host_table(100)
    with pytest.raises(RuntimeError, match="not covered by the MI355X path"):
        run_select(ops, host, {"s": ("sum", "a"), "by": {"x": "v", "y": "k"}})  # an f64 column among several keys: the reference host would take it
    assert ops.rfx_last_select_on_gpu() == 0


def test_missing_from(ops):
    d = ops.rfx_host_dict(H.symbols(["s"]), H.list_of([H.expr(("sum", "v"))]))
    r = ops.rfx_select(d)
    assert H.is_error(r) and "from" in H.error_text(r)
    ops.rfx_host_drop(r)
    ops.rfx_host_drop(d)


def test_single_operators(ops):
    n = 100_003
    host = host_table(n)
    a, v = H.vector(host["a"]), H.vector(host["v"])
    m1 = ops.rfx_lt(a, H.atom(500_000))
    m2 = ops.rfx_gt(v, H.atom(0.25))
    assert np.array_equal(H.to_numpy(m1), rfo.cmp("<", host["a"], 500_000))
    both = ops.rfx_and((C.c_void_p * 2)(m1, m2), 2)
    want = rfo.and_(rfo.cmp("<", host["a"], 500_000), rfo.cmp(">", host["v"], 0.25))
    assert np.array_equal(H.to_numpy(both), want)
    ids = ops.rfx_where(both)
    assert np.array_equal(H.to_numpy(ids), rfo.where(want))
    g = ops.rfx_at(v, ids)
    assert np.array_equal(H.to_numpy(g), host["v"][rfo.where(want)])
    s = ops.rfx_sum(a)
    assert H.header(s).type == -H.T_I64 and C.c_int64.from_address(s + 8).value == int(host["a"].sum())
    mx = ops.rfx_max(v)
    assert H.header(mx).type == -H.T_F64 and C.c_double.from_address(mx + 8).value == host["v"].max()
    for o in (a, v, m1, m2, both, ids, g, s, mx):
        ops.rfx_host_drop(o)


def test_select_predicates_over_expressions(ops):
    host = host_table(200_003, keys=900)
    host["b"] = rfo.gen_i64(200_003, 9, 9) - 1
    q = {"s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", ("*", "a", "v"))}
    for w in ((">", ("*", "a", "v"), 250_000.0), ("<=", ("-", "a", ("*", "b", 100_000)), "a"), ("and", ("<", ("+", "v", "v"), 0.6), (">", "a", 1000)),
              ("or", ("==", ("div", "a", 1000), 7), ("and", (">", ("*", "v", 2.0), 1.5), ("!=", "b", 3))), ("<", "v", ("*", "v", 3.0))):
        for extra in ({}, {"by": "k"}):
            check(run_select(ops, host, {**q, "where": w, **extra}), rfo.select({"from": host, **q, "where": w, **extra}))
            assert ops.rfx_last_select_on_gpu() == 1


def test_join_operators(ops):
    """rfx_left_join / rfx_inner_join: vary_f over (key symbols, left table, right table) with the reference's column rules; one
    key (dense and hashed tables), two keys (composite key), wide / null key tuples (row hash + column-by-column check)."""
    import golden_cases as G
    for case in G.join_cases():
        name, keys, left, right, want_lj, want_ij = case
        lt, rt, ks = H.table(left), H.table(right), H.symbols(keys)
        args = (C.c_void_p * 3)(ks, lt, rt)
        for fn, ora, want in (("rfx_left_join", rfo.left_join, want_lj), ("rfx_inner_join", rfo.inner_join, want_ij)):
            out = getattr(ops, fn)(args, 3)
            assert not H.is_error(out), (name, fn, H.error_text(out))
            got, o = H.table_to_numpy(out), ora(keys, left, right)
            assert list(got) == list(o), (name, fn)
            for c in o:
                assert got[c].dtype == o[c].dtype and np.array_equal(got[c].view(np.int64), o[c].view(np.int64)), (name, fn, c)  # bit-exact, NaN fill included
            for c in want:  # the reference's own answers where it gives typed columns
                G.same(got[c], want[c], f"{name} {fn} {c}")
            ops.rfx_host_drop(out)
        for o in (lt, rt, ks):
            ops.rfx_host_drop(o)
    # empty right side -> the left table; a non-table argument -> error object
    lt, rt, ks = H.table({"k": np.arange(5, dtype=np.int64), "v": np.arange(5, dtype=np.float64)}), H.table({"k": np.empty(0, np.int64), "w": np.empty(0, np.float64)}), H.symbols(["k"])
    out = ops.rfx_left_join((C.c_void_p * 3)(ks, lt, rt), 3)
    assert list(H.table_to_numpy(out)) == ["k", "v"]
    bad = ops.rfx_inner_join((C.c_void_p * 3)(ks, ks, rt), 3)
    assert H.is_error(bad)
    for o in (out, bad, lt, rt, ks):
        ops.rfx_host_drop(o)


def test_arithmetic_operators(ops):
    """rfx_add / sub / mul / div: binary_f over an i64 / f64 vector and a vector or atom (either order), the reference's promotion
    and null rules (oracle binop, pinned on the reference's 48 truth tables)."""
    n = 100_003
    host = host_table(n)
    host["a"][::97] = NULL
    host["b"] = rfo.gen_i64(n, 9, 9) - 1
    objs = {c: H.vector(host[c]) for c in ("a", "b", "v")}
    for name, op in (("add", "+"), ("sub", "-"), ("mul", "*"), ("div", "div")):
        for l, r in (("a", "b"), ("a", "v"), ("v", "a"), ("v", "v"), ("a", 7), (7, "a"), ("v", 2.5), (-1.5, "v"), ("a", 2.5), (3, "v")):
            lo = objs[l] if isinstance(l, str) else H.atom(l)
            ro = objs[r] if isinstance(r, str) else H.atom(r)
            out = getattr(ops, f"rfx_{name}")(lo, ro)
            assert not H.is_error(out), (name, l, r, H.error_text(out))
            got = H.to_numpy(out)
            want = rfo.binop(op, host[l] if isinstance(l, str) else l, host[r] if isinstance(r, str) else r)
            assert got.dtype == want.dtype, (name, l, r)
            if want.dtype == np.float64:
                fin = np.isfinite(want)
                assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)]), (name, l, r)
                assert np.all(np.abs(got[fin] - want[fin]) <= 2.3e-16 * np.abs(want[fin])), (name, l, r)
            else:
                assert np.array_equal(got, want), (name, l, r)
            ops.rfx_host_drop(out)
            for o, x in ((lo, l), (ro, r)):
                if not isinstance(x, str):
                    ops.rfx_host_drop(o)
    bad = ops.rfx_add(H.atom(1), H.atom(2))  # no vector: not this path's business, and no host to hand it to here
    assert H.is_error(bad)
    for o in list(objs.values()) + [bad]:
        ops.rfx_host_drop(o)


def test_lazy_mapfilter_aggregates(ops):
    """An FN_AGGR built-in receives the lazy TYPE_MAPFILTER pair (val, ids) (core/eval.c:723-728, core/filter.c:44-46): gathered and
    folded on the device."""
    n = 200_003
    host = host_table(n)
    host["a"][::97] = NULL
    ids = rfo.where(rfo.cmp("<", host["a"], 400_000))
    for col, fns in (("a", ("sum", "min", "max", "avg", "count")), ("v", ("sum", "min", "max", "avg"))):
        val, idv = H.vector(host[col]), H.vector(ids)
        pair = H.list_of([val, idv])
        H.header(pair).type = 71  # TYPE_MAPFILTER
        for fn in fns:
            r = getattr(ops, f"rfx_{fn}")(pair)
            assert not H.is_error(r), H.error_text(r)
            want = rfo.fold(fn, host[col][ids])
            got = C.c_double.from_address(r + 8).value if H.header(r).type == -H.T_F64 else C.c_int64.from_address(r + 8).value
            if isinstance(want, float):
                assert abs(got - want) <= 1e-9 * abs(want), (col, fn)
            else:
                assert got == (len(ids) if fn == "count" else want), (col, fn, got, want)
            ops.rfx_host_drop(r)
        ops.rfx_host_drop(pair)


def test_residency_cache(ops):
    ops.rfx_cache_clear()
    host = host_table(200_000)
    tab = H.table(host)
    p = ops.rfx_pin(tab)
    assert ops.rfx_cache_bytes() == 3 * 200_000 * 8
    u = ops.rfx_unpin(tab)
    assert ops.rfx_cache_bytes() == 0
    for o in (p, u, tab):
        ops.rfx_host_drop(o)


def test_residency_by_ownership(ops):
    """Round 6, the default: a cached copy holds a reference to its host vector (clone_obj); a later use of the SAME object is a pointer compare
    (rfx_stats[13]), never a checksum (rfx_stats[12]); a host that follows the reference's rule -- write in place only with rc == 1, else copy
    (cow_obj, core/rayforce.c:3003-3026) -- cannot be served a stale cell; a vector the host has dropped is released at the next operator call."""
    ops.rfx_cache_clear()
    n = 1_000_003
    a = rfo.gen_i64(n, 3, 1000)
    vec = H.vector(a)
    assert H.header(vec).rc == 1
    s0 = H.to_numpy(ops.rfx_stats(0))
    assert _sum_of(ops, vec) == int(a.sum())
    assert H.header(vec).rc == 2  # the cache's reference: from now on an owner that looks at the count copies before it writes
    for _ in range(5):
        assert _sum_of(ops, vec) == int(a.sum())
    s1 = H.to_numpy(ops.rfx_stats(0))
    assert s1[4] - s0[4] == 1 and s1[13] - s0[13] == 5 and s1[12] == s0[12] and s1[6] == s0[6]  # one upload, five pointer compares, no checksum

    def host_write(v, j, val):
        """what every writer of the reference does: cow_obj -- the object itself with rc == 1, a copy otherwise; the old reference is dropped"""
        if H.header(v).rc == 1:
            w = v
        else:
            w = H.vector(H.to_numpy(v))
            ops.rfx_host_drop(v)
        np.frombuffer((C.c_char * (n * 8)).from_address(H.payload(w)), dtype=np.int64)[j] = val
        return w

    total = int(a.sum())
    rng = np.random.default_rng(11)
    for i in range(40):
        j = int(rng.integers(0, n))
        old = int(H.to_numpy(vec)[j])
        new = int(rng.integers(0, 1 << 40))
        was = vec
        vec = host_write(vec, j, new)
        assert (vec == was) == (i % 2 == 0 and i > 0)  # (every other write finds rc == 1 -- the vector was not used in between -- and goes in place)
        total += new - old
        if i % 2 == 0:
            assert _sum_of(ops, vec) == total, i  # the copy is a new object -> a new upload; the old one is released at this call
            assert H.header(vec).rc == 2
    assert _sum_of(ops, vec) == total
    s2 = H.to_numpy(ops.rfx_stats(0))
    assert s2[12] == s0[12] and s2[6] == s0[6] and s2[14] - s1[14] >= 19  # still no checksum; the replaced vectors were let go
    assert ops.rfx_cache_bytes() == n * 8  # ... and their device copies with them
    # pinned or not: a dropped vector is released; rfx_invalidate / rfx_unpin give the reference back
    p = ops.rfx_pin(vec)
    assert H.header(vec).rc == 3  # (ours, the cache's, the clone rfx_pin returned)
    ops.rfx_host_drop(p)
    iv = ops.rfx_invalidate(vec)
    ops.rfx_host_drop(iv)
    assert H.header(vec).rc == 1 and ops.rfx_cache_bytes() == 0
    p = ops.rfx_pin(vec)
    ops.rfx_host_drop(p)
    ops.rfx_host_drop(vec)  # the host lets go of a pinned vector: nobody can name it any more
    other = H.vector(a[:1000].copy())
    assert _sum_of(ops, other) == int(a[:1000].sum())
    assert ops.rfx_cache_bytes() == 1000 * 8
    ops.rfx_host_drop(other)
    # tables: every column the query names is held; dropping the table releases them all
    host = host_table(100_003)
    tab = H.table(host)
    cols = H.list_items(H.list_items(tab)[1])
    d = H.select_dict({"s": ("sum", "v"), "where": ("<", "a", 500_000), "by": "k"}, tab)
    for _ in range(3):
        r = ops.rfx_select(d)
        assert not H.is_error(r), H.error_text(r)
        check(H.table_to_numpy(r), rfo.select({"from": host, "s": ("sum", "v"), "where": ("<", "a", 500_000), "by": "k"}))
        ops.rfx_host_drop(r)
    assert [H.header(c).rc for c in cols] == [2, 2, 2]
    ops.rfx_host_drop(d)
    ops.rfx_host_drop(tab)
    s3 = H.to_numpy(ops.rfx_stats(0))  # (an operator call like any other: what only the cache still refers to goes)
    assert ops.rfx_cache_bytes() == 0 and s3[12] == s0[12]
    assert ops.rfx_ops_set_validation(7) != 0


def _sum_of(ops, vec):
    s = ops.rfx_sum(vec)
    assert not H.is_error(s), H.error_text(s)
    got = C.c_int64.from_address(s + 8).value
    ops.rfx_host_drop(s)
    return got


def test_cache_never_serves_a_stale_cell(ops, checksum_mode):
    """An unpinned cached column is re-validated against a checksum of its FULL payload on every use: flip ONE random cell of a
    cached 1e7-row column in place, 1 000 times -- the answer follows every time (round 1 sampled 64 cells and could miss it)."""
    ops.rfx_cache_clear()
    n = 10_000_000
    a = rfo.gen_i64(n, 2, 1_000_000)
    vec = H.vector(a)
    view = np.frombuffer((C.c_char * (n * 8)).from_address(H.payload(vec)), dtype=np.int64)
    total = int(a.sum())
    assert _sum_of(ops, vec) == total
    rng = np.random.default_rng(7)
    before = H.to_numpy(ops.rfx_stats(0))
    for i in range(1000):
        j = int(rng.integers(0, n))
        delta = int(rng.integers(1, 1 << 40))
        view[j] += delta  # in place: same address, same length, same type
        total += delta
        assert _sum_of(ops, vec) == total, (i, j)
    after = H.to_numpy(ops.rfx_stats(0))
    assert after[6] - before[6] == 1000  # every change was noticed (stale entries refreshed) ...
    assert _sum_of(ops, vec) == total and H.to_numpy(ops.rfx_stats(0))[5] == after[5] + 1  # ... and an unchanged column is a cache hit
    ops.rfx_host_drop(vec)


def test_cache_sees_changes_in_the_top_bit_alone(ops, checksum_mode):
    """Cells that change in bit 63 only -- 0 <-> NULL_I64, an f64 <-> its negative -- in an even number of places: the checksum's lanes must
    carry high bits downward (a bare multiply-xor remembered such changes by their parity per lane, and a column of nulls written over a
    column of zeros of the same length at the same address was served from the stale copy: found by tools/fuzz_null_tuples.py, round 5)."""
    ops.rfx_cache_clear()
    for n in (8, 4096, 1 << 20, (1 << 21) + 8):  # (the last: checksummed by several threads)
        v = np.ones(n, np.float64)
        vec = H.vector(v)
        view = np.frombuffer((C.c_char * (n * 8)).from_address(H.payload(vec)), dtype=np.uint64)
        s = ops.rfx_sum(vec)
        assert C.c_double.from_address(s + 8).value == float(n)
        ops.rfx_host_drop(s)
        before = H.to_numpy(ops.rfx_stats(0))[6]
        view ^= np.uint64(1 << 63)  # every cell negated, in place
        s = ops.rfx_sum(vec)
        assert C.c_double.from_address(s + 8).value == -float(n), n
        ops.rfx_host_drop(s)
        view[8 % n] ^= np.uint64(1 << 63)  # ... and two cells of one lane
        view[(8 % n + 4) % n] ^= np.uint64(1 << 63)
        s = ops.rfx_sum(vec)
        assert C.c_double.from_address(s + 8).value == -float(n) + 4.0, n
        ops.rfx_host_drop(s)
        assert H.to_numpy(ops.rfx_stats(0))[6] - before == 2
        ops.rfx_host_drop(vec)
        # zeros -> nulls through a group-by key: one group either way, the KEY differs
        kv, vv = H.vector(np.zeros(n, np.int64)), H.vector(np.ones(n, np.float64))
        tab = ops.rfx_host_table(H.symbols(["k", "v"]), H.list_of([kv, vv]))
        d = H.select_dict({"c": ("count", "v"), "by": {"g": "k", "h": "k"}}, tab)
        r = ops.rfx_select(d)
        assert not H.is_error(r), H.error_text(r)
        got = H.table_to_numpy(r)
        ops.rfx_host_drop(r)
        assert got["g"].tolist() == [0] and got["c"].tolist() == [n]
        kview = np.frombuffer((C.c_char * (n * 8)).from_address(H.payload(kv)), dtype=np.uint64)
        kview ^= np.uint64(1 << 63)  # every key 0 -> NULL_I64
        r = ops.rfx_select(d)
        assert not H.is_error(r), H.error_text(r)
        got = H.table_to_numpy(r)
        ops.rfx_host_drop(r)
        assert got["g"].tolist() == [NULL] and got["h"].tolist() == [NULL] and got["c"].tolist() == [n], n
        ops.rfx_host_drop(d)
        ops.rfx_host_drop(tab)


def test_cache_validates_by_page_bits_where_the_kernel_tracks_them(ops, checksum_mode):
    """Round 3: where the kernel tracks soft-dirty pages (the MI355X boxes' does; probed at run time) an unchanged unpinned column is proven
    current by its pages' bits -- rfx_stats[11] counts those uses -- and a write anywhere in it (first page, last page, the middle) is still
    noticed at once; a second column uploaded in between (its clear_refs wipes the first column's evidence) must not hide a write that
    happened before it."""
    ops.rfx_cache_clear()
    n = 3_000_017
    a, b = rfo.gen_i64(n, 5, 1_000_000), rfo.gen_i64(n, 6, 1_000_000)
    va, vb = H.vector(a), H.vector(b)
    wa = np.frombuffer((C.c_char * (n * 8)).from_address(H.payload(va)), dtype=np.int64)
    total = int(a.sum())
    for _ in range(4):  # upload, two unchanged uses by checksum (the column proves stable), the use that starts the tracking
        assert _sum_of(ops, va) == total
    s0 = H.to_numpy(ops.rfx_stats(0))
    for _ in range(5):
        assert _sum_of(ops, va) == total
    s1 = H.to_numpy(ops.rfx_stats(0))
    assert s1[5] - s0[5] == 5 and s1[6] == s0[6]  # five cache hits either way
    tracked = s1[11] - s0[11]
    if os.environ.get("RFX_SOFT_DIRTY") == "1":
        assert tracked in (0, 5)  # opted in: all by page bits, or (no kernel support) all by checksum
    else:
        assert tracked == 0  # round 4: page tracking is OPT-IN (clear_refs write-protects every page of the host process)
    for j in (0, 1, n // 2, n - 2, n - 1, 511, 512):  # first / last partial page, whole pages
        wa[j] += 7
        total += 7
        assert _sum_of(ops, va) == total, j
        assert _sum_of(ops, va) == total
    # `a` tracked again; write to it, THEN let `b` become tracked (its clear wipes every page's bit), THEN ask for `a`: the write must not be lost
    for _ in range(4):
        assert _sum_of(ops, va) == total
    wa[n // 3] += 11
    total += 11
    for _ in range(5):
        assert _sum_of(ops, vb) == int(b.sum())
    assert _sum_of(ops, va) == total
    assert _sum_of(ops, vb) == int(b.sum()) and _sum_of(ops, va) == total
    s2 = H.to_numpy(ops.rfx_stats(0))
    assert s2[6] - s1[6] == 8  # every change refreshed the copy exactly once
    for o in (va, vb):
        ops.rfx_host_drop(o)


def test_page_bit_validation_when_opted_in(built):
    """The same test in a process started with RFX_SOFT_DIRTY=1: the soft-dirty path itself (clear -> checksum of every tracked column -> trust)."""
    import subprocess, sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, RFX_SOFT_DIRTY="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", "test_cache_validates_by_page_bits"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "1 passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_remembered_key_scope_follows_the_column(ops, checksum_mode):
    """Small inputs: the whole-column scope of a resident key column is remembered with its device copy (one host round trip less per
    group-by); a write into the column -- a key outside the old scope -- refreshes the copy and forgets the scope; a filter that selects
    nothing comes out as zero groups; a pinned column keeps copy and scope until rfx_invalidate."""
    ops.rfx_cache_clear()
    n = 60_007
    host = host_table(n, keys=100)
    tab = H.table(host)
    kvec = H.list_items(H.list_items(tab)[1])[0]
    kview = np.frombuffer((C.c_char * (n * 8)).from_address(H.payload(kvec)), dtype=np.int64)

    def ask(q):
        d = H.select_dict(q, tab)
        r = ops.rfx_select(d)
        assert r and not H.is_error(r), H.error_text(r)
        out = H.table_to_numpy(r)
        ops.rfx_host_drop(r)
        ops.rfx_host_drop(d)
        return out

    q = {"s": ("sum", "v"), "c": ("count", "a"), "by": "k"}
    qf = {"s": ("sum", "v"), "mx": ("max", "a"), "where": ("<", "a", 300_000), "by": "k"}
    for _ in range(3):  # the second and third ask find the scope remembered
        check(ask(q), rfo.select({"from": host, **q}))
        check(ask(qf), rfo.select({"from": host, **qf}))
    none = ask({"s": ("sum", "v"), "where": ("<", "a", -5), "by": "k"})
    assert len(none["k"]) == 0 and len(none["s"]) == 0
    kview[12_345] = 5_000  # far outside [0, 100): the old scope would drop (or misplace) this row
    host["k"] = kview.copy()
    check(ask(q), rfo.select({"from": host, **q}))
    check(ask(qf), rfo.select({"from": host, **qf}))
    kview[77] = -40  # ... and below it
    host["k"] = kview.copy()
    check(ask(q), rfo.select({"from": host, **q}))
    # pinned: trusted (copy and scope) until invalidated
    p = ops.rfx_pin(tab)
    check(ask(q), rfo.select({"from": host, **q}))
    kview[5] = 9_000
    stale = ask(q)
    assert 9_000 not in stale["k"]  # (documented contract of rfx_pin)
    iv = ops.rfx_invalidate(tab)
    host["k"] = kview.copy()
    check(ask(q), rfo.select({"from": host, **q}))
    for o in (p, iv, tab):
        ops.rfx_host_drop(o)


def test_cache_pin_trusts_until_invalidated(ops, checksum_mode):
    """rfx_pin: no per-use validation (the host promises rfx_invalidate before it writes); rfx_invalidate drops the copy."""
    ops.rfx_cache_clear()
    a = rfo.gen_i64(1_000_003, 3, 1000)
    vec = H.vector(a)
    view = np.frombuffer((C.c_char * (a.size * 8)).from_address(H.payload(vec)), dtype=np.int64)
    p = ops.rfx_pin(vec)
    assert _sum_of(ops, vec) == int(a.sum())
    view[12345] += 5
    assert _sum_of(ops, vec) == int(a.sum())  # pinned: the device copy is trusted (documented contract)
    iv = ops.rfx_invalidate(vec)
    assert ops.rfx_cache_bytes() == 0
    assert _sum_of(ops, vec) == int(a.sum()) + 5
    for o in (p, iv, vec):
        ops.rfx_host_drop(o)


def test_temporaries_at_recycled_addresses(ops):
    """Masks / id vectors are temporaries: freed by the host right after the call, the next one usually lands at the same address.
    (where (== c 5)) followed by (where (== c 7)) must not answer the first query's ids."""
    n = 300_007
    c = rfo.gen_i64(n, 11, 1000)
    col = H.vector(c)
    for k in (5, 7, 5, 900, 7):
        m = ops.rfx_eq(col, H.atom(k))
        ids = ops.rfx_where(m)
        assert np.array_equal(H.to_numpy(ids), np.nonzero(c == k)[0]), k
        g = ops.rfx_at(col, ids)
        assert np.array_equal(H.to_numpy(g), np.full(int((c == k).sum()), k, np.int64))
        for o in (g, ids, m):
            ops.rfx_host_drop(o)  # the standalone host frees: the next mask reuses the block
    ops.rfx_host_drop(col)


def test_at_out_of_range_ids_read_null(ops):
    """at_vec_i64_by_i64 / at_vec_f64_by_i64 (core/items.c:53-72): idx < 0 (null included) or >= len -> typed null, never a read
    beyond the column."""
    a = np.arange(1000, dtype=np.int64) * 3
    v = np.arange(1000, dtype=np.float64) / 7
    ids = np.array([0, 999, 1000, -1, NULL, 5, 2**40, 17], np.int64)
    av, vv, iv = H.vector(a), H.vector(v), H.vector(ids)
    ga, gv = ops.rfx_at(av, iv), ops.rfx_at(vv, iv)
    ok = (ids >= 0) & (ids < 1000)
    want_a = np.where(ok, a[np.where(ok, ids, 0)], NULL)
    got_v = H.to_numpy(gv)
    assert np.array_equal(H.to_numpy(ga), want_a)
    assert np.array_equal(np.isnan(got_v), ~ok) and np.array_equal(got_v[ok], v[ids[ok]])
    # the lazy MAPFILTER pair takes host ids too
    pair = H.list_of([av, iv])
    H.header(pair).type = 71
    s = ops.rfx_sum(pair)
    assert C.c_int64.from_address(s + 8).value == int(a[ids[ok]].sum())  # null entries are skipped by the scalar sum
    for o in (s, pair, ga, gv):
        ops.rfx_host_drop(o)


def test_cache_budget_never_evicts_what_the_call_reads(ops, monkeypatch):
    """A budget smaller than one query's columns: entries touched by the call in flight are not evictable (the descriptors hold
    their device pointers), the cache goes over budget instead."""
    ops.rfx_cache_clear()
    host = host_table(400_003, keys=1000)
    monkeypatch.setenv("RFX_CACHE_BYTES", str(400_003 * 8 + 100))  # room for ONE column
    try:
        q = {"s": ("sum", "v"), "mx": ("max", "a"), "by": "k", "where": ("<", "a", 700_000)}
        for _ in range(3):
            check(run_select(ops, host, q), rfo.select({"from": host, **q}))
    finally:
        monkeypatch.delenv("RFX_CACHE_BYTES")
        ops.rfx_cache_clear()


def test_null_group_keys_are_handed_back(ops):
    """A selected null key: the reference opens one group per null-key row (NULL_I64 is the empty marker of its table,
    core/index.c:1808-1816).  rfx_select does not answer differently at the boundary: it hands the query to the host -- here, with no
    host behind it, that is a loud error naming the reason."""
    host = host_table(50_000, keys=700)
    host["k"][::997] = NULL
    with pytest.raises(RuntimeError, match="null group key"):
        run_select(ops, host, {"s": ("sum", "v"), "by": "k"})
    # a filter that removes the null keys makes the query answerable again
    q = {"s": ("sum", "v"), "c": ("count", "a"), "by": "k", "where": (">", "k", 5)}
    check(run_select(ops, host, q), rfo.select({"from": host, **q}))


def run_update(ops, host, query):
    tab = H.table(host)
    d = H.select_dict(query, tab)
    r = ops.rfx_update(d)
    assert r, "null result"
    if H.is_error(r):
        msg = H.error_text(r)
        for o in (r, d, tab):
            ops.rfx_host_drop(o)
        raise RuntimeError(msg)
    out = H.table_to_numpy(r)
    for o in (r, d, tab):
        ops.rfx_host_drop(o)
    return out


UPDATES = [
    {"v": 99.5, "where": ("==", "k", 7)},                                   # atom under a filter (tests/lang.c:3060)
    {"v": ("*", "v", 1.5)},                                                 # element-wise, every row (tests/lang.c:3173)
    {"v": ("*", "v", 1.5), "where": (">", "a", 500_000)},                   # ... under a filter (tests/lang.c:3180)
    {"a": ("+", "a", ("*", "k", 10)), "where": ("and", ("<", "a", 300_000), (">", "v", 0.25))},
    {"n": 100},                                                             # new column, every row (tests/lang.c:3053)
    {"n": 7, "f": ("div", "a", 3), "where": ("<", "a", 100_000)},            # new columns: null outside the selection
    {"a": "k", "where": ("or", ("<", "a", 1000), (">", "v", 0.99))},         # a column as the mapping; nested where tree
    {"tot": ("sum", "v"), "by": "k"},                                       # per-group aggregate into a new column (tests/lang.c:3067)
    {"v": ("avg", "v"), "mx": ("max", "a"), "c": ("count", "a"), "by": "k"},
    {"v": ("sum", "v"), "fa": ("first", "a"), "by": "k", "where": ("<", "a", 600_000)},
    {"v": 1.0, "where": ("<", "a", -5)},                                    # nothing selected: the table comes back unchanged
]


@pytest.mark.parametrize("n,keys", [(1000, 13), (300_007, 5000)])
def test_update_where_by(ops, n, keys):
    """(update {...}) -- SURVEY 8f-4: K3 row ids + element-wise mappings / K7+K10 group aggregates + the two write kernels, against
    the oracle's restatement of ray_update (core/update.c:936-1106)."""
    host = host_table(n, keys=keys)
    host["a"][::97] = NULL
    for q in UPDATES:
        got = run_update(ops, host, q)
        assert ops.rfx_last_select_on_gpu() == 1, q
        want = rfo.update({"from": host, **q})
        check(got, want)
    # shapes that are the host's: a type conversion (f64 values into an i64 column), a sparse by: key
    with pytest.raises(RuntimeError, match="value type differs"):
        run_update(ops, host, {"a": 1.5, "where": ("<", "a", 10)})
    sparse = dict(host)
    sparse["k"] = host["k"] * 1_000_003
    with pytest.raises(RuntimeError, match="sparse or null keys"):
        run_update(ops, sparse, {"t": ("sum", "v"), "by": "k"})


def test_sampled_scope_at_the_operator_boundary(ops):
    """From 2^24 rows on rfx_select takes LDS-sized key scopes from a sample and its kernels report selected keys outside them (a report =
    the exact scope and the pass again): the same queries with the feature off (RFX_NO_SAMPLED_SCOPE) answer identically -- complete
    ranges, an outlier / a null key hidden between the sampled rows (unpinned columns: the changed cell is picked up by the payload
    validation), one and two key columns, with and without a filter."""
    import os
    n = (1 << 24) + 77
    host = {"k": rfo.gen_i64(n, 4, 100), "k2": rfo.gen_i64(n, 14, 7), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5)}
    queries = [{"s": ("sum", "a"), "c": ("count", "a"), "m": ("max", "a"), "f": ("first", "a"), "by": "k"},
               {"s": ("sum", "a"), "c": ("count", "v"), "by": {"x": "k", "y": "k2"}},
               {"s": ("sum", "a"), "mn": ("min", "v"), "where": ("<", "a", 500_000), "by": "k"}]

    state = {"tab": H.table(host)}  # ONE table object per content: the cache knows a column by its object (validation by ownership), and the planner
                                    # remembers by the device copy which key column's sampled scope was reported too small

    def ask(q):
        d = H.select_dict(q, state["tab"])
        r = ops.rfx_select(d)
        assert r, "null result"
        if H.is_error(r):
            msg = H.error_text(r)
            ops.rfx_host_drop(r)
            ops.rfx_host_drop(d)
            raise RuntimeError(msg)
        out = H.table_to_numpy(r)
        ops.rfx_host_drop(r)
        ops.rfx_host_drop(d)
        return out

    def rebuild():  # the host writes a cell: by the reference's rule that is a NEW vector (the old one is rc >= 2 while cached)
        ops.rfx_host_drop(state["tab"])
        state["tab"] = H.table(host)

    def both(q):
        got = ask(q)
        os.environ["RFX_NO_SAMPLED_SCOPE"] = "1"
        try:
            want = ask(q)
        finally:
            del os.environ["RFX_NO_SAMPLED_SCOPE"]
        assert list(got) == list(want)
        for name in want:
            assert np.array_equal(got[name].view(np.int64), want[name].view(np.int64)), name
        return got

    def stats():
        r = ops.rfx_stats(None)
        out = H.to_numpy(r).copy()
        ops.rfx_host_drop(r)
        return out

    s0 = stats()
    g = both(queries[0])
    assert np.array_equal(np.sort(g["k"]), np.arange(100)) and int(g["c"].sum()) == n
    assert np.array_equal(g["c"][np.argsort(g["k"])], np.bincount(host["k"], minlength=100))
    both(queries[1])
    both(queries[2])
    s1 = stats()
    assert s1[8] - s0[8] == 3 and s1[9] == s0[9], (s0, s1)  # three sampled scopes, none reported too small
    spot = 5 + 64 * 65_537  # never sampled (stride 64 from row 0, both ends 2^11 rows)
    for bad in (7_000, NULL):
        keep = host["k"][spot]
        host["k"][spot] = bad
        rebuild()
        try:
            if bad == NULL:  # a null group key is handed back to the host (no host here: loud failure), sampled scope or not
                with pytest.raises(RuntimeError, match="null group key"):
                    ask(queries[0])
            else:
                g = both(queries[0])
                assert len(g["k"]) == 101 and int(g["c"][g["k"] == bad][0]) == 1
                both(queries[1])
                assert stats()[9] - s1[9] == 1  # the first sampled scope was reported too small, the query ran again -- and that key column is not sampled a second time
        finally:
            host["k"][spot] = keep
    ops.rfx_host_drop(state["tab"])


def test_two_threads_hammer_the_operator_boundary(ops):
    """The reference calls built-ins from its pool workers (core/pool.c:168-219): two threads calling rfx_select / rfx_sum / rfx_lt at once
    must each get their own right answer (one lock around the residency cache, the per-call scratch and the device context).  Objects
    are built up front on the main thread (the standalone host model's intern table is not the thing under test)."""
    import threading
    n = 200_003
    hosts = [host_table(n, keys=300, seed=s) for s in (0, 7)]
    tabs = [H.table(h) for h in hosts]
    queries = [{"where": ("<", "a", 400_000), "by": "k", "s": ("sum", "v"), "c": ("count", "a")}, {"where": (">", "v", 0.25), "s": ("sum", "a"), "m": ("max", "v")}]
    dicts = [[H.select_dict(q, t) for q in queries] for t in tabs]
    vecs = [H.vector(h["a"]) for h in hosts]
    want_sel = [[rfo.select({"from": h, **q}) for q in queries] for h in hosts]
    want_sum = [int(np.sum(h["a"])) for h in hosts]
    errors = []

    def work(t):
        try:
            for it in range(60):
                for qi in range(2):
                    r = ops.rfx_select(dicts[t][qi])
                    assert r and not H.is_error(r), H.error_text(r)
                    got = H.table_to_numpy(r)
                    ops.rfx_host_drop(r)
                    for name, w in want_sel[t][qi].items():
                        g = got[name]
                        if w.dtype == np.float64:
                            assert np.allclose(g, w, rtol=1e-9, atol=0), (t, qi, name, it)
                        else:
                            assert np.array_equal(g, w), (t, qi, name, it)
                r = ops.rfx_sum(vecs[t])
                assert r and not H.is_error(r)
                assert int(C.c_int64.from_address(r + 8).value) == want_sum[t], (t, it)  # (an atom keeps its value where a vector keeps its length)
                ops.rfx_host_drop(r)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for o in [d for dd in dicts for d in dd] + vecs + tabs:
        ops.rfx_host_drop(o)
    assert not errors, errors[:3]


def test_and_or_as_special_forms(ops):
    """rfx_and_sf / rfx_or_sf take UNEVALUATED arms like the reference's special forms (core/env.c:224-225, core/logic.c:89-260): comparison
    trees over vectors -- nested and / or, arithmetic operands -- become one mask on the device; evaluated B8 masks are combined as rfx_and
    does; without a host to delegate to, anything else fails loudly."""
    n = 100_003
    h = host_table(n)
    a, v, k = H.vector(h["a"]), H.vector(h["v"]), H.vector(h["k"])

    def call(fn, *arms):
        arr = (C.c_void_p * len(arms))(*arms)
        r = fn(arr, len(arms))
        assert r and not H.is_error(r), H.error_text(r)
        out = H.to_numpy(r)
        ops.rfx_host_drop(r)
        return out

    fnobj = lambda nm: ops.rfx_host_fn(nm.encode())
    cmp_ = lambda op, vec, atom: H.list_of([fnobj(op), ops.rfx_host_clone(vec), H.atom(atom)])  # the vector object itself as the operand (it evaluates to itself)
    lt, gt, ne = cmp_("<", a, 300_000), cmp_(">", v, 0.5), cmp_("!=", k, 7)
    want_and = (rfo.cmp("<", h["a"], 300_000) & rfo.cmp(">", h["v"], 0.5) & rfo.cmp("!=", h["k"], 7)).astype(np.int8)
    assert np.array_equal(call(ops.rfx_and_sf, lt, gt, ne), want_and)
    want_or = (rfo.cmp("<", h["a"], 300_000) | rfo.cmp(">", h["v"], 0.5)).astype(np.int8)
    assert np.array_equal(call(ops.rfx_or_sf, lt, gt), want_or)
    # a nested tree, an arithmetic operand: (or (and (< a 300000) (> v 0.5)) (== (% a 3) 1))
    inner = H.list_of([fnobj("and"), ops.rfx_host_clone(lt), ops.rfx_host_clone(gt)])
    mod = H.list_of([fnobj("=="), H.list_of([fnobj("%"), ops.rfx_host_clone(a), H.atom(3)]), H.atom(1)])
    want = ((rfo.cmp("<", h["a"], 300_000) & rfo.cmp(">", h["v"], 0.5)) | rfo.cmp("==", rfo.binop("%", h["a"], 3), 1)).astype(np.int8)
    assert np.array_equal(call(ops.rfx_or_sf, inner, mod), want)
    # evaluated masks (a loader that evaluates the arguments first): the plain and / or
    m1, m2 = H.vector(rfo.cmp("<", h["a"], 300_000).astype(np.int8)), H.vector(rfo.cmp(">", h["v"], 0.5).astype(np.int8))
    assert np.array_equal(call(ops.rfx_and_sf, m1, m2), (rfo.cmp("<", h["a"], 300_000) & rfo.cmp(">", h["v"], 0.5)).astype(np.int8))
    # not a comparison tree and no host to hand it to: a loud error, never a guess
    arr = (C.c_void_p * 2)(lt, a)
    r = ops.rfx_and_sf(arr, 2)
    assert r and H.is_error(r) and "not covered" in H.error_text(r)
    ops.rfx_host_drop(r)
    for o in (lt, gt, ne, inner, mod, m1, m2, a, v, k):
        ops.rfx_host_drop(o)
