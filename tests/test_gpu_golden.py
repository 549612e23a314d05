"""The HIP path against the golden vectors captured from the compiled reference (tests/golden/) -- no oracle in between."""
import numpy as np
import pytest

import golden_cases as G

pytestmark = pytest.mark.gpu
NULL = -(2**63)


def dev(eng, t):
    return {k: eng.column(v) for k, v in t.items()}


def test_cmp_truth_tables(eng):
    for op, tag, l, r, want in G.cmp_special_cases():
        dl = eng.column(l)
        dr = eng.column(r) if isinstance(r, np.ndarray) else r
        got = eng.cmp(op, dl, dr).cpu().numpy()
        assert np.array_equal(got, want), (op, tag)


def test_scalar_aggregates_and_where_ids(eng):
    for name, t, w, want, ids in G.scalar_cases():
        d = dev(eng, t)
        q = {"from": d, **G.SCALAR_Q}
        if w is not None:
            q["where"] = w
        got = eng.select(q)
        for o in want:
            G.same(got[o].cpu().numpy(), want[o], f"{name}.{o}")
        if ids is not None:
            assert np.array_equal(eng.where(w, d).cpu().numpy(), ids), name


def test_group_by_order_and_aggregates(eng):
    for name, t, w, want in G.group_cases():
        q = {"from": dev(eng, t), "by": "k", **G.GROUP_Q}
        if w is not None:
            q["where"] = w
        got = eng.select(q)
        assert np.array_equal(got["k"].cpu().numpy(), want["k"]), f"{name}: group keys / first-occurrence order"
        for o in G.GROUP_Q:
            G.same(got[o].cpu().numpy(), want[o], f"{name}.{o}")


@pytest.mark.parametrize("case", list(G.multikey_cases()), ids=lambda c: c[0])
def test_group_by_several_keys(eng, case):
    _, t, names, want = case
    got = eng.select({"from": {k: eng.column(v) for k, v in t.items()}, "by": {nm: nm for nm in names}, **G.MULTIKEY_Q})
    assert list(got.keys()) == list(want.keys())
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)


@pytest.mark.parametrize("case", list(G.rowhash_cases()), ids=lambda c: c[0])
def test_group_by_several_keys_row_hash_path(eng, case):
    """Ranges beyond 64 bits / null keys: grouped on the reference's row hash (k_row_hash + sparse-key machinery, collision-proof
    through per-group min == max of every key column); groups, key columns, order (both arms of the reference) and values."""
    _, t, names, order, want = case
    got = eng.select({"from": {k: eng.column(v) for k, v in t.items()}, "by": {nm: nm for nm in names}, "order": order, **G.MULTIKEY_Q})
    assert list(got.keys()) == list(want.keys())
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)


def test_binop_truth_tables_as_columns(eng):
    """The reference's element-wise + - * div answers on the special-value vectors (48 tables), through rfx_hip_eval_expr."""
    for op, tag, l, r, want in G.binop_cases():
        t, e = {}, [op, l, r]
        for i, x in enumerate((l, r)):
            if isinstance(x, np.ndarray):
                t[f"c{i}"] = eng.column(x)
                e[1 + i] = f"c{i}"
        got = eng.eval_expr(tuple(e), t).cpu().numpy()
        assert got.dtype == want.dtype, (op, tag)
        if want.dtype == np.float64:
            fin = np.isfinite(want)
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)]), (op, tag)
            assert np.all(np.abs(got[fin] - want[fin]) <= 2.3e-16 * np.abs(want[fin])), (op, tag)
        else:
            assert np.array_equal(got, want), (op, tag)


@pytest.mark.parametrize("case", list(G.join_cases()), ids=lambda c: c[0])
def test_equi_joins(eng, case):
    """left-join / inner-join against the reference's answers: dense and hashed first-occurrence tables, two key columns through
    the row hash, null keys; right-only columns of the left join (generic lists in the reference) against the oracle."""
    _, keys, left, right, want_lj, want_ij = case
    dl, dr = {k: eng.column(v) for k, v in left.items()}, {k: eng.column(v) for k, v in right.items()}
    got = eng.left_join(keys, dl, dr)
    assert list(got.keys()) == keys + ["a", "v", "w", "z"]
    for o in want_lj:
        G.same(got[o].cpu().numpy(), want_lj[o], "lj " + o)
    from oracle import rfo  # the reference cannot hand over its generic-list columns / dies on some null-key joins: oracle there
    ora = rfo.left_join(keys, left, right)
    for o in ora:
        G.same(got[o].cpu().numpy(), ora[o], "lj " + o)
    gi = eng.inner_join(keys, dl, dr)
    want = want_ij or rfo.inner_join(keys, left, right)
    assert list(gi.keys()) == list(want.keys())
    for o in want:
        G.same(gi[o].cpu().numpy(), want[o], "ij " + o)
    assert np.array_equal(eng.join_index(keys, dl, dr).cpu().numpy(), rfo.join_index(keys, left, right))


@pytest.mark.parametrize("case", list(G.xagg_cases()), ids=lambda c: c[0])
def test_aggregates_over_expressions(eng, case):
    """(sum (* a v)) & co: folded on the fly in the scalar and LDS-table kernels, materialised for the partitioned path."""
    _, t, w, by, want = case
    q = {"from": {k: eng.column(v) for k, v in t.items()}, **G.XQ}
    if w:
        q["where"] = w
    if by:
        q["by"] = by
    got = eng.select(q)
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)


@pytest.mark.parametrize("case", list(G.q1_cases()), ids=lambda c: c[0])
def test_nested_expressions_q1_shape(eng, case):
    """TPC-H Q1 shape: nine aggregates, four of them over expressions up to three operations deep, folded on the fly."""
    _, t, extra, want = case
    got = eng.select({"from": {k: eng.column(v) for k, v in t.items()}, **G.Q1, **extra})
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)


def test_xbar_buckets(eng):
    x, tables, t, want = G.xbar_case()
    from rayforce_amd import _lib as L
    d = eng.column(x)
    for w, ref_out in tables.items():
        out = eng.empty(len(x))
        L.check(eng.lib.rfx_hip_xbar_i64(eng._ctx, d.data_ptr(), len(x), w, out.data_ptr()))
        assert np.array_equal(out.cpu().numpy(), ref_out), w
    got = eng.select({"from": {k: eng.column(v) for k, v in t.items()}, "by": {"b": ("xbar", "ts", 1000)}, "s": ("sum", "v"), "c": ("count", "a")})
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)


def test_group_by_sparse_keys(eng):
    t, want = G.sparse_case()
    got = eng.select({"from": dev(eng, t), "by": "k", "sf": ("sum", "v"), "c": ("count", "a"), "mxi": ("max", "a")})
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)


def test_null_semantics(eng):
    t, want, scalar_sum = G.nullsem_case()
    got = eng.select({"from": dev(eng, t), "by": "k", "s": ("sum", "v"), "fs": ("sum", "f"), "mn": ("min", "v"), "mx": ("max", "v"),
                      "fmn": ("min", "f"), "fmx": ("max", "f"), "c": ("count", "v"), "av": ("avg", "v")})
    for o in want:
        G.same(got[o].cpu().numpy(), want[o], o)
    assert int(eng.select({"from": dev(eng, t), "s": ("sum", "v")})["s"][0]) == scalar_sum


def test_hash_primitives(eng):
    from rayforce_amd import _lib as L
    keys = G.arr("hash_keys")
    d = eng.column(keys)
    out = eng.empty(len(keys))
    L.check(eng.lib.rfx_hip_hash_fnv1a_i64(eng._ctx, d.data_ptr(), len(keys), out.data_ptr()))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), G.arr("hash_fnv1a"))
    assert G.has("hash_index_u64")  # pinned by tests/golden/hash_index_harness.c (the reference's own `inline` header function)
    L.check(eng.lib.rfx_hip_hash_mix_u64(eng._ctx, d.data_ptr(), len(keys), 0x9ddfea08eb382d69, out.data_ptr()))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), G.arr("hash_index_u64"))


# ---------------------------------------------------------------- `/` (ray_div) and `%` (ray_mod): SURVEY 8f-3, second half
import divmod_cases as DM  # noqa: E402


def test_div_mod_truth_tables_as_columns(eng):
    """The reference's `/` and `%` on the special-value vectors (32 tables) through rfx_hip_eval_expr: bit for bit (floor and the fused
    remainder are exact operations: no tolerance)."""
    for op, tag, l, r, want in DM.truth_tables():
        t, e = {}, [op, l, r]
        for i, x in enumerate((l, r)):
            if isinstance(x, np.ndarray):
                t[f"c{i}"] = eng.column(x)
                e[1 + i] = f"c{i}"
        DM.same(eng.eval_expr(tuple(e), t).cpu().numpy(), want, (op, tag))


@pytest.mark.parametrize("case", list(DM.query_cases()), ids=lambda c: c[0])
def test_aggregates_over_div_mod(eng, case):
    _, (n, seed, keys), w, grouped, want = case
    q = {"from": {k: eng.column(v) for k, v in DM.gen_table(n, seed, keys).items()}, **DM.XQ}
    if w:
        q["where"] = w
    if grouped:
        q["by"] = "k"
    got = eng.select(q)
    for o in want:
        DM.same(got[o].cpu().numpy(), want[o], o, sums=DM.XQ.get(o, ("", ""))[0] in ("sum", "avg"))
