"""GPU parity: every HIP entry point against the CPU oracle on the same seeded inputs (bit-exact for integer / index /
key-order output, <= 1e-9 relative for f64 sums and averages -- the tolerance BASELINE.json's north_star states).

All calls go through the C ABI of librfx.so (rayforce_amd.engine is a ctypes host).  The oracle (oracle/rfo.py) is the
checker only."""
import math

import os

import numpy as np
import pytest
import torch

from oracle import rfo

pytestmark = pytest.mark.gpu

NULL = -(2**63)
RTOL = 1e-9


def dev(eng, table):
    return {k: eng.column(v) for k, v in table.items()}


def same_f64(a, b, rtol=RTOL, scale=None):
    """<= 1e-9 relative (the north star's bound for f64 sums).  `scale` = the per-group sum of |x_i|: the bound a
    summation in ANY order can be held to -- a group whose terms cancel (sum of w in [-0.5, 0.5)) has |result| << sum |x_i|, and the
    device's order (atomics) differs from the CPU's chunks.  Without `scale` (min / max / first, exact operations) the value itself."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "NaN pattern differs"
    ok = ~nan_a
    if ok.any():
        inf = np.isinf(a[ok]) | np.isinf(b[ok])
        assert np.array_equal(a[ok][inf], b[ok][inf])
        fin = ~inf
        ref_scale = np.abs(b[ok][fin]) if scale is None else np.maximum(np.abs(np.asarray(scale, np.float64)[ok][fin]), np.abs(b[ok][fin]))
        ref_scale = np.maximum(np.where(np.isnan(ref_scale), np.abs(b[ok][fin]), ref_scale), 1e-300)
        assert np.all(np.abs(a[ok][fin] - b[ok][fin]) <= rtol * ref_scale), float(np.max(np.abs(a[ok][fin] - b[ok][fin]) / ref_scale))


def _abs_scale(host, q, name):
    """sum / avg of |column| over the same selection and groups: the scale of the f64 tolerance for output `name` (None: exact op)."""
    spec = q[name]
    if not (isinstance(spec, tuple) and spec[0] in ("sum", "avg") and isinstance(spec[1], str)):
        return None
    col = host[spec[1]]
    habs = dict(host)
    habs["__abs"] = np.abs(col.astype(np.float64)) if col.dtype == np.float64 else np.abs(np.where(col == NULL, 0, col)).astype(np.float64)
    q2 = {k: v for k, v in q.items() if k in ("where", "by")}
    q2[name] = (spec[0], "__abs")
    return rfo.select({"from": habs, **q2})[name]


def check_select(eng, host, q, devt=None):
    got = eng.select({"from": dev(eng, host) if devt is None else devt, **q})
    want = rfo.select({"from": host, **q})
    assert list(got.keys()) == list(want.keys())
    for name in want:
        g = got[name].cpu().numpy()
        w = want[name]
        assert g.dtype == w.dtype, (name, g.dtype, w.dtype)
        if w.dtype == np.float64:
            same_f64(g, w, scale=_abs_scale(host, q, name) if name in q else None)
        else:
            assert np.array_equal(g, w), name
    return got


def table(n, seed=0, keys=1000, nulls=False):
    t = {
        "k": rfo.gen_i64(n, 4 + seed, keys),
        "a": rfo.gen_i64(n, 2 + seed, 1_000_000),
        "v": rfo.gen_f64(n, 5 + seed),
        "w": rfo.gen_f64(n, 6 + seed) - 0.5,
    }
    if nulls and n:
        r = rfo.gen_i64(n, 99 + seed, 100)
        t["a"][r == 0] = NULL
        t["v"][r == 1] = np.nan
        t["w"][r == 2] = np.nan
    return t


# ---------------------------------------------------------------- generator
@pytest.mark.parametrize("n,row0", [(0, 0), (1, 0), (1000, 7), (100_003, 12345)])
def test_generator_matches_oracle(eng, n, row0):
    assert np.array_equal(eng.gen_i64(n, 2, 1_000_000, row0).cpu().numpy(), rfo.gen_i64(n, 2, 1_000_000, row0))
    assert np.array_equal(eng.gen_f64(n, 5, row0).cpu().numpy(), rfo.gen_f64(n, 5, row0))


# ---------------------------------------------------------------- K1/K5 fused filter -> aggregates
SIZES = [0, 1, 2, 63, 64, 511, 512, 513, 2047, 2048, 4097, 25_001, 100_003, 1_000_003]


@pytest.mark.parametrize("n", SIZES)
def test_filter_sum_i64_bit_exact(eng, n):
    host = table(n)
    check_select(eng, host, {"where": ("<", "a", 100_000), "s": ("sum", "a"), "c": ("count", "a")})


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("nulls", [False, True])
def test_all_scalar_aggregates(eng, n, nulls):
    host = table(n, nulls=nulls)
    q = {"where": ("<", "a", 500_000), "s": ("sum", "a"), "f": ("sum", "v"), "mn": ("min", "a"), "mx": ("max", "a"),
         "fm": ("min", "w"), "fx": ("max", "w"), "av": ("avg", "v"), "ai": ("avg", "a")}
    check_select(eng, host, q)
    check_select(eng, host, {"c": ("count", "v"), "s": ("sum", "w"), "first": ("first", "a")})


@pytest.mark.parametrize("op", ["==", "!=", "<", ">", "<=", ">="])
@pytest.mark.parametrize("kind", ["i64_i64", "i64_f64", "f64_f64", "f64_i64", "vec_vec", "vec_vec_mixed"])
def test_comparison_matrix(eng, op, kind):
    n = 10_007
    host = table(n, nulls=True)
    host["b"] = rfo.gen_i64(n, 77, 1_000_000)
    lhs, rhs = {"i64_i64": ("a", 500_000), "i64_f64": ("a", 499_999.5), "f64_f64": ("v", 0.5), "f64_i64": ("w", 0),
                "vec_vec": ("a", "b"), "vec_vec_mixed": ("a", "v")}[kind]
    d = dev(eng, host)
    mask = eng.cmp(op, d[lhs], d[rhs] if isinstance(rhs, str) else rhs).cpu().numpy()
    want = rfo.cmp(op, host[lhs], host[rhs] if isinstance(rhs, str) else rhs)
    assert np.array_equal(mask, want)
    check_select(eng, host, {"where": (op, lhs, rhs), "c": ("count", "a"), "s": ("sum", "a")})


def test_null_and_nan_ordering(eng):
    # core/ops.h:96-121 : 0Nl < anything ; NaN is the smallest f64 ; NaN == NaN ; -0.0 == 0.0
    a = np.array([1, NULL, 3, NULL, -5], np.int64)
    f = np.array([0.0, -0.0, np.nan, 1.5, -np.inf], np.float64)
    host = {"a": a, "f": f}
    d = dev(eng, host)
    for op in ["==", "!=", "<", ">", "<=", ">="]:
        for rhs in (2, NULL, 0):
            assert np.array_equal(eng.cmp(op, d["a"], rhs).cpu().numpy(), rfo.cmp(op, a, rhs)), (op, rhs)
        for rhs in (0.0, -0.0, float("nan"), 1.5):
            assert np.array_equal(eng.cmp(op, d["f"], rhs).cpu().numpy(), rfo.cmp(op, f, rhs)), (op, rhs)
    assert eng.cmp("<", d["a"], 2).cpu().tolist() == [1, 1, 0, 1, 1]


def test_multi_predicate_and_or(eng):
    host = table(200_003, nulls=True)
    w3 = ("and", ("<", "v", 0.316228), (">", "w", 0.183772), ("!=", "a", 250_000))
    check_select(eng, host, {"where": w3, "av": ("avg", "w"), "mn": ("min", "w"), "mx": ("max", "w"), "c": ("count", "w")})
    wo = ("or", ("<", "a", 1000), (">", "v", 0.999), ("==", "k", 7))
    check_select(eng, host, {"where": wo, "s": ("sum", "a"), "c": ("count", "a")})
    nested = ("and", ("or", ("<", "a", 1000), (">", "v", 0.9)), ("!=", "k", 3))
    check_select(eng, host, {"where": nested, "s": ("sum", "a"), "f": ("sum", "v")})


def test_empty_selection_rules(eng):
    # tests/lang.c:2508,2535,4070 : empty sum = 0, empty min/max = null, avg = NaN
    host = table(5000)
    got = check_select(eng, host, {"where": ("<", "a", -1), "s": ("sum", "a"), "f": ("sum", "v"), "mn": ("min", "a"), "mx": ("max", "v"),
                                   "av": ("avg", "a"), "c": ("count", "a")})
    assert int(got["s"][0]) == 0 and int(got["c"][0]) == 0 and int(got["mn"][0]) == NULL
    assert math.isnan(float(got["mx"][0])) and math.isnan(float(got["av"][0]))


def test_i64_sum_wraps(eng):
    a = np.full(1000, 2**62, np.int64)
    got = eng.select({"from": dev(eng, {"a": a}), "s": ("sum", "a")})
    assert int(got["s"][0]) == int(np.sum(a.astype(np.uint64)).astype(np.int64))
    assert int(got["s"][0]) == int(rfo.select({"from": {"a": a}, "s": ("sum", "a")})["s"][0])


# ---------------------------------------------------------------- K2 masks, K3 where, K4 gather
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 511, 512, 513, 25_001, 1_000_003])
def test_where_ids_bit_exact(eng, n):
    host = table(n)
    d = dev(eng, host)
    for spec in [("<", "a", 100_000), ("<", "a", 10), (">=", "a", 0), ("and", ("<", "a", 500_000), (">", "v", 0.5))]:
        want = rfo.where(rfo.mask_of(spec, host))
        got = eng.where(spec, d)
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want), spec
        got_m = eng.where(eng.mask_of(spec, d))
        assert np.array_equal(got_m.cpu().numpy(), want), spec
    if n:
        assert np.array_equal(eng.where(("<", "a", 100_000), d, row0=10**12).cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 100_000), host)) + 10**12)


NO_WHERE_ONCE = 2097152  # RFX_TUNE_NO_WHERE_ONCE: rfx_hip_where_once takes the two-pass form (bitmap -> scan -> emit)


@pytest.mark.parametrize("flags", [0, NO_WHERE_ONCE])
def test_where_one_pass_and_two_pass_agree(eng, flags):
    """rfx_where_once.hip: ballots -> decoupled look-back -> ids in one kernel.  5e6 rows = 306 tiles (several look-back rounds of 64), a
    ragged last tile, 1 .. 4 predicate columns and 1 .. 4 comparisons (the kernel's instantiations), five columns (the two-pass form behind the
    same entry point), every / no row selected, a row offset; RFX_TUNE_NO_WHERE_ONCE runs the same calls through the two-pass form."""
    n = 5_000_017
    host = table(n)
    host["b"] = rfo.gen_i64(n, 77, 1000)
    d = dev(eng, host)
    specs = [("<", "a", 100_000), ("<", "a", 0), (">=", "a", 0), ("and", ("<", "a", 500_000), (">", "v", 0.5)),
             ("or", ("<", "a", 1000), (">", "w", 0.49), ("==", "k", 7)),
             ("and", ("<", "a", 900_000), (">", "v", 0.1), ("<", "w", 0.4), ("!=", "k", 3)),
             ("and", (">", "a", "b"), ("<", "v", 0.3)),
             ("and", ("<", "a", 900_000), (">", "v", 0.1), ("<", "w", 0.4), ("!=", "k", 3), ("<", "b", 900)),
             ("and", ("or", ("<", "a", 1000), (">", "w", 0.45)), ("<", "v", 0.9))]
    try:
        eng.tune(flags=flags)
        for spec in specs:
            want = rfo.where(rfo.mask_of(spec, host))
            got = eng.where(spec, d)
            assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want), spec
        got = eng.where(specs[0], d, row0=10**12)
        assert np.array_equal(got.cpu().numpy(), rfo.where(rfo.mask_of(specs[0], host)) + 10**12)
    finally:
        eng.tune(flags=0)


def test_where_plan_kernels(eng):
    """rfx_rtc.hip: the one-pass `where` compiled for ONE predicate list (operators, domains and conversions as constants) -- the same specs as
    above through kernels built at first sight (RFX_RTC_EAGER), against the oracle; atoms are kernel arguments (a second constant reuses the
    kernel: no further compilation); a deeper tree, null / NaN atoms and column-against-column comparisons included."""
    n = 3_000_019
    host = table(n)
    host["b"] = rfo.gen_i64(n, 77, 1000)
    host["a"][5::1013] = L_NULL
    host["v"][3::977] = np.nan
    d = dev(eng, host)
    specs = [("<", "a", 100_000), ("!=", "a", L_NULL), (">=", "v", 0.25), ("==", "v", float("nan")),
             ("and", ("<", "a", 500_000), (">", "v", 0.5)),
             ("or", ("<", "a", 1000), (">", "w", 0.49), ("==", "k", 7)),
             ("and", ("<", "a", 900_000), (">", "v", 0.1), ("<", "w", 0.4), ("!=", "k", 3)),
             ("and", (">", "a", "b"), ("<", "v", 0.3)), ("<", "a", "v"),
             ("and", ("or", ("<", "a", 1000), (">", "w", 0.45)), ("<", "v", 0.9)),
             ("or", ("and", ("<", "a", 300_000), ("or", (">", "w", 0.3), ("<", "v", 0.2))), ("==", "k", 5))]
    l0, c0 = _rtc_stats(eng)
    os.environ["RFX_RTC_EAGER"] = "1"
    try:
        for spec in specs:
            want = rfo.where(rfo.mask_of(spec, host))
            got = eng.where(spec, d)
            assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want), spec
        l1, c1 = _rtc_stats(eng)
        if c1 == c0 and l1 == l0:
            pytest.skip("no run-time compiler on this box: the prebuilt kernels answered")
        assert l1 - l0 >= len(specs)
        got = eng.where(("<", "a", 7_777), d, row0=10**12)  # another atom, a row offset: the first spec's kernel
        assert np.array_equal(got.cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 7_777), host)) + 10**12)
        assert _rtc_stats(eng)[1] == c1
    finally:
        del os.environ["RFX_RTC_EAGER"]


def test_where_one_pass_estimate_and_overflow(eng):
    """The buffer of the one-pass `where` is sized by 2^15 strided rows.  A selection that sits between the sampled rows (a contiguous block:
    the sample sees a few of its rows at most... here none, the block lies inside one stride) is underestimated only within the 1 % margin;
    one that is far larger than the sample says (every row of 30 % of the column, none of them on the sample's stride) overflows the buffer:
    the count comes back exact with RFX_ELIMIT, and the second call with that capacity writes every id."""
    import ctypes as C
    from rayforce_amd import _lib as L
    n = 4_000_037
    stride = n // (1 << 15)
    a = np.ones(n, dtype=np.int64)
    a[np.arange(n) % stride == 0] = 0  # the sampled rows: not selected; everything else (99 %) is
    d = {"a": eng.column(a)}
    p = (L.Pred * 1)()
    p[0].d_col, p[0].col_type, p[0].rhs_type, p[0].op, p[0].rhs_i = d["a"].data_ptr(), L.RFX_I64, L.RFX_I64, L.OPS["=="], 1
    est, cnt = C.c_int64(), C.c_int64()
    L.check(eng.lib.rfx_hip_where_estimate(eng._ctx, p, 1, L.RFX_AND, n, C.byref(est)))
    want = np.nonzero(a == 1)[0]
    assert est.value < len(want)  # the sample saw nothing
    out = torch.full((est.value + 64,), -1, dtype=torch.int64, device=eng.device)
    rc = eng.lib.rfx_hip_where_once(eng._ctx, p, 1, L.RFX_AND, n, 0, out.data_ptr(), est.value, C.byref(cnt))
    assert rc == L.RFX_ELIMIT and cnt.value == len(want)
    assert np.array_equal(out[:est.value].cpu().numpy(), want[:est.value]) and bool((out[est.value:] == -1).all())  # nothing beyond cap
    assert np.array_equal(eng.where(("==", "a", 1), d).cpu().numpy(), want)  # the Engine runs it again with the exact size
    # counting only (cap 0, no buffer)
    rc = eng.lib.rfx_hip_where_once(eng._ctx, p, 1, L.RFX_AND, n, 0, None, 0, C.byref(cnt))
    assert rc == L.RFX_ELIMIT and cnt.value == len(want)


def test_where_across_parallel_threshold(eng):
    # tests/lang.c:2893-2897 crosses POOL_SPLIT_THRESHOLD (16 384) at 25 001 rows
    n = 25_001
    a = np.arange(n, dtype=np.int64)
    d = {"a": eng.column(a)}
    got = eng.where(("and", (">=", "a", 5), ("<", "a", 20_000)), d)
    assert np.array_equal(got.cpu().numpy(), np.arange(5, 20_000, dtype=np.int64))


def test_mask_logic_and_gather(eng):
    n = 100_003
    host = table(n)
    d = dev(eng, host)
    m1, m2 = eng.lt(d["a"], 500_000), eng.gt(d["v"], 0.25)
    assert np.array_equal(eng.and_(m1, m2).cpu().numpy(), rfo.and_(rfo.cmp("<", host["a"], 500_000), rfo.cmp(">", host["v"], 0.25)))
    assert np.array_equal(eng.or_(m1, m2).cpu().numpy(), rfo.or_(rfo.cmp("<", host["a"], 500_000), rfo.cmp(">", host["v"], 0.25)))
    ids = eng.where(m1)
    assert np.array_equal(eng.at_ids(d["v"], ids).cpu().numpy(), rfo.at_ids(host["v"], ids.cpu().numpy()))
    assert np.array_equal(eng.at_ids(d["a"], ids).cpu().numpy(), host["a"][ids.cpu().numpy()])
    # select without aggregates = filter_collect of every column (core/filter.c:51-165)
    check_select(eng, host, {"where": ("<", "a", 1000)})


# ---------------------------------------------------------------- group-by
GSIZES = [1, 2, 513, 25_001, 300_007]


@pytest.mark.parametrize("n", GSIZES)
@pytest.mark.parametrize("keys", [1, 7, 1000, 20_000])
def test_group_by_dense_order_and_aggregates(eng, n, keys):
    host = table(n, keys=keys)
    q = {"by": "k", "s": ("sum", "v"), "si": ("sum", "a"), "c": ("count", "a"), "mn": ("min", "a"), "mx": ("max", "w"), "av": ("avg", "v"),
         "ai": ("avg", "a")}
    check_select(eng, host, q)


@pytest.mark.parametrize("n", [1000, 200_003])
def test_group_by_with_where_and_nulls(eng, n):
    host = table(n, keys=300, nulls=True)
    q = {"where": ("and", ("<", "a", 700_000), (">", "v", 0.1)), "by": "k", "s": ("sum", "v"), "si": ("sum", "a"), "c": ("count", "v"),
         "mn": ("min", "w"), "mx": ("max", "a"), "av": ("avg", "w"), "fi": ("first", "a")}
    check_select(eng, host, q)


def test_group_null_semantics_golden(eng):
    # SURVEY 0.6 (oracle-verified): grouped sum PROPAGATES null, scalar sum skips it; grouped min of an all-null group is
    # INF, max is null; count counts nulls; avg divides by the non-null count.
    k = np.array([1, 1, 2, 3, 3], np.int64)
    v = np.array([1, NULL, 5, NULL, NULL], np.int64)
    f = np.array([1.0, np.nan, 5.0, np.nan, np.nan])
    host = {"k": k, "v": v, "f": f}
    got = check_select(eng, host, {"by": "k", "s": ("sum", "v"), "fs": ("sum", "f"), "mn": ("min", "v"), "mx": ("max", "v"), "fmn": ("min", "f"),
                                   "fmx": ("max", "f"), "c": ("count", "v"), "av": ("avg", "v")})
    assert got["k"].tolist() == [1, 2, 3]
    assert got["s"].tolist() == [NULL, 5, NULL]
    assert got["mn"].tolist() == [1, 5, 2**63 - 1] and got["mx"].tolist() == [1, 5, NULL]
    assert got["c"].tolist() == [2, 1, 2]
    assert got["fmn"].cpu().numpy()[2] == np.inf and np.isnan(got["fmx"].cpu().numpy()[2])
    assert int(eng.select({"from": dev(eng, host), "s": ("sum", "v")})["s"][0]) == 6


def test_group_first_occurrence_order(eng):
    # keys [3 1 3 2 1 3] -> groups 3, 1, 2
    host = {"k": np.array([3, 1, 3, 2, 1, 3], np.int64), "v": np.arange(6, dtype=np.float64)}
    got = check_select(eng, host, {"by": "k", "s": ("sum", "v")})
    assert got["k"].tolist() == [3, 1, 2] and got["s"].tolist() == [7.0, 5.0, 3.0]


def test_group_by_large_range_paths(eng):
    # range 1e6 > LDS table budget -> partitioned / device-atomic path; IDS vs SHIFT boundary (524 288) on both sides
    n = 2_000_003
    for keys in (524_288, 524_289, 1_000_000):
        host = table(n, keys=keys)
        check_select(eng, host, {"by": "k", "s": ("sum", "v"), "c": ("count", "v")})
        check_select(eng, host, {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v"), "mx": ("max", "a")})


def test_group_by_sparse_keys_hash_path(eng):
    # range > rows -> open-addressed path (core/index.c:1959-1977).  Order contract there is key -> aggregate map only
    # when the reference runs multi-threaded; single-threaded (what the oracle restates) it is first occurrence.
    n = 50_000
    host = table(n, keys=1000)
    host["k"] = host["k"] * 1_000_003 - 77  # spread 1000 distinct keys over a huge range
    got = check_select(eng, host, {"by": "k", "s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", "a")})
    assert len(got["k"]) == 1000
    host["k"][::5] = rfo.gen_i64(n, 123, 2**62)[::5]  # many distinct keys
    check_select(eng, host, {"where": (">", "v", 0.5), "by": "k", "s": ("sum", "a")})


@pytest.mark.parametrize("flags", [0, 2, 2048])
def test_sparse_keys_partitioned_hash_path(eng, flags):
    """range > rows on inputs large enough for the partitioned form (hash partition -> LDS tables -> one merge) and, with
    RFX_TUNE_NO_PARTITION, the direct device-wide table: few keys (everything lives in LDS), ~LDS-capacity keys per partition,
    far more (LDS overflow -> direct updates; device-wide table grows x16), null keys, predicates, 0..2 value planes."""
    n = 400_003
    try:
        eng.tune(flags=flags)
        for distinct in (5, 3000, 200_000, 399_000):
            host = table(n, keys=distinct, nulls=True)
            host["k"] = host["k"] * 1_000_003 - 77_777
            check_select(eng, host, {"by": "k", "s": ("sum", "v"), "c": ("count", "a")})
            check_select(eng, host, {"by": "k", "mx": ("max", "a"), "av": ("avg", "w"), "f": ("first", "v")})
            check_select(eng, host, {"where": ("<", "a", 400_000), "by": "k", "s": ("sum", ("*", "v", "a")), "c": ("count", "a")})
            check_select(eng, host, {"by": "k", "c": ("count", "a")})
        # null keys: ONE group here (dedicated slot), where the reference opens a group per null row (DESIGN.md deviation 3)
        host = table(n, keys=3000)
        host["k"] = host["k"] * 1_000_003 - 77_777
        nul = rfo.gen_i64(n, 321, 97) == 0
        host["k"][nul] = NULL
        got = eng.select({"from": dev(eng, host), "by": "k", "c": ("count", "a"), "s": ("sum", "v")})
        gk = got["k"].cpu().numpy()
        assert (gk == NULL).sum() == 1 and len(gk) == len(np.unique(host["k"]))
        i = int(np.nonzero(gk == NULL)[0][0])
        assert int(got["c"][i]) == int(nul.sum()) and abs(float(got["s"][i]) - host["v"][nul].sum()) <= 1e-9 * host["v"][nul].sum()
        first_rows = {k: r for r, k in reversed(list(enumerate(host["k"].tolist())))}
        assert [first_rows[k] for k in gk.tolist()] == sorted(first_rows.values())  # first-occurrence order, null group included
    finally:
        eng.tune(flags=0)


@pytest.mark.parametrize("distinct", [40, 3000, 150_000, 390_000])
def test_sparse_keys_through_hash_partitioned_planes(eng, distinct):
    """Round 3 (rfx_group_plane.hip: k_plane_scatter<..., HK> + k_plane_hash_aggregate): sparse keys partitioned by hash into {key, value,
    meta} planes, every partition aggregated in LDS open-addressed tables by 1 / 2 / 4 workgroups (each keeps its share of the hash bits).
    Forced from 2^16 rows on (RFX_TUNE_CHUNK_SMALL); the context's counters say which queries took it: one value column (+ count / first),
    at most one predicate column beside key and value; the others (two value columns, two predicate columns, expressions) keep round 1's
    kernels.  Same answers (the oracle) either way; null keys form ONE group (the flat ABI's rule) through the device-wide table."""
    n = 400_003
    host = table(n, keys=distinct, nulls=True)
    host["k"] = host["k"] * 1_000_003 - 77_777
    planes = [{"by": "k", "s": ("sum", "v")},
              {"by": "k", "s": ("sum", "v"), "c": ("count", "a")},
              {"by": "k", "mx": ("max", "a"), "f": ("first", "a"), "mn": ("min", "a")},
              {"by": "k", "av": ("avg", "w")},
              {"where": ("<", "a", 400_000), "by": "k", "av": ("avg", "v"), "c": ("count", "v")},
              {"where": (">", "v", 0.9), "by": "k", "s": ("sum", "v")},
              {"by": "k", "s": ("sum", "k"), "c": ("count", "k")}]
    others = [{"where": ("and", ("<", "a", 100_000), (">", "w", -0.3)), "by": "k", "s": ("sum", "v")},
              {"by": "k", "s1": ("sum", "v"), "s2": ("sum", "w")},
              {"where": ("<", "a", 400_000), "by": "k", "s": ("sum", ("*", "v", "a"))}]
    try:
        eng.tune(flags=CHUNK_SMALL)
        before = (eng.stat(0), eng.stat(1), eng.stat(2))
        for q in planes:
            check_select(eng, host, q)
        after = (eng.stat(0), eng.stat(1), eng.stat(2))
        took = len(planes) if distinct >= 100_000 else 0  # (a few thousand keys or fewer: the sampled estimate keeps them on round 1's LDS kernels)
        assert after[0] - before[0] == took and after[2] - before[2] == took and after[1] == before[1], (before, after)
        for q in others:
            check_select(eng, host, q)
        assert (eng.stat(0), eng.stat(2)) == (after[0], after[2])
        eng.tune(flags=CHUNK_SMALL | NO_PLANE)
        check_select(eng, host, planes[1])
        assert eng.stat(0) == after[0]
        eng.tune(flags=CHUNK_SMALL)
        host = table(n, keys=3000)
        host["k"] = host["k"] * 1_000_003 - 77_777
        nul = rfo.gen_i64(n, 321, 97) == 0
        host["k"][nul] = NULL
        got = eng.select({"from": dev(eng, host), "by": "k", "c": ("count", "a"), "s": ("sum", "v")})
        gk = got["k"].cpu().numpy()
        assert (gk == NULL).sum() == 1 and len(gk) == len(np.unique(host["k"]))
        i = int(np.nonzero(gk == NULL)[0][0])
        assert int(got["c"][i]) == int(nul.sum()) and abs(float(got["s"][i]) - host["v"][nul].sum()) <= 1e-9 * host["v"][nul].sum()
        first_rows = {k: r for r, k in reversed(list(enumerate(host["k"].tolist())))}
        assert [first_rows[k] for k in gk.tolist()] == sorted(first_rows.values())
        # one key takes a third of the rows: its partition's regions overflow, the launch gives up (counter 1), round 1's kernels answer
        host = table(n, keys=50_000)
        host["k"] = host["k"] * 1_000_003 + 9
        host["k"][rfo.gen_i64(n, 55, 3) == 0] = 424_242_424_242
        fb = eng.stat(1)
        check_select(eng, host, {"by": "k", "s": ("sum", "v")})
        assert eng.stat(1) == fb + 1
    finally:
        eng.tune(flags=0)


def test_sparse_keys_through_hash_partitioned_planes_default_threshold(eng):
    """5e6 rows, 1e6 sparse keys: the route at its default row threshold (2^22), two workgroups per partition."""
    n = 5_000_011
    host = table(n, keys=1_000_000)
    host["k"] = host["k"] * 999_983 + 12_345
    before = eng.stat(0)
    check_select(eng, host, {"by": "k", "s": ("sum", "v")})
    check_select(eng, host, {"where": ("<", "a", 500_000), "by": "k", "s": ("sum", "v"), "c": ("count", "v")})
    assert eng.stat(0) - before == 2


def test_hash_primitives_pinned(eng):
    import ctypes as C
    from rayforce_amd import _lib as L
    x = rfo.gen_i64(4096, 11, 2**62) - 2**61
    d = eng.column(x)
    out = eng.empty(len(x))
    L.check(eng.lib.rfx_hip_hash_fnv1a_i64(eng._ctx, d.data_ptr(), len(x), out.data_ptr()))
    want = np.array([rfo.lib().rfo_hash_fnv1a(int(v)) for v in x], np.uint64)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    L.check(eng.lib.rfx_hip_hash_mix_u64(eng._ctx, d.data_ptr(), len(x), 0x9ddfea08eb382d69, out.data_ptr()))
    want = np.array([rfo.lib().rfo_hash_index_u64(0x9ddfea08eb382d69, int(v) & (2**64 - 1)) for v in x], np.uint64)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("n, flags", [(5003, 0), (64_001, 0), (300_000, 2)])
def test_insert_pass_says_every_rows_slot(eng, n, flags):
    """rfx_hip_group_hash_accumulate_slots: when the rows go straight into the device-wide table (small inputs; RFX_TUNE_NO_PARTITION; about as many groups as
    rows) the insert pass leaves every row's slot -- the same slots a probe of the finished table finds (-1 for rows the filter drops), and
    rfx_hip_hash_slot_first turns them into the groups' first rows: what the row-hash route needs without probing all rows again."""
    import ctypes as C
    from rayforce_amd import _lib as L
    k = rfo.gen_i64(n, 31, max(8, n // 2)) * 1_000_003 - 99
    k[7::113] = NULL
    a = rfo.gen_i64(n, 32, 1000)
    dk, da = eng.column(k), eng.column(a)
    cap = 1 << int(np.ceil(np.log2(2 * n)))
    arr = [torch.empty(cap + 1, dtype=torch.int64, device=dk.device) for _ in range(4)]  # keys, first, acc, cnt
    ht = L.HashTables()
    ht.capacity, ht.nagg, ht.d_keys, ht.d_first = cap, 1, arr[0].data_ptr(), arr[1].data_ptr()
    ht.d_acc[0], ht.d_cnt[0] = arr[2].data_ptr(), arr[3].data_ptr()
    agg = L.Agg()
    agg.d_col, agg.col_type, agg.kind = da.data_ptr(), L.RFX_I64, L.RFX_AGG_SUM
    pred = L.Pred()
    pred.d_col, pred.col_type, pred.rhs_type, pred.op, pred.rhs_i = da.data_ptr(), L.RFX_I64, L.RFX_I64, L.RFX_LT, 600
    eng.tune(flags=flags)
    try:
        L.check(eng.lib.rfx_hip_hash_tables_init(eng._ctx, C.byref(agg), C.byref(ht)), "init")
        slots = torch.full((n,), 12345, dtype=torch.int64, device=dk.device)
        rec = C.c_int(-1)
        L.check(eng.lib.rfx_hip_group_hash_accumulate_slots(eng._ctx, dk.data_ptr(), C.byref(pred), 1, L.RFX_AND, C.byref(agg), n, 0, C.byref(ht), slots.data_ptr(), C.byref(rec)),
                "accumulate_slots")
        eng.sync()
        if n < 65_536 or flags == 2:
            assert rec.value == 1  # (the partitioned forms decline: the device-wide kernel ran)
        if rec.value == 1:
            ids, pslots, mine = eng.empty(n), eng.empty(n), eng.empty(n)
            L.check(eng.lib.rfx_hip_join_probe_hash_slots(eng._ctx, dk.data_ptr(), n, C.byref(ht), ids.data_ptr(), pslots.data_ptr()), "probe")
            L.check(eng.lib.rfx_hip_hash_slot_first(eng._ctx, C.byref(ht), slots.data_ptr(), n, mine.data_ptr()), "slot_first")
            eng.sync()
            sel = torch.from_numpy(a < 600).to(dk.device)
            assert torch.equal(slots[sel], pslots[sel]) and torch.equal(mine[sel], ids[sel])
            assert bool((slots[~sel] == -1).all()) and bool((mine[~sel] == NULL).all())
            first = mine[sel].cpu().numpy()
            want_first = {}
            for i in np.flatnonzero(a < 600):
                want_first.setdefault(int(k[i]), int(i))
            assert np.array_equal(first, np.array([want_first[int(x)] for x in k[a < 600]], np.int64))
    finally:
        eng.tune(flags=0)


# ---------------------------------------------------------------- row-sharded driver on one rank (collectives are no-ops)
def test_sharded_engine_single_rank(eng):
    from rayforce_amd.dist import ShardedEngine
    n = 150_001
    host = table(n, keys=3000, nulls=True)
    d = dev(eng, host)
    sh = ShardedEngine(eng, n)
    assert sh.shard.row0 == 0 and sh.shard.total_rows == n
    aggs = [("sum", "a"), ("avg", "v"), ("min", "w"), ("max", "a"), ("count", "a")]
    vals, sel = sh.filter_aggr(aggs, ("<", "a", 400_000), d)
    want = rfo.select({"from": host, "where": ("<", "a", 400_000), **{f"o{i}": a for i, a in enumerate(aggs)}})
    for i, v in enumerate(vals):
        w = want[f"o{i}"][0]
        assert (abs(v - w) <= RTOL * abs(w)) if isinstance(v, float) else v == int(w)
    assert sel == int(rfo.mask_of(("<", "a", 400_000), host).sum())
    r = sh.group_by("k", [("sum", "v"), ("count", "a")], None, d)
    w = rfo.select({"from": host, "by": "k", "s": ("sum", "v"), "c": ("count", "a")})
    assert np.array_equal(r["keys"].cpu().numpy(), w["k"]) and np.array_equal(r["results"][1].cpu().numpy(), w["c"])
    same_f64(r["results"][0].cpu().numpy(), w["s"])
    assert np.array_equal(sh.where(("<", "a", 1000), d).cpu().numpy(), rfo.where(rfo.mask_of(("<", "a", 1000), host)))


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 4, 6, 8, 16, 32, 128, 144, 160, 256, 384, 1024, 1028, 2048, 2064, 4096, 4100, 16384, 32768, 32772])
def test_group_by_every_code_path_agrees(eng, flags):
    """RFX_TUNE_* force the LDS-table / partitioned (fused and unfused scope) / device-atomic paths: same answers."""
    n = 400_003
    try:
        eng.tune(flags=flags)
        for keys in (900, 2000, 7000, 60_000, 300_000):  # 64 KB LDS / 160 KB LDS / LDS in several passes / partitioned / partitioned
            host = table(n, keys=keys, nulls=True)
            check_select(eng, host, {"by": "k", "s": ("sum", "v"), "si": ("sum", "a"), "c": ("count", "a"), "mx": ("max", "w"), "av": ("avg", "a")})
            check_select(eng, host, {"where": ("<", "a", 300_000), "by": "k", "s": ("sum", "v"), "mn": ("min", "a")})
    finally:
        eng.tune(flags=0)


@pytest.mark.parametrize("flags", [0, 128, 256, 2048, 32768])
@pytest.mark.parametrize("thr", [1_000, 100_000, 480_000, 520_000, 990_000])
def test_filtered_partitioned_group_by(eng, flags, thr):
    """Partitioned path under a filter: <= 50 % selected -> compact (bitmap, ordered compaction of key / value planes / row
    ids) then the unfiltered pipeline with `first` translated back (1..3 value planes; with and without the fused scope count;
    288 k survivors = fewer tiles than workgroups); above 50 % the write-combining scatter with predicates; tiny selections
    fall back to device atomics."""
    n = 600_011
    host = table(n, keys=200_000, nulls=True)
    try:
        eng.tune(flags=flags)
        w = ("<", "a", thr)
        check_select(eng, host, {"where": w, "by": "k", "s": ("sum", "v")})
        check_select(eng, host, {"where": w, "by": "k", "s": ("sum", "v"), "mn": ("min", "a"), "c": ("count", "a")})
        check_select(eng, host, {"where": ("and", w, (">", "v", 0.2)), "by": "k", "s": ("sum", "v"), "mx": ("max", "a"), "av": ("avg", "w"), "f": ("first", "a")})
    finally:
        eng.tune(flags=0)


CHUNK_SMALL = 32768  # RFX_TUNE_CHUNK_SMALL: the one-pass chunk partitioning from 2^16 rows on


CHUNK_QUEUE, CHUNK_BINS = 131072, 262144  # selective filters: always the sorted-queue kernel / always per-partition bins


NO_PLANE = 1048576  # RFX_TUNE_NO_PLANE: round 2's 16-byte records in chunks, never the 8 + 4-byte planes


@pytest.mark.parametrize("flags", [CHUNK_SMALL, CHUNK_SMALL | 4, CHUNK_SMALL | NO_PLANE, CHUNK_SMALL | CHUNK_QUEUE, CHUNK_SMALL | CHUNK_BINS, 16384])
@pytest.mark.parametrize("shape", ["uniform", "offset", "skew", "wide", "narrow"])
def test_chunk_partitioned_group_by(eng, flags, shape):
    """rfx_hip_group_scope: the scope pass that also radix-partitions (rfx_group_chunk.hip) -- chunk allocation, slab switches,
    one partition taking every row (several chunks per tile), negative keys / key >> 8 wrap, filtered and unfiltered, every
    single-column aggregate set; flag 16384 (RFX_TUNE_NO_CHUNK) is the column-pass form of the same queries.  Selective filters take
    per-partition bins when the sampled keys spread and the sorted queue otherwise ("skew"): both kernels forced on every shape
    (bins under skew: a drain every 32 survivors)."""
    n = 700_001
    host = table(n, keys=200_000, nulls=True)
    if shape == "offset":
        host["k"] = host["k"] - 70_000  # negative keys: (key >> 8) is an arithmetic shift, the header keeps it mod 2^32
    elif shape == "skew":
        host["k"] = (host["k"] % 900) * 256 + 7  # every key in partition 7
    elif shape == "wide":
        host["k"] = rfo.gen_i64(n, 41, 600_000) + (1 << 40)
    elif shape == "narrow":
        host["k"] = rfo.gen_i64(n, 42, 20_000) * 3 - 1
    try:
        eng.tune(flags=flags)
        check_select(eng, host, {"by": "k", "s": ("sum", "v")})
        check_select(eng, host, {"by": "k", "s": ("sum", "v"), "c": ("count", "v"), "mn": ("min", "v"), "av": ("avg", "v"), "f": ("first", "v")})
        check_select(eng, host, {"by": "k", "si": ("sum", "a"), "mx": ("max", "a")})
        check_select(eng, host, {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")})
        check_select(eng, host, {"where": ("and", ("<", "a", 600_000), (">", "w", -0.25)), "by": "k", "s": ("sum", "v"), "c": ("count", "a")})
        check_select(eng, host, {"where": ("<", "a", -5), "by": "k", "s": ("sum", "v")})  # nothing (or only nulls) selected
        # the selective kernel's instantiations: 1 / <= 3 / <= 8 predicates, predicate on the key and on the value, key == value column
        check_select(eng, host, {"where": ("and", ("<", "a", 300_000), (">", "w", -0.4), ("<", "v", 0.7)), "by": "k", "mx": ("max", "v")})
        check_select(eng, host, {"where": ("and", ("<", "a", 400_000), (">", "w", -0.4), ("<", "v", 0.9), (">=", "a", 5), ("!=", "k", 77)), "by": "k", "s": ("sum", "v")})
        check_select(eng, host, {"where": ("<", "v", 0.25), "by": "k", "s": ("sum", "v"), "c": ("count", "v")})
        check_select(eng, host, {"where": ("<", "a", 250_000), "by": "k", "s": ("sum", "k"), "mn": ("min", "k")})
        check_select(eng, host, {"where": ("or", ("<", "a", 50_000), (">", "a", 950_000)), "by": "k", "s": ("sum", "v")})
    finally:
        eng.tune(flags=0)


def test_chunk_partitioned_group_by_larger(eng):
    """5e6 rows: more than one slab per workgroup, uneven last workgroup, the default row threshold (2^22)."""
    n = 5_000_011
    host = table(n, keys=1_000_000)
    check_select(eng, host, {"by": "k", "s": ("sum", "v")})
    check_select(eng, host, {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v"), "c": ("count", "v")})


def test_plane_partitioned_group_by_takes_the_plane_kernels(eng):
    """Spread keys go through k_plane_scatter / k_plane_aggregate (rfx_group_plane.hip): the context's path counters say so; RFX_TUNE_NO_PLANE
    and a skewed low key byte keep to the chunk kernels.  Same answers (the oracle) either way."""
    n = 700_001
    host = table(n, keys=50_000, nulls=True)  # (50 000 keys: the 10 % selection below still has more rows than slots -- the dense arm)
    qs = [{"by": "k", "s": ("sum", "v")}, {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v"), "c": ("count", "v"), "f": ("first", "v")}]
    try:
        eng.tune(flags=CHUNK_SMALL)
        before = [eng.stat(i) for i in range(5)]
        for q in qs:
            check_select(eng, host, q)
        after = [eng.stat(i) for i in range(5)]
        assert after[0] - before[0] == 2 and after[2] - before[2] == 2 and after[1] == before[1], (before, after)
        assert after[3] == before[3] and after[4] == before[4], (before, after)
        eng.tune(flags=CHUNK_SMALL | NO_PLANE)
        before = after
        for q in qs:
            check_select(eng, host, q)
        after = [eng.stat(i) for i in range(5)]
        assert after[0] == before[0] and after[3] - before[3] == 2 and after[4] - before[4] == 2, (before, after)
    finally:
        eng.tune(flags=0)


def test_remembered_sample_does_not_outlive_its_use(eng):
    """rfx_chunk_scope remembers the sample of a repeated query (key range, selectivity).  A column rewritten IN PLACE behind the context's
    back (a torch tensor; the operator layer's uploads drop the memory themselves) makes it wrong: the filter now keeps every row instead
    of 1 %, and a sample that still says "1 %" routes the query elsewhere (few rows over a wide range: the hashed tables) -- slower, never
    wrong.  A remembered sample serves 32 queries: every answer in between equals the oracle's, and within 34 queries the planes answer
    again (RFX_STAT_PLANE_SCATTER moves)."""
    n = 700_001
    host = table(n, keys=50_000)
    q = {"where": ("<", "a", 10_000), "by": "k", "s": ("sum", "v"), "c": ("count", "a")}
    try:
        eng.tune(flags=CHUNK_SMALL)
        d = dev(eng, host)
        check_select(eng, host, q, d)
        check_select(eng, host, q, d)  # (the remembered sample)
        d["a"] //= 100  # in place: a < 10 000 now keeps every row
        host2 = dict(host, a=host["a"] // 100)
        s0 = eng.stat(0)
        back = 0
        for i in range(34):
            check_select(eng, host2, q, d)
            if eng.stat(0) > s0:
                back = i + 1
                break
        assert 0 < back <= 34, back
    finally:
        eng.tune(flags=0)


@pytest.mark.parametrize("run", [64, 4096])
def test_plane_rings_under_pressure_and_region_overflow(eng, run):
    """Keys whose low byte comes in runs: the strided sample sees the 256 partitions evenly filled, a wave step does not.  run = 64: eight
    partitions take a whole 512-row step (every ring half is waited for, the slow path of k_plane_scatter), regions still fill evenly;
    run = 4096: a row block feeds 64 partitions only, their regions overflow, the launch gives up (counter 1) and the chunk kernels answer."""
    n = 900_001
    i = np.arange(n, dtype=np.int64)
    host = table(n, keys=10)
    host["k"] = ((i // run) % 256) + 256 * rfo.gen_i64(n, 77, 700)
    try:
        eng.tune(flags=CHUNK_SMALL)
        before = [eng.stat(j) for j in range(5)]
        check_select(eng, host, {"by": "k", "s": ("sum", "v"), "mn": ("min", "v")})
        check_select(eng, host, {"where": ("<", "a", 500_000), "by": "k", "s": ("sum", "v"), "f": ("first", "v")})
        after = [eng.stat(j) for j in range(5)]
        assert after[0] - before[0] == 2, (before, after)
        if run == 64:
            assert after[1] == before[1] and after[2] - before[2] == 2, (before, after)
        else:
            assert after[1] - before[1] == 2 and after[2] == before[2], (before, after)
    finally:
        eng.tune(flags=0)


def test_plane_partitioned_group_by_several_value_columns(eng):
    """Two and three value columns (the H2O `sum v1, v2, v3` / `avg` shapes over many keys) ride as separate 8-byte planes through ONE
    scatter; the aggregate pass runs once per set of aggregates whose tables fit a CU's LDS.  Counters: one scatter per query, at least
    one aggregate launch, no chunk kernels; a null key anywhere hands the query back to the exact scope pass."""
    n = 900_001
    host = table(n, keys=60_000, nulls=True)
    host["u"] = rfo.gen_f64(n, 77) * 10.0
    qs = [{"by": "k", "s1": ("sum", "v"), "s2": ("sum", "w"), "s3": ("sum", "u")},
          {"by": "k", "a1": ("avg", "v"), "a2": ("avg", "w"), "a3": ("avg", "u"), "c": ("count", "v")},
          {"where": ("<", "a", 600_000), "by": "k", "s": ("sum", "v"), "mx": ("max", "a"), "mn": ("min", "w"), "f": ("first", "u")},
          {"by": "k", "si": ("sum", "a"), "sf": ("sum", "v")}]
    try:
        eng.tune(flags=CHUNK_SMALL)
        st = lambda: [eng.stat(i) for i in range(5)]
        for q in qs:
            b0 = st()
            check_select(eng, host, q)
            b1 = st()
            assert b1[0] - b0[0] == 1 and b1[2] - b0[2] >= 1 and b1[1] == b0[1] and b1[3] == b0[3], (q, b0, b1)
        host["k"][12345] = NULL
        b0 = st()
        check_select(eng, host, qs[0])
        assert st()[2] == b0[2]
    finally:
        eng.tune(flags=0)


# ---------------------------------------------------------------- several `by:` columns (composite key)
def mk_table(n, mods, offs, seed=31):
    t = {f"k{j + 1}": rfo.gen_i64(n, seed + 10 * j, m) + o for j, (m, o) in enumerate(zip(mods, offs))}
    t.update(v=rfo.gen_f64(n, seed + 5), a=rfo.gen_i64(n, seed + 6, 1_000_000))
    return t


@pytest.mark.parametrize("n,mods,offs", [(1, (3, 3), (0, 0)), (1000, (7, 13), (0, 100)), (300_007, (100, 100), (-5, 10**12)),
                                         (300_007, (40, 30, 20, 4), (0, 1, 2, 3)), (700_001, (1000, 1000), (0, 0)),
                                         (50_000, (3000, 4000), (0, 0))])
def test_group_by_several_keys(eng, n, mods, offs):
    """Dense composite (LDS tables / partitioned / atomics by range) and the sparse arm (range > rows).  Without `where:`
    this is pinned by the golden multikey cases; with `where:` the oracle restates the evident intent (filter, then group)
    because the reference's own result for that combination is defective."""
    host = mk_table(n, mods, offs)
    by = {f"g{j}": f"k{j + 1}" for j in range(len(mods))}
    check_select(eng, host, {"by": by, "s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", "a"), "av": ("avg", "v")})
    check_select(eng, host, {"where": ("and", ("<", "a", 400_000), (">", "v", 0.125)), "by": by, "s": ("sum", "v"), "f": ("first", "a")})
    check_select(eng, host, {"where": ("<", "a", -1), "by": by, "s": ("sum", "v")})  # nothing selected


@pytest.mark.parametrize("flags", [0, 1, 4, 64, 66])
def test_group_by_several_keys_every_path(eng, flags):
    """Keys folded on the fly in the LDS-table kernels (small / 160 KB) vs the materialised composite column feeding the
    partitioned path vs on-the-fly into device atomics (64+2): same answers."""
    try:
        eng.tune(flags=flags)
        for mods in ((20, 30), (60, 50), (300, 400)):
            host = mk_table(400_003, mods, (3, -7))
            by = {"x": "k1", "y": "k2"}
            check_select(eng, host, {"by": by, "s": ("sum", "v"), "c": ("count", "a"), "mn": ("min", "a")})
            check_select(eng, host, {"where": (">", "v", 0.3), "by": by, "s": ("sum", "v")})
    finally:
        eng.tune(flags=0)


def test_group_by_several_keys_overflow_takes_the_row_hash_path(eng):
    """Product of ranges beyond i64 (a null key always is): the reference leaves its perfect path for the row-hash one; so
    does the Engine (the flat composite entry points still say RFX_ELIMIT: rfx_composite_plan)."""
    from rayforce_amd import _lib as L
    import ctypes as C
    host = mk_table(1000, (10, 10), (0, 0))
    host["k2"][7] = NULL
    check_select(eng, host, {"by": {"x": "k1", "y": "k2"}, "s": ("sum", "v")})
    mults, tmax = (C.c_int64 * 2)(), C.c_int64()
    assert eng.lib.rfx_composite_plan((C.c_int64 * 2)(0, NULL), (C.c_int64 * 2)(9, 9), 2, mults, C.byref(tmax)) == L.RFX_ELIMIT
    # one key through the dict spelling is the plain single-key path (null key -> hashed), core/index.c:2741-2742
    check_select(eng, host, {"by": {"y": "k2"}, "s": ("sum", "v")})


def test_row_hash_path_with_more_columns_than_one_launch_reads(eng):
    """The row-hash route with eight aggregates over eight distinct columns beside a filter column: more than one launch of the insert kernel reads
    (RFX_MAX_COLS) -- the unpacked insert splits into two passes over the same slots; the PACKED insert (RFX_EMIT_BY_ROWS=2 takes it at this size) declines
    (RFX_ESTATE) and the planner lays the same block out field by field and goes on (ph_pass)."""
    n = 40_009
    host = mk_table(n, (10, 10), (0, 0))
    host["k2"][7::131] = NULL  # (a null key: the row-hash path)
    for j in range(8):
        host[f"c{j}"] = rfo.gen_f64(n, 90 + j) if j % 2 else rfo.gen_i64(n, 90 + j, 1000)
    q = {"by": {"x": "k1", "y": "k2"}, "where": ("<", "a", 700_000)}
    for j, fn in enumerate(("sum", "max", "sum", "min", "count", "avg", "max", "sum")):
        q[f"o{j}"] = (fn, f"c{j}")
    check_select(eng, host, q)


def test_sharded_several_keys_single_rank(eng):
    from rayforce_amd.dist import ShardedEngine
    n = 200_003
    host = mk_table(n, (50, 60), (5, -5))
    sh = ShardedEngine(eng, n)
    r = sh.group_by(["k1", "k2"], [("sum", "v"), ("count", "a")], ("<", "a", 600_000), dev(eng, host))
    w = rfo.select({"from": host, "where": ("<", "a", 600_000), "by": {"k1": "k1", "k2": "k2"}, "s": ("sum", "v"), "c": ("count", "a")})
    assert np.array_equal(r["key_columns"][0].cpu().numpy(), w["k1"]) and np.array_equal(r["key_columns"][1].cpu().numpy(), w["k2"])
    assert np.array_equal(r["results"][1].cpu().numpy(), w["c"])
    same_f64(r["results"][0].cpu().numpy(), w["s"])


# ---------------------------------------------------------------- element-wise arithmetic feeding aggregates
def test_expression_special_values(eng):
    """+ - * div on nulls / NaN / +-inf / -0.0 / zero divisors / wrap-around, every type shape: sum, min and max of the
    expression over ONE selected row at a time reproduce the element-wise truth table (bit-exact for i64; f64 `div` by an atom
    may differ by 1 ulp from the reference build, which multiplies by the reciprocal)."""
    import golden_cases as G
    for op, tag, l, r, want in G.binop_cases():
        n = len(want)
        t = {"row": np.arange(n, dtype=np.int64)}
        lhs = "l" if isinstance(l, np.ndarray) else l
        rhs = "r" if isinstance(r, np.ndarray) else r
        if isinstance(l, np.ndarray):
            t["l"] = l
        if isinstance(r, np.ndarray):
            t["r"] = r
        d = dev(eng, t)
        for i in range(n):
            got = eng.select({"from": d, "where": ("==", "row", i), "mx": ("max", (op, lhs, rhs)), "s": ("sum", (op, lhs, rhs))})
            w = want[i]
            if want.dtype == np.float64:
                g = float(got["mx"][0])
                if np.isnan(w):
                    assert np.isnan(g) and float(got["s"][0]) == 0.0, (op, tag, i)
                else:
                    assert g == w or abs(g - w) <= 2.3e-16 * abs(w), (op, tag, i, g, w)
            else:
                assert int(got["mx"][0]) == int(w), (op, tag, i)
                assert int(got["s"][0]) == (0 if int(w) == NULL else int(w)), (op, tag, i)


@pytest.mark.parametrize("n", [1, 1000, 300_007])
@pytest.mark.parametrize("keys", [7, 3000, 150_000])
def test_expression_aggregates(eng, n, keys):
    host = table(n, keys=keys, nulls=True)
    host["b"] = rfo.gen_i64(n, 77, 9) - 1
    q = {"s1": ("sum", ("*", "a", "v")), "s2": ("sum", ("*", "a", "b")), "av": ("avg", ("-", "a", "b")), "mx": ("max", ("*", "v", "w")),
         "mn": ("min", ("-", 100, "a")), "s4": ("sum", ("div", "a", "b")), "s5": ("sum", ("+", "v", 2))}
    check_select(eng, host, q)
    check_select(eng, host, {**q, "where": ("and", ("<", "b", 5), (">", "v", 0.1))})
    check_select(eng, host, {**q, "where": ("or", ("<", "b", 1), ("and", (">", "v", 0.5), ("!=", "k", 3)))})  # nested tree
    check_select(eng, host, {**q, "by": "k"})
    check_select(eng, host, {**q, "by": "k", "where": ("<", "b", 5)})
    check_select(eng, host, {"plain": ("sum", "v"), "x": ("sum", ("*", "v", "w")), "c": ("count", "a"), "by": "k"})  # mixed plain / expression


@pytest.mark.parametrize("flags", [0, 4096, 8192])
@pytest.mark.parametrize("keys", [2, 3, 5, 600, 40_000])
def test_nested_expression_trees_by_group(eng, flags, keys):
    """Expression TREES under by: -- evaluated inside the LDS-table pass (flags 0, tables <= 64 KB) or materialised by k_derive
    (flags 4096, and every larger table): same answers, nulls / NaN in the operands included; wide plans (7 distinct columns,
    8 outputs, two key columns) stay in one launch.  Two to four groups: lane-private table replicas (off with flags 8192)."""
    n = 250_003
    host = table(n, keys=keys, nulls=True)
    host["b"] = rfo.gen_i64(n, 77, 9) - 1
    host["g"] = rfo.gen_i64(n, 78, 3)
    q = {"s1": ("sum", ("*", "v", ("-", 1, "w"))), "s2": ("sum", ("*", ("*", "v", ("-", 1, "w")), ("+", 1, "w"))), "s3": ("sum", ("+", ("*", "a", "b"), "b")),
         "mx": ("max", ("-", ("*", "a", 2), "b")), "av": ("avg", ("*", ("+", "a", 1), "v")), "p": ("sum", "v"), "c": ("count", "a"), "mn": ("min", ("div", ("+", "a", "b"), 3))}
    try:
        eng.tune(flags=flags)
        check_select(eng, host, {**q, "by": "k"})
        check_select(eng, host, {**q, "by": "k", "where": ("and", ("<", "b", 6), (">", "v", 0.05))})
        if keys <= 600:
            check_select(eng, host, {**q, "by": {"g": "g", "k": "k"}})
        if keys <= 3:  # plain aggregates over a handful of groups take the replicated form too
            check_select(eng, host, {"by": "k", "s": ("sum", "v"), "si": ("sum", "a"), "av": ("avg", "w"), "mn": ("min", "a"), "mx": ("max", "v"), "c": ("count", "a"), "f": ("first", "a")})
            check_select(eng, host, {"by": "k", "where": (">", "v", 0.5), "av": ("avg", "a"), "s": ("sum", "w")})
    finally:
        eng.tune(flags=0)


def test_expressions_as_columns_and_inside_predicates(eng):
    """(op x y) and expression trees as device columns (ray_add / sub / mul / div over vectors and atoms, one pass whatever the
    depth), and as operands of `where:` comparisons -- flat, nested, under by:, against a column or an atom, nulls / NaN included."""
    n = 300_007
    host = table(n, keys=700, nulls=True)
    host["b"] = rfo.gen_i64(n, 77, 9) - 1
    d = dev(eng, host)
    for e in (("*", "a", "v"), ("-", 100, "a"), ("div", "a", "b"), ("+", ("*", "a", "b"), "b"), ("*", ("*", "v", ("-", 1, "w")), ("+", 1, "w")),
              ("div", ("+", "a", "b"), 3), ("-", ("*", "a", 2), "b"), ("*", "w", 2.0)):
        got, want = eng.eval_expr(e, d).cpu().numpy(), rfo.eval_arg(e, host)
        assert got.dtype == want.dtype, e
        if want.dtype == np.float64:
            same_f64(got, want)
        else:
            assert np.array_equal(got, want), e
    q = {"s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", ("*", "a", "v"))}
    for w in ((">", ("*", "a", "v"), 250_000.0), ("<=", ("-", "a", ("*", "b", 100_000)), "a"), ("and", ("<", ("+", "v", "w"), 0.6), (">", "a", 1000)),
              ("or", ("==", ("div", "a", 1000), 7), ("and", (">", ("*", "v", 2.0), 1.5), ("!=", "b", 3))), ("<", "v", ("*", "w", 3.0))):
        check_select(eng, host, {**q, "where": w})
        check_select(eng, host, {**q, "where": w, "by": "k"})  # (a nested tree under by: goes masks -> ids -> gather -> group, as the reference)
        ids = eng.where(w, d).cpu().numpy()
        assert np.array_equal(ids, rfo.where(rfo.mask_of(w, host)))


def test_expression_aggregates_refusals(eng):
    from rayforce_amd._lib import RfxError
    d = dev(eng, table(100))
    with pytest.raises(RfxError, match="count of an expression"):
        eng.select({"from": d, "c": ("count", ("*", "a", "v"))})
    with pytest.raises(RfxError, match="first of an expression"):
        eng.select({"from": d, "f": ("first", ("*", "a", "v")), "by": "k"})
    with pytest.raises(RfxError, match="unsupported expression"):
        eng.select({"from": d, "s": ("sum", ("xor", "a", "k"))})  # (`/` and `%` are covered since round 3: tests/test_gpu_golden.py)


def test_group_by_key_tuples_beyond_the_composite_key(eng):
    """Row-hash path (index_group_list, core/index.c:2731-2790): six key columns whose ranges multiply beyond 64 bits (H2O Q7
    shape), null keys, `where:`, expression aggregates, more outputs than one launch carries; both group orders."""
    n = 300_007
    host = table(n, keys=100, nulls=True)
    host.update({"id1": rfo.gen_i64(n, 81, 100) * 1_000_000_007, "id2": rfo.gen_i64(n, 82, 100) - 50, "id3": rfo.gen_i64(n, 83, 1000) * (1 << 40),
                 "id4": rfo.gen_i64(n, 84, 100), "id5": rfo.gen_i64(n, 85, 100) * 3, "id6": rfo.gen_i64(n, 86, 1000) * (1 << 33)})
    by6 = {f"id{i}": f"id{i}" for i in range(1, 7)}
    check_select(eng, host, {"by": by6, "s": ("sum", "v"), "c": ("count", "a")})  # H2O Q7
    check_select(eng, host, {"by": by6, "order": "radix", "s": ("sum", "w"), "mx": ("max", "a")})
    two = {"id1": "id1", "id3": "id3"}
    q = {"by": two, "s": ("sum", "v"), "si": ("sum", "a"), "av": ("avg", "w"), "mn": ("min", "a"), "mx": ("max", "v"), "c": ("count", "a"),
         "x": ("sum", ("*", "v", ("-", 1, "w"))), "f": ("first", "a"), "av2": ("avg", "a")}
    check_select(eng, host, q)
    check_select(eng, host, {**q, "where": ("and", (">", "v", 0.3), ("<", "a", 700_000))})
    host["id2"][rfo.gen_i64(n, 87, 40) == 0] = NULL  # null keys are keys like any other here
    host["id4"][rfo.gen_i64(n, 88, 30) == 0] = NULL
    check_select(eng, host, {"by": {"id2": "id2", "id4": "id4"}, "s": ("sum", "v"), "c": ("count", "a")})
    check_select(eng, host, {"by": {"id2": "id2", "id4": "id4"}, "order": "radix", "where": (">", "v", 0.5), "s": ("sum", "v"), "c": ("count", "a")})
    check_select(eng, host, {"by": by6, "s": ("sum", "v")})


def test_equi_joins_at_size_and_edges(eng):
    """left_join / inner_join / join_index against the oracle: 1e6-row left side, dense and hashed right tables, three key
    columns (row hash + tuple check), f64 and i64 payloads, no match at all, every row matching, empty sides."""
    nl, nr = 1_000_003, 200_001
    for keys, mul in ((["k"], 1), (["k"], 1_000_003), (["k", "g", "h"], 1), (["k", "g", "h"], 1 << 45)):  # dense / hashed / composite key / row hash
        left = {"k": rfo.gen_i64(nl, 91, 150_000) * mul, "g": rfo.gen_i64(nl, 92, 3), "h": rfo.gen_i64(nl, 93, 2) - 1, "a": rfo.gen_i64(nl, 94, 10**6), "v": rfo.gen_f64(nl, 95)}
        right = {"k": rfo.gen_i64(nr, 96, 180_000) * mul, "g": rfo.gen_i64(nr, 97, 3), "h": rfo.gen_i64(nr, 98, 2) - 1, "v": rfo.gen_f64(nr, 99) + 5, "z": rfo.gen_i64(nr, 90, 77)}
        left["v"][::53] = np.nan
        right["z"][::31] = NULL
        dl, dr = dev(eng, left), dev(eng, right)
        assert np.array_equal(eng.join_index(keys, dl, dr).cpu().numpy(), rfo.join_index(keys, left, right))
        for fn in ("left_join", "inner_join"):
            got, want = getattr(eng, fn)(keys, dl, dr), getattr(rfo, fn)(keys, left, right)
            assert list(got) == list(want), fn
            for c in want:
                g = got[c].cpu().numpy()
                assert g.dtype == want[c].dtype and np.array_equal(g.view(np.int64), want[c].view(np.int64)), (fn, keys, c)  # bit-exact, NaN included
    # nothing matches / everything matches / empty sides
    left = {"k": np.arange(1000, dtype=np.int64), "v": rfo.gen_f64(1000, 1)}
    right = {"k": np.arange(5000, 6000, dtype=np.int64), "w": rfo.gen_f64(1000, 2)}
    dl, dr = dev(eng, left), dev(eng, right)
    assert (eng.join_index("k", dl, dr).cpu().numpy() == NULL).all() and eng.inner_join("k", dl, dr)["k"].numel() == 0
    assert np.isnan(eng.left_join("k", dl, dr)["w"].cpu().numpy()).all()
    same = eng.inner_join("k", dl, dl)
    assert np.array_equal(same["v"].cpu().numpy(), left["v"])
    empty = {"k": eng.empty(0), "w": eng.empty(0, torch.float64)}
    assert list(eng.left_join("k", dl, empty)) == ["k", "v"] and list(eng.inner_join("k", empty, dr)) == ["k", "w"]
    from rayforce_amd._lib import RfxError
    with pytest.raises(RfxError, match="i64-like"):
        eng.join_index("v", dl, dev(eng, {"v": left["v"]}))


def test_nested_where_tree_under_by(eng):
    """and / or trees of any depth with by: -- the reference's own plan (masks, where, gather, group) on the device: one key, two
    keys, bucketed key, key tuples on the row-hash path, `first`, expression aggregates."""
    n = 300_007
    host = table(n, keys=3000, nulls=True)
    host["b"] = rfo.gen_i64(n, 77, 9) - 1
    host["ts"] = rfo.gen_i64(n, 55, 10**6) - 5 * 10**5
    host["wide"] = rfo.gen_i64(n, 56, 40) * (1 << 58)
    w = ("or", ("and", ("<", "a", 300_000), (">", "v", 0.2)), ("and", ("==", "b", 3), ("or", ("<", "w", -0.3), (">=", "a", 900_000))))
    q = {"s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", ("*", "a", "v")), "f": ("first", "a"), "av": ("avg", "w")}
    for by in ("k", {"k": "k", "b": "b"}, {"t": ("xbar", "ts", 50_000)}, {"wide": "wide", "k": "k", "b": "b"}):
        check_select(eng, host, {**q, "where": w, "by": by})
    r = eng.group_by("k", [("sum", "v")], w, dev(eng, host))
    gi, fi, _, _ = rfo.group_index(host["k"], rfo.where(rfo.mask_of(w, host)))
    assert np.array_equal(r["first"].cpu().numpy(), rfo.where(rfo.mask_of(w, host))[fi])  # first = ORIGINAL row ids


def test_group_by_xbar_buckets(eng):
    """by: {b: (xbar a width)} -- bucketed keys (negative values, dense and sparse bucket ranges, one or two key columns, with
    where:).  Null keys are left out: they take the sparse path, where one null group here stands against one group per null
    row in the reference (DESIGN.md deviation 3)."""
    n = 300_007
    host = table(n, keys=5000)
    host["ts"] = rfo.gen_i64(n, 55, 10**9) - 5 * 10**8
    for by in ({"b": ("xbar", "ts", 60_000)}, {"b": ("xbar", "ts", 1_000_003)}, {"b": ("xbar", "a", 7)}, {"k": "k", "b": ("xbar", "a", 250_000)}):
        q = {"by": by, "s": ("sum", "v"), "c": ("count", "a"), "mx": ("max", "w")}
        try:
            check_select(eng, host, q)
            check_select(eng, host, {**q, "where": (">", "v", 0.4)})
        except rfo.NotPerfect:
            pass
    from rayforce_amd._lib import RfxError
    with pytest.raises(RfxError, match="width must be positive"):
        eng.select({"from": dev(eng, host), "by": {"b": ("xbar", "a", 0)}, "s": ("sum", "v")})


# ---------------------------------------------------------------- a handful of groups: one kernel per plan, compiled at run time
NO_RTC = 524288  # RFX_TUNE_NO_RTC


def _rtc_stats(eng):
    import ctypes as C
    a, b = C.c_int64(), C.c_int64()
    eng.lib.rfx_hip_rtc_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


@pytest.mark.parametrize("groups", [1, 2, 6, 8])
@pytest.mark.parametrize("n", [1, 700, 300_007])
def test_few_groups_run_time_compiled_kernels(eng, groups, n):
    """Group-bys over at most 8 slots take a kernel generated for the plan (rfx_group_few_rtc.hpp through hiprtc: per-lane register
    accumulators for every (aggregate, group), no atomics in the row loop): every aggregate kind over i64 / f64 with nulls and
    NaNs, expressions, 0 / 1 / 3 / 5 predicates, one and two key columns, a key offset -- against the oracle, and against the
    prebuilt LDS-table kernel (RFX_TUNE_NO_RTC) bit for bit on everything that is not an f64 sum."""
    host = table(n, seed=groups, keys=groups, nulls=True)
    host["k"] = host["k"] + 1000  # kmin != 0
    if n > 100:  # infinities: the masked-fma accumulation zeroes them out of its products and adds them on a side path
        host["w"][7::97] = np.inf
        host["w"][11::389] = -np.inf
        host["v"][13::211] = np.inf
    host["g2"] = rfo.gen_i64(n, 77, 2)
    host["k4"] = rfo.gen_i64(n, 78, max(1, groups // 2))
    queries = [
        {"by": "k", "s": ("sum", "v"), "c": ("count", "v"), "mn": ("min", "v"), "mx": ("max", "v"), "av": ("avg", "v"), "f": ("first", "v")},
        {"by": "k", "si": ("sum", "a"), "mi": ("min", "a"), "xa": ("max", "a"), "aa": ("avg", "a"), "fa": ("first", "a"), "ca": ("count", "a")},
        {"where": ("<", "a", 600_000), "by": "k", "s": ("sum", "w"), "mx": ("max", "a")},
        {"where": ("and", ("<", "a", 800_000), (">", "w", -0.4), ("<", "v", 0.9)), "by": "k", "s": ("sum", "v"), "av": ("avg", "a")},
        {"where": ("or", ("<", "a", 100_000), (">", "a", 900_000), ("<", "w", -0.45), (">", "v", 0.97), ("==", "k", 1000)), "by": "k", "c": ("count", "a")},
        {"by": "k", "e1": ("sum", ("*", "v", ("-", 1.0, "w"))), "e2": ("sum", ("*", ("*", "v", ("-", 1.0, "w")), ("+", 1.0, "v"))), "e3": ("max", ("+", "a", 5)), "c": ("count", "a")},
        {"where": ("<=", "a", 950_000), "by": {"x": "k4", "y": "g2"}, "s": ("sum", "v"), "q": ("sum", "a"), "av": ("avg", "w"), "c": ("count", "a")},
        {"where": ("<", "a", -1), "by": "k", "s": ("sum", "v")},  # nothing selected
    ]
    l0, c0 = _rtc_stats(eng)
    results = []
    os.environ["RFX_RTC_EAGER"] = "1"  # compile at first sight (the library waits for a plan to come back over >= 2^24 rows)
    try:
        for q in queries:
            results.append(check_select(eng, host, q))
    finally:
        del os.environ["RFX_RTC_EAGER"]
    l1, c1 = _rtc_stats(eng)
    if c1 == c0 and l1 == l0:
        pytest.skip("no run-time compiler on this box (libhiprtc.so / kernel sources): the prebuilt kernels answered")
    assert l1 > l0
    try:
        eng.tune(flags=NO_RTC)
        for q, r in zip(queries, results):
            again = check_select(eng, host, q)
            for name in r:
                a, b = r[name].cpu().numpy(), again[name].cpu().numpy()
                if a.dtype != np.float64 or (name in q and q[name][0] in ("min", "max", "first")):
                    assert np.array_equal(a.view(np.int64), b.view(np.int64)), name
        assert _rtc_stats(eng)[0] == l1  # the flag kept every launch on the prebuilt kernels
    finally:
        eng.tune(flags=0)


# ---------------------------------------------------------------- the ranking's device-side bound, the remembered sample
@pytest.mark.parametrize("shape", ["last_row", "first_rows", "one_group", "spread"])
def test_rank_stops_at_the_last_first_row(eng, shape):
    """rfx_rank_slots bounds its bitmap, chunk counts and scan by the last first row, found on the device (k_first_bound): a key that shows up
    in the table's LAST row only (the bound is the whole table), cyclic keys (every first row within the first 200 000 rows: the bound is a
    few hundred chunks), one group, random keys -- group order, first rows and sums against the oracle, on the plane path (2^22+ rows) with
    and without a filter.  The same query runs twice: the second one takes rfx_chunk_scope's remembered sample and must answer alike."""
    n = (1 << 22) + 4_099
    host = table(n, seed=3, keys=200_000)
    if shape == "last_row":
        host["k"][-1] = 777_777
        host["a"][-1] = 5  # (selected by the filter below)
    elif shape == "first_rows":
        host["k"] = (np.arange(n, dtype=np.int64) * 7) % 200_000
    elif shape == "one_group":
        host["k"][:] = 42
    d = dev(eng, host)
    for q in ({"by": "k", "s": ("sum", "v"), "f": ("first", "a")}, {"where": ("<", "a", 300_000), "by": "k", "s": ("sum", "v"), "c": ("count", "a")}):
        r1 = check_select(eng, host, q, d)
        r2 = check_select(eng, host, q, d)
        for name in r1:
            assert torch.equal(r1[name].view(torch.int64), r2[name].view(torch.int64)) or name == "s", name


# ---------------------------------------------------------------- sampled key scopes (large inputs, LDS-sized ranges)
def test_sampled_scope_reports_what_it_missed(eng):
    """From 2^24 rows on, a group-by over an LDS-sized key range takes its scope from a SAMPLE (2^18 strided rows + both ends) instead of
    index_scope_i64's full pass, and the LDS-table kernels report selected rows whose key lies outside it; a report sends the
    query through the exact scope.  Here: ranges the sample sees completely (no retry), one outlier key / a null key / a key
    below the minimum hidden between the sampled rows (retry, same answer), an outlier that the filter removes (no report),
    one and two key columns -- device against itself with the feature off (RFX_NO_SAMPLED_SCOPE) on everything that is
    order-independent, and against numpy on counts."""
    n = (1 << 24) + 12_345
    k = eng.gen_i64(n, 4, 100)
    k2 = eng.gen_i64(n, 14, 7)
    v = eng.gen_f64(n, 5)
    a = eng.gen_i64(n, 2, 1_000_000)
    t = {"k": k, "k2": k2, "v": v, "a": a}

    def run(key, where=None):
        r = eng.group_by(key, [("sum", "a"), ("count", "a"), ("max", "a"), ("first", "a")], where, t)
        return r

    def same(r1, r2):
        assert r1["groups"] == r2["groups"]
        assert torch.equal(r1["keys"], r2["keys"]) and torch.equal(r1["first"], r2["first"])
        for x, y in zip(r1["results"], r2["results"]):
            assert torch.equal(x, y)
        for x, y in zip(r1.get("key_columns", []), r2.get("key_columns", [])):
            assert torch.equal(x, y)

    def exact(key, where=None):
        os.environ["RFX_NO_SAMPLED_SCOPE"] = "1"
        try:
            return run(key, where)
        finally:
            del os.environ["RFX_NO_SAMPLED_SCOPE"]

    hidden = 5  # stride = n >> 18 = 64: row 5 + 64 * j is never sampled, and lies past the first 2^11 rows for j >= 32
    spot = hidden + 64 * 65_537
    base = eng.spec_retries
    same(run("k"), exact("k"))
    same(run(["k", "k2"]), exact(["k", "k2"]))
    same(run("k", ("<", "a", 500_000)), exact("k", ("<", "a", 500_000)))
    assert eng.spec_retries == base  # the sample saw the whole range: no second pass
    counts = torch.bincount(k, minlength=100)
    assert torch.equal(run("k")["results"][1].sum(), counts.sum())
    for bad in (1_000, -3, L_NULL):
        keep = int(k[spot])
        k[spot] = bad
        eng.forget_scopes()  # (a column whose sample failed is not sampled again: forget that between the cases)
        before = eng.spec_retries
        same(run("k"), exact("k"))
        assert eng.spec_retries == before + 1, bad  # reported, ran again under the exact scope
        same(run("k"), exact("k"))
        assert eng.spec_retries == before + 1, bad  # ... and remembered: the same column is not sampled a second time
        eng.forget_scopes()
        before = eng.spec_retries
        same(run(["k", "k2"]), exact(["k", "k2"]))
        assert eng.spec_retries == before + 1, bad
        eng.forget_scopes()
        if bad != L_NULL:
            a_keep = int(a[spot])
            a[spot] = 999_999  # the filter drops the outlier's row: nothing to report
            before = eng.spec_retries
            same(run("k", ("<", "a", 500_000)), exact("k", ("<", "a", 500_000)))
            assert eng.spec_retries == before
            a[spot] = a_keep
        k[spot] = keep


L_NULL = -(2**63)


def test_k1_plan_kernels_match_the_prebuilt_kernels(eng):
    """K1 compiled at run time for one plan (filter_aggr_body with the plan's descriptors as a constexpr, rfx_rtc.hip) against the
    prebuilt instantiations (RFX_TUNE_NO_RTC) on 2^24 + rows: every aggregate kind over i64 / f64 columns with nulls and NaNs, 0 / 1 /
    3 / 6 predicates (atoms, a column operand, a NaN atom), AND / OR, one expression and an expression tree, a ragged tail --
    integer results and exact operations bit for bit, f64 sums to 1e-9 of the column scale."""
    n = (1 << 24) + 4_321
    a = eng.gen_i64(n, 2, 1_000_000)
    b = eng.gen_i64(n, 3, 1_000)
    v = eng.gen_f64(n, 5)
    w = eng.gen_f64(n, 6) - 0.5
    a[17::1009] = L_NULL
    v[23::997] = float("nan")
    t = {"a": a, "b": b, "v": v, "w": w}
    plans = [
        ([("sum", "a"), ("count", "a"), ("min", "a"), ("max", "a"), ("avg", "a"), ("first", "a")], None),
        ([("sum", "v"), ("count", "v"), ("min", "v"), ("max", "v"), ("avg", "v"), ("first", "v")], ("<", "a", 300_000)),
        ([("sum", "w"), ("max", "b")], ("and", ("<", "a", 900_000), (">", "w", -0.3), ("!=", "b", 7))),
        ([("avg", "w"), ("min", "a")], ("or", ("<", "a", 10_000), (">", "w", 0.45), ("==", "b", 3), (">=", "v", 0.99), ("<", "b", "a"), ("<", "v", float("nan")))),
        ([("sum", ("*", "v", "w"))], ("and", ("<", "b", 500), (">=", "w", -0.25), ("<=", "w", 0.25))),
        ([("sum", ("*", ("*", "v", ("-", 1.0, "w")), ("+", 1.0, "v"))), ("max", ("+", "a", "b"))], (">", "a", 123_456)),
    ]
    l0, c0 = _rtc_stats(eng)
    os.environ["RFX_RTC_EAGER"] = "1"
    try:
        fast = [eng.filter_aggr(aggs, where, t) for aggs, where in plans]
    finally:
        del os.environ["RFX_RTC_EAGER"]
    l1, c1 = _rtc_stats(eng)
    if l1 == l0:
        pytest.skip("no run-time compiler on this box (libhiprtc.so / kernel sources): the prebuilt kernels answered")
    try:
        eng.tune(flags=NO_RTC)
        slow = [eng.filter_aggr(aggs, where, t) for aggs, where in plans]
        assert _rtc_stats(eng)[0] == l1
    finally:
        eng.tune(flags=0)
    for (aggs, where), (fv, fs), (sv, ss) in zip(plans, fast, slow):
        assert fs == ss, where
        for (fn, col), x, y in zip(aggs, fv, sv):
            if isinstance(x, float) and fn in ("sum", "avg"):
                assert (math.isnan(x) and math.isnan(y)) or abs(x - y) <= 1e-9 * 2.1 * (n if fn == "sum" else 1), (fn, col, x, y)  # 1e-9 of sum |x_i| <= 2.1 n (avg: per row): summation order only
            else:
                assert (x == y) or (isinstance(x, float) and math.isnan(x) and math.isnan(y)), (fn, col, x, y)


@pytest.mark.parametrize("knob", ["RFX_EMIT_BY_ROWS=2", "RFX_EMIT_BY_ROWS=2,RFX_NO_PACKED_TABLE=1", "RFX_EMIT_BY_ROWS=2,RFX_NO_INSERT_SLOTS=1", "RFX_EMIT_BY_ROWS=0", "RFX_PLH_VAR=0",
                                  "RFX_PLH_VAR=1"])
def test_path_variants_in_a_process_of_their_own(built, knob):
    """Path choices that are read once per process: the hashed group-by's result emitted BY ROWS wherever the probe exists (RFX_EMIT_BY_ROWS=2: the
    route 1e8-group row-hash queries take, forced at test sizes) / by ranking the slots only (=0); the sparse-key aggregate's slot-by-slot probing forms
    (RFX_PLH_VAR=0 / 1; the default is the four-key buckets).  The row-hash, key-tuple, join and sparse-key tests of this file and the golden sets must
    answer the same under each."""
    import subprocess, sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    name, val = knob.split(",")[0].split("=")
    env = dict(os.environ, **dict(kv.split("=") for kv in knob.split(",")))  # (=2 alone: the packed table with the insert pass's slots; then without either)
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "row_hash or key_tuples or several_keys or join" if name == "RFX_EMIT_BY_ROWS" else "sparse or hash"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), os.path.join(here, "test_gpu_golden.py"), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        "-k", f"({sel}) and not variants_in_a_process"], env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2500:] + p.stderr[-1500:]
