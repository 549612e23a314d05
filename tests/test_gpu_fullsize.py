"""BASELINE.json's full sizes (1e9 rows) through size-independent properties -- the oracle would take minutes there.

  * partition of unity: aggregates over `p` and over `not p` add up to the aggregate over everything (i64 exact);
  * `where` ids: ascending, exactly `count` of them, every gathered value satisfies the predicate, none missed;
  * group-by: counts add up to N, sums add up to the column sum, keys distinct, first ids strictly ascending
    (= first-occurrence order) and key[first[g]] == keys[g]; partitioned and atomic paths agree bit-exactly on
    integer outputs.
Columns are generated on the device with the same counter-based generator the oracle has (checked at small n)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N = 1_000_000_000


@pytest.fixture(scope="module")
def big(eng):
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2**30:
        pytest.skip("needs ~60 GB of free HBM")
    t = {"a": eng.gen_i64(N, 2, 1_000_000), "k": eng.gen_i64(N, 4, 1_000_000), "v": eng.gen_f64(N, 5)}
    eng.sync()
    yield t
    del t
    torch.cuda.empty_cache()


def test_c2_filter_sum_partition_of_unity(eng, big):
    aggs = [("sum", "a"), ("count", "a"), ("min", "a"), ("max", "a")]
    (s_all, c_all, mn_all, mx_all), sel_all = eng.filter_aggr(aggs, None, big)
    (s_lo, c_lo, mn_lo, mx_lo), sel_lo = eng.filter_aggr(aggs, ("<", "a", 100_000), big)
    (s_hi, c_hi, mn_hi, mx_hi), sel_hi = eng.filter_aggr(aggs, (">=", "a", 100_000), big)
    assert sel_all == c_all == N and sel_lo + sel_hi == N and c_lo == sel_lo
    assert s_lo + s_hi == s_all  # exact integer arithmetic
    assert mn_all == min(mn_lo, mn_hi) == 0 and mx_all == max(mx_lo, mx_hi) == 999_999 and mx_lo == 99_999 and mn_hi == 100_000
    assert abs(sel_lo / N - 0.1) < 1e-4  # uniform [0, 1e6): 10 % selectivity
    # f64 column, multi-predicate (the C5 shape) -- partition by the first predicate
    w = ("and", ("<", "v", 0.316228), (">=", "a", 500_000))
    (fs, fc), _ = eng.filter_aggr([("sum", "v"), ("count", "v")], w, big)
    (gs, gc), _ = eng.filter_aggr([("sum", "v"), ("count", "v")], ("and", (">=", "v", 0.316228), (">=", "a", 500_000)), big)
    (ts, tc), _ = eng.filter_aggr([("sum", "v"), ("count", "v")], (">=", "a", 500_000), big)
    assert fc + gc == tc and abs((fs + gs) - ts) <= 1e-9 * ts


def test_where_ids_full_size(eng, big):
    ids = eng.where(("<", "a", 1000), big)  # 0.1 % selectivity -> ~1e6 ids
    (cnt,), _ = eng.filter_aggr([("count", "a")], ("<", "a", 1000), big)
    assert ids.numel() == cnt
    assert bool((ids[1:] > ids[:-1]).all()) and int(ids[0]) >= 0 and int(ids[-1]) < N
    got = eng.at_ids(big["a"], ids)
    assert bool((got < 1000).all())
    (s,), _ = eng.filter_aggr([("sum", "a")], ("<", "a", 1000), big)
    assert int(got.sum()) == s
    # 10 % selectivity: 1e8 ids (0.8 GB), same checks on the count / order / sum
    ids = eng.where(("<", "a", 100_000), big)
    (cnt, s), _ = eng.filter_aggr([("count", "a"), ("sum", "a")], ("<", "a", 100_000), big)
    assert ids.numel() == cnt and bool((ids[1:] > ids[:-1]).all())
    assert int(eng.at_ids(big["a"], ids).sum()) == s


def test_c3_group_by_full_size(eng, big):
    r = eng.group_by("k", [("sum", "v"), ("count", "v"), ("max", "a")], None, big)
    g = r["groups"]
    assert g == 1_000_000  # every key of [0, 1e6) occurs in 1e9 uniform draws
    keys, first, (sums, counts, maxa) = r["keys"], r["first"], r["results"]
    assert int(counts.sum()) == N
    (tot,), _ = eng.filter_aggr([("sum", "v")], None, big)
    assert abs(float(sums.sum()) - tot) <= 1e-9 * tot
    assert int(torch.unique(keys).numel()) == g
    assert bool((first[1:] > first[:-1]).all())  # group order == first-occurrence order
    assert torch.equal(eng.at_ids(big["k"], first), keys)  # the key at each group's first row is that group's key
    assert int(maxa.max()) == 999_999
    # the same query through device-scope atomics (tune flag 2 disables the partitioned path): integer outputs identical
    eng.tune(flags=2)  # RFX_TUNE_NO_PARTITION
    try:
        r2 = eng.group_by("k", [("sum", "v"), ("count", "v"), ("max", "a")], None, big)
    finally:
        eng.tune(flags=0)
    assert torch.equal(r2["keys"], keys) and torch.equal(r2["first"], first)
    assert torch.equal(r2["results"][1], counts) and torch.equal(r2["results"][2], maxa)
    assert torch.allclose(r2["results"][0], sums, rtol=1e-9, atol=0)
    # with a filter: groups are the keys that survive, counts add up to the selected rows
    r3 = eng.group_by("k", [("count", "v")], ("<", "a", 100_000), big)
    (sel,), _ = eng.filter_aggr([("count", "a")], ("<", "a", 100_000), big)
    assert int(r3["results"][0].sum()) == sel and bool((r3["first"][1:] > r3["first"][:-1]).all())


def test_c3w_filtered_group_by_full_size(eng, big):
    """The metric's literal shape at 1e9 rows: compaction-first path vs the write-combining scatter with predicates (tune 256)."""
    w = ("<", "a", 100_000)
    r = eng.group_by("k", [("sum", "v"), ("count", "a"), ("min", "a")], w, big)
    (sel, tot), _ = eng.filter_aggr([("count", "a"), ("sum", "v")], w, big)
    assert int(r["results"][1].sum()) == sel and abs(float(r["results"][0].sum()) - tot) <= 1e-9 * tot
    assert bool((r["first"][1:] > r["first"][:-1]).all()) and torch.equal(eng.at_ids(big["k"], r["first"]), r["keys"])
    assert bool((eng.at_ids(big["a"], r["first"]) < 100_000).all()) and int(r["results"][2].max()) < 100_000
    eng.tune(flags=256)  # RFX_TUNE_NO_SEL_COMPACT
    try:
        r2 = eng.group_by("k", [("sum", "v"), ("count", "a"), ("min", "a")], w, big)
    finally:
        eng.tune(flags=0)
    assert torch.equal(r2["keys"], r["keys"]) and torch.equal(r2["first"], r["first"])
    assert torch.equal(r2["results"][1], r["results"][1]) and torch.equal(r2["results"][2], r["results"][2])
    assert torch.allclose(r2["results"][0], r["results"][0], rtol=1e-9, atol=0)


def test_multikey_expression_and_sparse_full_size(eng, big):
    """1e9 rows through the widened paths: two by: columns, an expression aggregate, bucketed keys, sparse keys (partitioned hash
    path vs the device-wide table)."""
    id1, id2 = eng.gen_i64(N, 10, 100), eng.gen_i64(N, 11, 100)
    r = eng.group_by([id1, id2], [("sum", big["v"]), ("count", big["a"])], None, None)
    assert r["groups"] == 10_000 and int(r["results"][1].sum()) == N
    k1, k2 = r["key_columns"]
    assert torch.equal(eng.at_ids(id1, r["first"]), k1) and torch.equal(eng.at_ids(id2, r["first"]), k2)
    assert int(torch.unique(k1 * 100 + k2).numel()) == 10_000 and bool((r["first"][1:] > r["first"][:-1]).all())
    del id1, id2, r
    # sum(v * a) = sum over groups of sum(v * a); i64 product sum exact
    (xs, xi), _ = eng.filter_aggr([("sum", ("*", big["v"], big["a"])), ("sum", ("*", big["a"], 3))], ("<", big["a"], 500_000), None)
    rg = eng.group_by(("xbar", big["a"], 1000), [("sum", ("*", big["v"], big["a"])), ("sum", ("*", big["a"], 3)), ("count", big["a"])],
                      ("<", big["a"], 500_000), None)
    assert rg["groups"] == 500 and int(rg["results"][1].sum()) == xi and abs(float(rg["results"][0].sum()) - xs) <= 1e-9 * abs(xs)
    assert bool((rg["keys"] % 1000 == 0).all())
    del rg
    # sparse keys: 1e6 distinct keys spread over 1e12
    ks = big["k"] * 1_000_003 - 77
    r = eng.group_by(ks, [("sum", big["v"]), ("count", big["a"])], None, None)
    assert r["groups"] == 1_000_000 and not r["dense"] and int(r["results"][1].sum()) == N
    assert torch.equal(eng.at_ids(ks, r["first"]), r["keys"]) and bool((r["first"][1:] > r["first"][:-1]).all())
    eng.tune(flags=2)
    try:
        r2 = eng.group_by(ks, [("sum", big["v"]), ("count", big["a"])], None, None)
    finally:
        eng.tune(flags=0)
    assert torch.equal(r2["keys"], r["keys"]) and torch.equal(r2["results"][1], r["results"][1])
    assert torch.allclose(r2["results"][0], r["results"][0], rtol=1e-9, atol=0)


def test_beyond_32_bit_row_counts(eng):
    """2^32 + 12 345 rows (34 GB per column): K1, `where` and the group-by fall-backs index rows with 64 bits."""
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2**30:
        pytest.skip("needs ~150 GB of free HBM")
    n = (1 << 32) + 12_345
    a = eng.gen_i64(n, 2, 1_000_000)
    (cnt, s, mx), sel = eng.filter_aggr([("count", a), ("sum", a), ("max", a)], ("<", a, 100_000), None)
    (cnt2, s2), sel2 = eng.filter_aggr([("count", a), ("sum", a)], (">=", a, 100_000), None)
    (tot,), _ = eng.filter_aggr([("sum", a)], None, None, nrows=n)
    assert sel + sel2 == n and cnt == sel and s + s2 == tot and mx == 99_999
    ids = eng.where(("<", a, 100), None)
    assert bool((ids[1:] > ids[:-1]).all()) and int(ids[-1]) < n and int(ids[-1]) > (1 << 32) - 10_000_000
    assert bool((eng.at_ids(a, ids) < 100).all())
    del ids
    k = eng.gen_i64(n, 4, 1000)
    r = eng.group_by(k, [("count", a), ("sum", a)], None, None)
    assert r["groups"] == 1000 and int(r["results"][0].sum()) == n and int(r["results"][1].sum()) == tot
    assert bool((r["first"][1:] > r["first"][:-1]).all()) and torch.equal(eng.at_ids(k, r["first"]), r["keys"])
