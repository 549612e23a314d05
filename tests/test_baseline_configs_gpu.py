"""Every BASELINE.json config's EXACT workload -- same seeds, same columns, same predicates, same aggregates as bench.py times --
against the oracle at sizes it finishes in seconds (1e6 / 1e7 rows), and at the config's full size through size-independent
properties (partition of unity, group sums adding up to the scalar sum).  The test ids carry the config names.

  C1  configs[0]  (sum v), v f64 seed 1
  C2  configs[1]  select sum(a) where a < 100000, a i64 seed 2 mod 1e6
  C2b north star  select sum(b) where a < 100000, + b f64 seed 3
  C3  configs[2]  select sum(v) by k, k i64 seed 4 mod 1e6, v f64 seed 5
  C3w metric      C3 where a < 100000
  C4  configs[3]  C3 row-range sharded: tests/test_sharded_gpu.py (the planner's shards on one device), tests/test_dist_multi_gpu.py (N devices)
  C5  configs[4]  avg, min, max(d) where a < 0.316228 and b > 0.683772 and c != 0.25, a, b, c, d f64 seeds 6, 7, 8, 9
"""
import numpy as np
import pytest
import torch

from oracle import rfo

pytestmark = pytest.mark.gpu
RTOL = 1e-9

C5_WHERE = ("and", ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25))
C5_AGGS = {"x": ("avg", "d"), "y": ("min", "d"), "z": ("max", "d")}


def host_columns(name, n):
    if name == "c1":
        return {"v": rfo.gen_f64(n, 1)}
    if name == "c2":
        return {"a": rfo.gen_i64(n, 2, 1_000_000)}
    if name == "c2b":
        return {"a": rfo.gen_i64(n, 2, 1_000_000), "b": rfo.gen_f64(n, 3)}
    if name in ("c3", "c4"):
        return {"k": rfo.gen_i64(n, 4, 1_000_000), "v": rfo.gen_f64(n, 5)}
    if name == "c3w":
        return {"k": rfo.gen_i64(n, 4, 1_000_000), "v": rfo.gen_f64(n, 5), "a": rfo.gen_i64(n, 2, 1_000_000)}
    if name == "c5":
        return {c: rfo.gen_f64(n, s) for c, s in zip("abcd", (6, 7, 8, 9))}
    raise KeyError(name)


QUERIES = {
    "c1": {"s": ("sum", "v")},
    "c2": {"where": ("<", "a", 100_000), "s": ("sum", "a")},
    "c2b": {"where": ("<", "a", 100_000), "s": ("sum", "b")},
    "c3": {"by": "k", "s": ("sum", "v")},
    "c3w": {"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")},
    "c5": {"where": C5_WHERE, **C5_AGGS},
}


def check(eng, host, q):
    got = eng.select({"from": {k: eng.column(v) for k, v in host.items()}, **q})
    want = rfo.select({"from": host, **q})
    assert list(got.keys()) == list(want.keys())
    for name, w in want.items():
        g = got[name].cpu().numpy()
        assert g.dtype == w.dtype and g.shape == w.shape, name
        if w.dtype == np.float64 and name in q and q[name][0] in ("sum", "avg"):
            assert np.all(np.abs(g - w) <= RTOL * np.maximum(np.abs(w), 1e-300)), name  # positive terms: |sum| = sum |x|
        else:
            assert np.array_equal(g, w, equal_nan=(w.dtype == np.float64)), name  # keys, order, min / max: bit-exact
    return got


@pytest.mark.parametrize("n", [1_000_000, 10_000_000])
@pytest.mark.parametrize("config", ["c1", "c2", "c2b", "c3", "c3w", "c5"])
def test_baseline_config_against_the_oracle(eng, config, n):
    if config in ("c3", "c3w") and n == 10_000_000:
        eng.tune(flags=0)  # (default thresholds: 1e7 rows take the one-pass plane partitioning)
    check(eng, host_columns(config, n), QUERIES[config])


def test_c5_exact_workload_selectivity_and_device_generator(eng):
    """C5 at 1e7 rows: the device-side generator gives the oracle's columns, the three predicates keep ~10 % (0.316228 x 0.316228 x 1)."""
    n = 10_000_000
    host = host_columns("c5", n)
    for c, s in zip("abcd", (6, 7, 8, 9)):
        assert torch.equal(eng.gen_f64(n, s).cpu(), torch.from_numpy(host[c]))
    sel = rfo.where(rfo.cmp("<", host["a"], 0.316228) & rfo.cmp(">", host["b"], 0.683772) & rfo.cmp("!=", host["c"], 0.25))
    assert abs(len(sel) / n - 0.1) < 2e-3


def test_c5_full_size_partition_of_unity(eng):
    """configs[4] at its full 2e9 rows x 4 f64 columns (64 GB): the three-predicate selection and its complement's three pieces tile the
    table (counts add up exactly, sums within 1e-9, min / max fold to the whole column's), and the device generator's seeds are the ones
    the oracle checked at 1e7 rows above."""
    n = 2_000_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs ~80 GB of free HBM")
    t = {c: eng.gen_f64(n, s) for c, s in zip("abcd", (6, 7, 8, 9))}
    aggs = [("sum", "d"), ("count", "d"), ("min", "d"), ("max", "d"), ("avg", "d")]
    pa, pb, pc = ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25)
    na, nb, nc = (">=", "a", 0.316228), ("<=", "b", 0.683772), ("==", "c", 0.25)
    pieces = [("and", pa, pb, pc), na, ("and", pa, nb), ("and", pa, pb, nc)]  # disjoint, together everything
    res = [eng.filter_aggr(aggs, w, t, nrows=n) for w in pieces]
    (s_all, c_all, mn_all, mx_all, _), sel_all = eng.filter_aggr(aggs, None, t, nrows=n)
    assert sel_all == n == c_all
    assert sum(r[1] for r in res) == n and all(r[0][1] == r[1] for r in res)
    assert abs(sum(r[0][0] for r in res) - s_all) <= RTOL * s_all
    assert min(r[0][2] for r in res if r[1]) == mn_all and max(r[0][3] for r in res if r[1]) == mx_all
    (s, c, mn, mx, av), sel = res[0]
    assert abs(sel / n - 0.1) < 1e-3 and abs(av - s / c) <= RTOL * av and mn_all <= mn <= av <= mx <= mx_all
    del t
    torch.cuda.empty_cache()
