"""Run-to-run reproducibility (INTEGRATION.md "Reproducibility"): integers, counts, group order and first rows are bit-identical from run
to run on every path; SCALAR f64 sums are too (K1 folds per-workgroup partials in a fixed order, as the reference's pool merges its
workers' partials in task order -- core/pool.c:415-424); GROUPED f64 sums fold through LDS / device atomics whose order follows the wave
schedule, so two runs may differ in the last bits: within 1e-12 of the group's sum of magnitudes here, against the 1e-9 the north star
allows versus the reference."""
import numpy as np
import pytest

from oracle import rfo

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("shape", ["lds", "plane", "plane_filtered", "hash"])
def test_two_runs_agree(eng, shape):
    n = 6_000_011 if shape != "lds" else 1_000_003
    keys = {"lds": 3000, "plane": 700_000, "plane_filtered": 700_000, "hash": 300_000}[shape]
    host = {"k": rfo.gen_i64(n, 4, keys), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5) - 0.5, "w": rfo.gen_f64(n, 6)}
    if shape == "hash":
        host["k"] = host["k"] * 1_000_003 - 77
    dev = {c: eng.column(x) for c, x in host.items()}
    where = ("<", "a", 300_000) if shape == "plane_filtered" else None
    aggs = [("sum", "v"), ("avg", "w"), ("sum", "a"), ("count", "v"), ("min", "v"), ("max", "a")]
    runs = []
    for _ in range(3):
        r = eng.group_by("k", aggs, where, dev)
        runs.append({"keys": r["keys"].cpu().numpy(), "first": r["first"].cpu().numpy(), "res": [x.cpu().numpy() for x in r["results"]]})
    a = runs[0]
    absw = None
    for b in runs[1:]:
        assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["first"], b["first"])  # the groups, their order, their first rows
        for i, (fn, col) in enumerate(aggs):
            x, y = a["res"][i], b["res"][i]
            if fn in ("sum", "avg") and x.dtype == np.float64:  # order-dependent folds: last bits only
                scale = 1.0 if fn == "avg" else float(np.abs(host[col]).sum()) / max(1, len(a["keys"])) * 50
                assert np.all(np.abs(x - y) <= 1e-12 * np.maximum(np.abs(x), scale)), (shape, fn, col, float(np.abs(x - y).max()))
            else:  # integer sums, counts, min / max (exact operations): bit for bit
                assert np.array_equal(_bits(x), _bits(y)), (shape, fn, col)
    # scalar folds: bit for bit, f64 sums included
    q = [("sum", "v"), ("avg", "w"), ("sum", "a"), ("min", "v"), ("max", "w")]
    s0 = eng.filter_aggr(q, ("<", "a", 500_000), dev)
    for _ in range(3):
        s1 = eng.filter_aggr(q, ("<", "a", 500_000), dev)
        assert s0[1] == s1[1]
        for u, w in zip(s0[0], s1[0]):
            assert (u == w) or (u != u and w != w), (shape, u, w)
            if isinstance(u, float):
                assert np.float64(u).view(np.uint64) == np.float64(w).view(np.uint64), (shape, u, w)
