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


_DET = r'''
import os, sys, hashlib
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rfo
from rayforce_amd import hostobj as H
ops = H.lib()
ops.rfx_host_bind()
assert ops.rfx_ops_set_deterministic(MODE) == 0
NULL = -(2**63)
out = []
for shape, n, keys in (("lds", 1_000_003, 3000), ("plane", 6_000_011, 700_000), ("hash", 5_000_017, 300_000)):
    host = {"k": rfo.gen_i64(n, 4, keys), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5) - 0.5, "w": rfo.gen_f64(n, 6) * 1000.0}
    if shape == "hash":
        host["k"] = host["k"] * 1_000_003 - 77
    tab = H.table(host)
    st0 = H.to_numpy(ops.rfx_stats(0))
    for q in ({"s": ("sum", "v"), "x": ("avg", "w"), "e": ("sum", ("*", "v", ("-", 1, "w"))), "i": ("sum", "a"), "c": ("count", "v"), "m": ("min", "v"), "by": "k"},
              {"s": ("sum", "v"), "x": ("avg", "w"), "where": ("<", "a", 300_000), "by": "k"}):
        d = H.select_dict(q, tab)
        runs = []
        for _ in range(3):
            r = ops.rfx_select(d)
            assert r and not H.is_error(r), H.error_text(r)
            runs.append(H.table_to_numpy(r))
            ops.rfx_host_drop(r)
        want = rfo.select({"from": host, **q})
        # every group's sum of magnitudes of each argument: what 1e-9 is relative to (v is symmetric around 0: a group's sum may cancel to almost nothing)
        sel = host["a"] < 300_000 if "where" in q else np.ones(n, bool)
        order = np.argsort(want["k"], kind="stable")
        grp = order[np.searchsorted(want["k"][order], host["k"][sel])]
        args = {"s": host["v"], "x": host["w"], "e": host["v"] * (1 - host["w"])}
        cnt = np.bincount(grp, minlength=len(want["k"]))
        mags = {nm: np.bincount(grp, weights=np.abs(args[nm][sel]), minlength=len(want["k"])) / (cnt if nm == "x" else 1) for nm in args if nm in q}
        # ... and the fixed-point bound itself: a cell is rounded to a multiple of 2^-k (k = 62 - e - b, 2^e > max |x| over the COLUMN, 2^b >= the table's rows):
        # a group's sum is off by at most rows_in_group * 2^-(k+1) ABSOLUTE (an average: 2^-(k+1)) -- a one-row group holding a tiny value has no relative bound
        cell = {nm: 2.0 ** -(62 - int(np.frexp(np.abs(args[nm]).max())[1]) - int(np.ceil(np.log2(n))) + 1) * (1 if nm == "x" else cnt) for nm in mags}
        if MODE == 2:  # the second limb adds up what the first one's cells rounded away, scaled by 2^(62 - b) more: far below the oracle's own f64 rounding
            cell = {nm: c * 2.0 ** -(62 - int(np.ceil(np.log2(n)))) for nm, c in cell.items()}
        for name in want:
            for b in runs[1:]:
                assert np.array_equal(np.ascontiguousarray(runs[0][name]).view(np.uint64), np.ascontiguousarray(b[name]).view(np.uint64)), (shape, name, "run to run")
            g, w = runs[0][name], want[name]
            assert g.dtype == w.dtype and g.shape == w.shape, (shape, name)
            if name in mags:  # against the oracle: the fixed-point bound + the oracle's own f64 rounding (1e-12 of the group's sum / mean of magnitudes)
                tol = cell[name] * (1 + 1e-6) + 1e-12 * mags[name]
                assert np.all(np.abs(g - w) <= tol), (shape, name, float((np.abs(g - w) / tol).max()))
                assert np.mean(np.abs(g - w) <= 1e-9 * mags[name]) > 0.999, (shape, name)  # (and 1e-9 relative for all but the tiny one-row groups)
                if MODE == 2:  # two limbs: every group as close to the oracle as two f64 summation orders are to each other
                    assert np.all(np.abs(g - w) <= 1e-12 * mags[name] + 1e-300), (shape, name, float((np.abs(g - w) / np.maximum(mags[name], 1e-300)).max()))
            elif w.dtype == np.float64:
                assert np.allclose(g, w, rtol=1e-12, atol=0), (shape, name)
            else:
                assert np.array_equal(g, w), (shape, name)
            out.append(hashlib.sha256(np.ascontiguousarray(runs[0][name]).tobytes()).hexdigest())
        ops.rfx_host_drop(d)
    # the plain columns' fixed-point images were made ONCE (v, w: two per table) and found again by the later calls; the expression went through scratch
    st = H.to_numpy(ops.rfx_stats(0))
    assert int(st[15] - st0[15]) == 2 * MODE and int(st[16] - st0[16]) == 10 * MODE, (shape, int(st[15] - st0[15]), int(st[16] - st0[16]))
    # a NaN in the argument: that aggregate keeps the default path (NaN semantics as ever), the others stay reproducible
    host2 = dict(host)
    host2["v"] = host["v"].copy()
    host2["v"][12345] = np.nan
    t2 = H.table(host2)
    d = H.select_dict({"s": ("sum", "v"), "x": ("avg", "w"), "by": "k"}, t2)
    r = ops.rfx_select(d)
    assert r and not H.is_error(r), H.error_text(r)
    got, want = H.table_to_numpy(r), rfo.select({"from": host2, "s": ("sum", "v"), "x": ("avg", "w"), "by": "k"})
    assert np.array_equal(np.isnan(got["s"]), np.isnan(want["s"])) and int(np.isnan(got["s"]).sum()) == 1
    for o in (r, d, t2, tab):
        ops.rfx_host_drop(o)
print("DIGEST", hashlib.sha256("".join(out).encode()).hexdigest())
'''


@pytest.mark.parametrize("mode", [1, 2])
def test_deterministic_mode_is_bit_stable_across_runs_and_shard_counts(built, mode):
    """Opt-in reproducible grouped f64 sums (rfx_ops_set_deterministic / RFX_DETERMINISTIC): (sum x) / (avg x) over f64 under by: run as
    integer sums over x scaled by a power of two -- bit-identical from run to run, AND across 1 / 3 / 4 shards (the scale depends on the table only), within
    1e-9 of the oracle; expression aggregates too; a column with a NaN keeps the default path."""
    import os, subprocess, sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for shards, extra in ((1, {}), (3, {"RFX_EXEC_SLICE_SHARDS": "1"}), (4, {})):
        env = dict(os.environ, RFX_SHARDS=str(shards), **extra)
        env.pop("RFX_DETERMINISTIC", None)
        p = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\nMODE = {mode}\n" + _DET], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "DIGEST" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
        digests.append(p.stdout.split("DIGEST")[1].split()[0])
    assert len(set(digests)) == 1, digests
