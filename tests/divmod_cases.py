"""The reference's answers for `/` (ray_div) and `%` (ray_mod): tests/golden/divmod_golden.npz, written by tests/golden/make_divmod_golden.py
from the compiled reference.  Shared by the oracle test (CPU) and the device test."""
import os

import numpy as np

from oracle import rfo

_g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "divmod_golden.npz"))
NULL = np.iinfo(np.int64).min
XI, XJ, XF, XG = _g["d_xi"], _g["d_xj"], _g["d_xf"], _g["d_xg"]
ARGS = {"ii": (XI, XJ), "if": (XI, XG), "fi": (XF, XJ), "ff": (XF, XG), "ia": (XI, 3), "ian": (XI, -3), "ai": (100, XJ), "iaf": (XI, 2.5), "iaf3": (XI, 3.0),
        "fa": (XF, 2), "faf": (XF, -1.5), "faf3": (XF, 3.0), "afi": (2.5, XJ), "aff": (7.5, XG), "iz": (XI, 0), "fz": (XF, 0.0)}
OPS = ["/", "%"]
XQ = {"s1": ("sum", ("/", "a", "b")), "s2": ("sum", ("%", "a", "b")), "mx": ("max", ("%", "a", 7)), "mn": ("min", ("/", "a", 1000)), "s3": ("sum", ("%", "v", 0.25)),
      "s4": ("sum", ("/", "v", "w")), "av": ("avg", ("/", "a", 3)), "s5": ("sum", ("*", ("%", "a", 10), "v"))}
TABLES = [(32_769, 141, 50), (70_003, 142, 3000)]
WHERES = [None, ("<", "b", 3)]


def truth_tables():
    for oi, op in enumerate(OPS):
        for tag, (l, r) in ARGS.items():
            yield op, tag, l, r, _g[f"dm_{oi}_{tag}"]


def gen_table(n, seed, keys):  # == make_divmod_golden.gen_table
    t = {"k": rfo.gen_i64(n, 4 + seed, keys), "a": rfo.gen_i64(n, 2 + seed, 1_000_000), "v": rfo.gen_f64(n, 5 + seed), "w": rfo.gen_f64(n, 6 + seed) - 0.5}
    t["b"] = rfo.gen_i64(n, seed + 7, 9) - 4
    r = rfo.gen_i64(n, 99 + seed, 100)
    t["a"][r == 0] = NULL
    t["b"][r == 1] = NULL
    t["v"][r == 2] = np.nan
    return t


def query_cases():
    for ti, (n, seed, keys) in enumerate(TABLES):
        for wi, w in enumerate(WHERES):
            for grouped in (False, True):
                tag = "g" if grouped else "s"
                want = {nm: _g[f"dq_{ti}_{wi}_{tag}_{nm}"] for nm in XQ}
                if grouped:
                    want["k"] = _g[f"dq_{ti}_{wi}_g_k"]
                yield f"t{ti}w{wi}{tag}", (n, seed, keys), w, grouped, want


def same(got, want, what, sums=False):
    """bit-exact for integers and for element-wise f64 results; sums / averages of f64 within 1e-9 relative to sum |x| (any order)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype and got.shape == want.shape, (what, got.dtype, want.dtype, got.shape, want.shape)
    if want.dtype == np.float64:
        assert np.array_equal(np.isnan(got), np.isnan(want)), what
        ok = ~np.isnan(want)
        if sums:
            assert np.all(np.abs(got[ok] - want[ok]) <= 1e-9 * np.maximum(np.abs(want[ok]), 1.0)), what
        else:
            assert np.array_equal(got[ok], want[ok]), (what, got[ok][got[ok] != want[ok]][:4], want[ok][got[ok] != want[ok]][:4])
    else:
        assert np.array_equal(got, want), (what, got[:8], want[:8])
