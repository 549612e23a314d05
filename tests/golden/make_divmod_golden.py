#!/usr/bin/env python
"""Golden vectors for the reference's `/` (ray_div, core/math.c:1138-1364) and `%` (ray_mod, :1449-1530) -- build container only
(needs the compiled reference, oracle/_ref):

    python tests/golden/make_divmod_golden.py      ->  tests/golden/divmod_golden.npz

  * truth tables on special values: every vector (x) vector | atom arm over i64 / f64 (nulls, zero divisors, signs, -0.0, +-inf, NaN);
  * exact multiples and near-multiples with vector AND atom divisors (floor(x / y) at a boundary: where a reciprocal multiply in the
    reference's -O3 -funsafe-math-optimizations build would show);
  * aggregates over such expressions: (sum (/ a b)), (max (% a 7)), (sum (% v 0.25)) ... scalar, filtered and grouped.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref, rfo  # noqa: E402

NULL = np.iinfo(np.int64).min

XI = np.array([0, 1, -1, NULL, 2**63 - 1, 5, -5, 7, -7, 12, -12, 2**62, -(2**62), 9, 3 * (2**40 - 1), 100], np.int64)
XJ = np.array([3, 0, -2, 4, NULL, -3, 3, -7, 7, 4, -4, 3, 7, 3, 3, -100], np.int64)
XF = np.array([0.0, -0.0, np.nan, 1.5, -1.5, np.inf, -np.inf, 2.5, 7.0, -7.0, 0.75, 1e300, 9.0, 0.3, 6.0, -6.0], np.float64)
XG = np.array([2.0, 0.0, 1.0, np.nan, -0.5, 3.0, -2.0, 0.75, 2.0, 2.0, 0.25, 1e-300, 3.0, 0.1, 3.0, 3.0], np.float64)
TAGS = {"ii": "({op} xi xj)", "if": "({op} xi xg)", "fi": "({op} xf xj)", "ff": "({op} xf xg)", "ia": "({op} xi 3)", "ian": "({op} xi -3)", "ai": "({op} 100 xj)",
        "iaf": "({op} xi 2.5)", "iaf3": "({op} xi 3.0)", "fa": "({op} xf 2)", "faf": "({op} xf -1.5)", "faf3": "({op} xf 3.0)", "afi": "({op} 2.5 xj)", "aff": "({op} 7.5 xg)",
        "iz": "({op} xi 0)", "fz": "({op} xf 0.0)"}
OPS = ["/", "%"]
XQ = {"s1": "(sum (/ a b))", "s2": "(sum (% a b))", "mx": "(max (% a 7))", "mn": "(min (/ a 1000))", "s3": "(sum (% v 0.25))", "s4": "(sum (/ v w))",
      "av": "(avg (/ a 3))", "s5": "(sum (* (% a 10) v))"}


def gen_table(n, seed, keys):
    t = {"k": rfo.gen_i64(n, 4 + seed, keys), "a": rfo.gen_i64(n, 2 + seed, 1_000_000), "v": rfo.gen_f64(n, 5 + seed), "w": rfo.gen_f64(n, 6 + seed) - 0.5}
    t["b"] = rfo.gen_i64(n, seed + 7, 9) - 4  # -4 .. 4: zero divisors and both signs
    r = rfo.gen_i64(n, 99 + seed, 100)
    t["a"][r == 0] = NULL
    t["b"][r == 1] = NULL
    t["v"][r == 2] = np.nan
    return t


def main():
    assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
    arrays = dict(d_xi=XI, d_xj=XJ, d_xf=XF, d_xg=XG)
    with ref.Session() as s:
        for nm, a in (("xi", XI), ("xj", XJ), ("xf", XF), ("xg", XG)):
            s.put(nm, a)
        for oi, op in enumerate(OPS):
            for tag, e in TAGS.items():
                s.out(f"dm_{oi}_{tag}", e.format(op=op))
        r = s.run(threads=8)
    for k, v in r.items():
        arrays[k] = v
    for ti, (n, seed, keys) in enumerate([(32_769, 141, 50), (70_003, 142, 3000)]):
        t = gen_table(n, seed, keys)
        for wi, w in enumerate(["", " where: (< b 3)"]):
            with ref.Session() as s:
                s.table("t", t)
                body = " ".join(f"{nm}: {e}" for nm, e in XQ.items())
                s.eval(f"(set r (select {{{body} from: t{w}}}))")
                s.eval(f"(set g (select {{{body} from: t{w} by: k}}))")
                for nm in XQ:
                    s.out(f"dq_{ti}_{wi}_s_{nm}", f"(at r '{nm})")
                    s.out(f"dq_{ti}_{wi}_g_{nm}", f"(at g '{nm})")
                s.out(f"dq_{ti}_{wi}_g_k", "(at g 'k)")
                r = s.run(threads=8)
            arrays.update(r)
    out = os.path.join(HERE, "divmod_golden.npz")
    np.savez_compressed(out, **arrays)
    print(f"{out}: {len(arrays)} arrays")
    for k in ("dm_0_ii", "dm_1_ii", "dm_0_if", "dm_0_ff", "dm_1_ff", "dm_0_iaf3", "dm_1_faf3"):
        print(k, arrays[k].dtype, arrays[k][:10])


if __name__ == "__main__":
    main()
