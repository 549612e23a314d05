"""One-off: random shapes through round 3's new kernels against the oracle -- k_where_once (1 .. 4 predicate columns, 1 .. 4 comparisons, any
selectivity, ragged sizes, a row offset) and the hash-partitioned planes (sparse keys, one value column, an optional filter column,
RFX_TUNE_CHUNK_SMALL so that they run from 2^16 rows on).  python tools/fuzz_round3.py <first seed> <last seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import rfo
from rayforce_amd.engine import Engine
from test_gpu_parity import check_select, table, dev

eng = Engine(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
OPS = ["<", "<=", ">", ">=", "==", "!="]
for seed in range(lo, hi):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([70_001, 131_072, 200_003, 655_361, 1_500_017]))
    host = table(n, seed=int(rng.integers(0, 50)), keys=int(rng.choice([7, 1000, 50_000])), nulls=bool(rng.integers(0, 2)))
    try:
        # ---- where
        cols = list(rng.choice(["a", "v", "w", "k"], size=int(rng.integers(1, 5)), replace=False))
        preds = []
        for _ in range(int(rng.integers(1, 5))):
            c = str(rng.choice(cols))
            thr = {"a": int(rng.integers(0, 1_000_000)), "k": int(rng.integers(0, 1000)), "v": float(rng.random()), "w": float(rng.random() - 0.5)}[c]
            preds.append((str(rng.choice(OPS)), c, thr))
        spec = preds[0] if len(preds) == 1 else (str(rng.choice(["and", "or"])), *preds)
        row0 = int(rng.choice([0, 10**12]))
        want = rfo.where(rfo.mask_of(spec, host)) + row0
        d = dev(eng, host)
        got = eng.where(spec, d, row0=row0).cpu().numpy()
        assert np.array_equal(got, want), ("where", spec, len(got), len(want))
        # ---- sparse keys through the planes
        sp = dict(host)
        distinct = int(rng.choice([5_000, 60_000, 300_000]))
        sp["k"] = rfo.gen_i64(n, 900 + seed, distinct) * int(rng.choice([1_000_003, 7_777_777_777])) - int(rng.integers(0, 10**9))
        agg = str(rng.choice(["sum", "avg", "min", "max"]))
        vcol = str(rng.choice(["v", "w", "a"]))
        q = {"by": "k", "x": (agg, vcol)}
        if rng.integers(0, 2):
            q["c"] = ("count", vcol)
        if rng.integers(0, 2):
            q["f"] = ("first", vcol)
        if rng.integers(0, 2):
            q["where"] = (str(rng.choice(OPS[:4])), "a", int(rng.integers(100_000, 900_000)))
        eng.tune(flags=32768)
        check_select(eng, sp, q)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "n", n, "->", repr(e)[:300], flush=True)
    finally:
        eng.tune(flags=0)
print("done", hi - lo, "cases,", bad, "failures; plane scatter launches", eng.stat(0), "fallbacks", eng.stat(1))
