"""One-off: random group-bys at PLANE-PATH sizes (2^22 .. 2^24 rows: k_plane_scatter / k_plane_aggregate / the hash-partitioned planes, the ranking's
device-side bound, remembered samples) through Engine.select against the oracle: python tools/fuzz_large.py <first seed> <last seed>.
Key cardinality 1e3 .. 2e6 or sparse, 0-2 comparisons, 1-3 aggregates of every kind, each query asked twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import rfo
from rayforce_amd.engine import Engine
from test_gpu_parity import check_select, dev
lo, hi = int(sys.argv[1]), int(sys.argv[2])
eng = Engine(0, shards=int(os.environ.get("FUZZ_SHARDS", "1")))
fails = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1 << 22, 1 << 24)) | 1
    card = int(rng.choice([1_000, 30_000, 200_000, 1_000_000, 2_000_000]))
    host = {"k": rfo.gen_i64(n, 4 + seed, card), "a": rfo.gen_i64(n, 2 + seed, 1_000_000), "v": rfo.gen_f64(n, 5 + seed), "w": rfo.gen_f64(n, 6 + seed) - 0.5}
    if rng.random() < 0.25:
        host["k"] = host["k"] * 1_000_003 + 17  # sparse
    if rng.random() < 0.3:
        host["k"] = np.sort(host["k"]) if rng.random() < 0.5 else (np.arange(n, dtype=np.int64) % card)  # clustered / cyclic first rows
    if rng.random() < 0.3:
        host["a"][rng.integers(0, n, n // 50)] = -(2**63)
        host["v"][rng.integers(0, n, n // 50)] = np.nan
    q = {"by": "k"}
    pool = [("sum", "v"), ("sum", "a"), ("avg", "w"), ("avg", "a"), ("min", "a"), ("max", "v"), ("count", "a"), ("count", "v"), ("first", "a"), ("min", "w")]
    for i, j in enumerate(rng.choice(len(pool), int(rng.integers(1, 4)), replace=False)):
        q[f"x{i}"] = pool[int(j)]
    r = rng.random()
    if r < 0.35:
        q["where"] = ("<", "a", int(rng.choice([10_000, 100_000, 500_000, 950_000])))
    elif r < 0.55:
        q["where"] = ("and", ("<", "a", int(rng.choice([200_000, 800_000]))), (">", "w", float(rng.choice([-0.4, 0.0, 0.3]))))
    elif r < 0.65:
        q["where"] = ("or", ("<", "a", 50_000), (">", "v", 0.9))
    try:
        d = dev(eng, host)
        check_select(eng, host, q, d)
        check_select(eng, host, q, d)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"seed {seed}: n {n} card {card} {q}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    del host
    eng.trim()
print(f"done seeds {lo}..{hi}: {fails} failures", flush=True)
