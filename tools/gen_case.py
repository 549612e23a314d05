"""The cases of tools/fuzz_null_tuples.py: gen(seed) -> rows, table, by, (kind, span, nulls) per key column, query."""
import numpy as np
NULL=-(2**63)
def gen(seed):
    rng = np.random.default_rng(77_000 + seed)
    n = int(rng.choice([1, 2, 63, 1000, 20_011, 150_003, 400_009]))
    nk = int(rng.integers(2, 5))
    t, by, kinds = {}, {}, []
    for k in range(nk):
        kind = rng.integers(0, 5)
        span = int(rng.choice([1, 2, 7, 100, 5000]))
        col = rng.integers(0, span, n).astype(np.int64)
        if kind == 1: col = col * (1 << 50) - (1 << 52)
        elif kind == 2: col = col - span // 2
        elif kind == 3: col[:] = NULL
        elif kind == 4: col = col + (2**63 - 1 - span)
        nul = 0
        if kind != 3 and rng.random() < 0.7:
            m = rng.random(n) < rng.choice([0.001, 0.05, 0.5]); col[m] = NULL; nul = int(m.sum())
        t[f"k{k}"] = col; by[f"g{k}"] = f"k{k}"; kinds.append((int(kind), span, nul))
    t["v"] = rng.random(n)
    t["a"] = rng.integers(-1000, 1000, n).astype(np.int64)
    t["a"][rng.random(n) < 0.02] = NULL
    pool = [("s", ("sum", "v")), ("c", ("count", "a")), ("mx", ("max", "a")), ("mn", ("min", "v")), ("av", ("avg", "a")), ("si", ("sum", "a")), ("f", ("first", "v"))]
    q = {nm: a for nm, a in pool if rng.random() < 0.5} or {"s": ("sum", "v")}
    q["by"] = by
    return n, t, by, kinds, q
if __name__ == "__main__":
    import sys
    sys.path.insert(0,'/root/repo')
    from oracle import rfo
    for seed in map(int, sys.argv[1:]):
        n,t,by,kinds,q = gen(seed)
        want = rfo.select({"from": t, "c": ("count","a"), "by": by})
        tup = np.stack([t[f"k{k}"] for k in range(len(kinds))],1)
        uniq = len(np.unique(tup, axis=0))
        print(seed, n, kinds, "oracle groups", len(want["c"]), "true distinct tuples", uniq)
