"""LDS-table group-by under a filter, narrow plans (1e9 rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
eng.tune(flags=int(os.environ.get("RFX_FLAGS", "0")))
N = 1_000_000_000
T = {"k": eng.gen_i64(N, 4, 100), "k4": eng.gen_i64(N, 5, 10_000), "a": eng.gen_i64(N, 2, 1_000_000), "v": eng.gen_f64(N, 5), "w": eng.gen_f64(N, 6)}
for name, key, aggs, w in (("sum v by k(100)", "k", [("sum", "v")], None), ("sum v by k(100) where a<5e5", "k", [("sum", "v")], ("<", "a", 500_000)),
                           ("sum v, avg w by k(100) where a<5e5 and v>0.1", "k", [("sum", "v"), ("avg", "w")], ("and", ("<", "a", 500_000), (">", "v", 0.1))),
                           ("sum v by k(1e4) where a<5e5", "k4", [("sum", "v")], ("<", "a", 500_000)), ("sum v by k(1e4)", "k4", [("sum", "v")], None)):
    for _ in range(2):
        r = eng.group_by(key, aggs, w, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = eng.group_by(key, aggs, w, T)
    torch.cuda.synchronize()
    print(f"{name:<50} {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
