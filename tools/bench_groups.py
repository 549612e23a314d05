"""Group-by across key cardinalities / aggregate counts (H2O-like shapes: BASELINE.md section 1 Q4/Q5) on 1e9 rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
eng.tune(flags=int(os.environ.get("RFX_FLAGS", "0")))
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
v1, v2, v3 = eng.gen_f64(N, 11), eng.gen_f64(N, 12), eng.gen_f64(N, 13)
KEYS = [int(float(x)) for x in os.environ.get("RFX_KEYS", "100,1e4,1e5,1e6").split(",")]
for keys in KEYS:
    k = eng.gen_i64(N, 4, keys)
    t = {"k": k, "v1": v1, "v2": v2, "v3": v3}
    for name, aggs in (("sum v1", [("sum", "v1")]), ("sum v1, avg v3", [("sum", "v1"), ("avg", "v3")]), ("avg v1,v2,v3", [("avg", "v1"), ("avg", "v2"), ("avg", "v3")]),
                       ("sum v1,v2,v3", [("sum", "v1"), ("sum", "v2"), ("sum", "v3")])):
        for _ in range(2):
            r = eng.group_by("k", aggs, None, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = eng.group_by("k", aggs, None, t)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        byts = 8 * (1 + len(aggs)) * N
        print(f"keys {keys:>8} {name:<14} groups {r['groups']:>8} {ms:8.2f} ms  {N / ms / 1e6:7.1f} G rows/s  {byts / ms / 1e6:6.0f} GB/s ({byts / ms / 8e7:.0f}% of 8 TB/s)", flush=True)
    del k
