"""Where the wide-plan group-by spends its time: the Q1 shape with parts taken away (1e9 rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
eng.tune(flags=int(os.environ.get("RFX_FLAGS", "0")))
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
q = eng.gen_i64(N, 73, 50); q.add_(1)
d, t = eng.gen_f64(N, 75), eng.gen_f64(N, 76); d.mul_(0.1); t.mul_(0.08)
T = {"rf": eng.gen_i64(N, 71, 3), "ls": eng.gen_i64(N, 72, 2), "q": q, "p": eng.gen_f64(N, 74), "d": d, "t": t, "sd": eng.gen_i64(N, 77, 2500)}
e1, e2 = ("*", "p", ("-", 1, "d")), ("*", ("*", "p", ("-", 1, "d")), ("+", 1, "t"))
full = [("sum", "q"), ("sum", "p"), ("sum", e1), ("sum", e2), ("avg", "q"), ("avg", "p"), ("avg", "d"), ("count", "q")]
plain = [("sum", "q"), ("sum", "p"), ("sum", "d"), ("sum", "t"), ("avg", "q"), ("avg", "p"), ("avg", "d"), ("count", "q")]
W = ("<=", "sd", 2400)
cases = [("full Q1 shape", ["rf", "ls"], full, W), ("no expressions (8 plain aggregates)", ["rf", "ls"], plain, W), ("no where", ["rf", "ls"], full, None),
         ("one key column", "rf", full, W), ("4 plain sums", ["rf", "ls"], plain[:4], W), ("2 expression sums only", ["rf", "ls"], full[2:4], W),
         ("1 plain sum", ["rf", "ls"], plain[:1], W), ("8 plain, one key, no where", "rf", plain, None)]
for name, key, aggs, w in cases:
    for _ in range(2):
        r = eng.group_by(key, aggs, w, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = eng.group_by(key, aggs, w, T)
    torch.cuda.synchronize()
    print(f"{name:<40} {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms  groups {r['groups']}", flush=True)
