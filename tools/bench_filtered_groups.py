"""dev: FILTERED group-bys with two / three value planes and sparse keys (the plane scatter's NP >= 1, NV >= 2 instantiations) on 1e9 rows:
one line per shape, checked against torch on the group count and the count / sum totals."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
v1, v2, v3 = eng.gen_f64(N, 11), eng.gen_f64(N, 12), eng.gen_f64(N, 13)
a = eng.gen_i64(N, 2, 1_000_000)
k = eng.gen_i64(N, 4, 1_000_000)
ks = k * 1_000_003 + 17  # sparse: range >> rows
t = {"k": k, "ks": ks, "a": a, "v1": v1, "v2": v2, "v3": v3}
shapes = [("k", "a < 100000", ("<", "a", 100_000), [("sum", "v1")]),
          ("k", "a < 100000", ("<", "a", 100_000), [("sum", "v1"), ("avg", "v3")]),
          ("k", "a < 500000", ("<", "a", 500_000), [("sum", "v1"), ("avg", "v3")]),
          ("k", "a < 500000", ("<", "a", 500_000), [("sum", "v1"), ("sum", "v2"), ("sum", "v3")]),
          ("k", "v1 < 0.5", ("<", "v1", 0.5), [("sum", "v1"), ("sum", "v2")]),
          ("ks", "none", None, [("sum", "v1")]),
          ("ks", "a < 500000", ("<", "a", 500_000), [("sum", "v1")]),
          ("ks", "v1 < 0.5", ("<", "v1", 0.5), [("sum", "v1")])]
for key, wname, where, aggs in shapes:
    for _ in range(2):
        r = eng.group_by(key, aggs + [("count", aggs[0][1])], where, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = eng.group_by(key, aggs + [("count", aggs[0][1])], where, t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    sel = torch.ones(N, dtype=torch.bool, device="cuda") if where is None else (t[where[1]] < where[2])
    nsel = int(sel.sum())
    cnt = torch.as_tensor(r["results"][-1], device="cuda")
    s0 = torch.as_tensor(r["results"][0], device="cuda").sum().item()
    want = t[aggs[0][1]][sel].sum().item()
    ok = int(cnt.sum()) == nsel and abs(s0 - want) <= 1e-9 * abs(want)
    del sel
    print(f"by {key:<2} where {wname:<10} {'+'.join(f + ' ' + c for f, c in aggs):<28} groups {r['groups']:>8} {ms:8.2f} ms  {'ok' if ok else 'WRONG'}", flush=True)
