"""dev: sparse keys with TWO aggregates (sum + count: 28-byte LDS entries) at 1e9 rows / 1e6 keys -- ms per query, checked against torch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
k0 = eng.gen_i64(rows, 4, 1_000_000)
k = k0 * 1_000_003 - 77
v = eng.gen_f64(rows, 5)
t = {"k": k, "v": v}
aggs = [("sum", "v"), ("count", "v")]
for _ in range(2):
    r = eng.group_by("k", aggs, None, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    r = eng.group_by("k", aggs, None, t)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
keys = r["keys"]
kk = (keys + 77) // 1_000_003
want_s = torch.zeros(1_000_000, dtype=torch.float64, device=v.device).index_add_(0, k0, v)
want_c = torch.bincount(k0, minlength=1_000_000)
ok = bool(torch.allclose(r["results"][0], want_s[kk], rtol=1e-9, atol=0)) and bool(torch.equal(r["results"][1], want_c[kk])) and int(r["groups"]) == int((want_c > 0).sum())
first = r["first"]
ok = ok and bool((first[1:] > first[:-1]).all())
print(f"k9 sum+count RFX_PLH_VAR={os.environ.get('RFX_PLH_VAR', '(default)')}: {ms:.2f} ms/query  groups {int(r['groups'])}  verified {ok}  paths plane_scatter {eng.stat(0)} plane_aggregate {eng.stat(2)}", flush=True)
