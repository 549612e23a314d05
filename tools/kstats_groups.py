"""dev: one group-by shape for tools/kstats_py.sh: RFX_KEYS=<n> RFX_AGGS=<1|2|3> RFX_WHERE=<0|1> python tools/kstats_groups.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
N = 1_000_000_000
keys = int(float(os.environ.get("RFX_KEYS", "1e5")))
na = int(os.environ.get("RFX_AGGS", "1"))
v = [eng.gen_f64(N, 11 + i) for i in range(na)]
k = eng.gen_i64(N, 4, keys)
t = {"k": k, **{f"v{i}": v[i] for i in range(na)}}
where = None
if os.environ.get("RFX_WHERE", "0") == "1":
    t["a"] = eng.gen_i64(N, 2, 1_000_000)
    where = ("<", "a", 500_000)
for _ in range(6):
    r = eng.group_by("k", [("sum", f"v{i}") for i in range(na)], where, t)
torch.cuda.synchronize()
print(r["groups"])
