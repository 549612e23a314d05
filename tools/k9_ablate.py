import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
rows = 1_000_000_000
k = eng.gen_i64(rows, 4, 1_000_000); k.mul_(1_000_003).sub_(77)
t = {"k": k, "v": eng.gen_f64(rows, 5)}
for _ in range(2): eng.group_by("k", [("sum", "v")], None, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): eng.group_by("k", [("sum", "v")], None, t)
torch.cuda.synchronize()
print(f"RFX_PLH_DBG={os.environ.get('RFX_PLH_DBG','0')}: {(time.perf_counter()-t0)/5*1e3:.2f} ms/query", flush=True)
