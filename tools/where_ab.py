"""dev: time Engine.where (rfx_hip_where_estimate + rfx_hip_where_once) on the w2 shape: tools/where_ab.py <rows> <a < threshold of 1e6>."""
import sys, time, torch
sys.path.insert(0, ".")
from rayforce_amd.engine import Engine
eng = Engine(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
a = eng.gen_i64(n, 2, 1_000_000)
for _ in range(3):
    ids = eng.where(("<", "a", thr), {"a": a})
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    ids = eng.where(("<", "a", thr), {"a": a})
torch.cuda.synchronize()
print(f"rows {n} selected {ids.numel()} ms/query {(time.perf_counter() - t) * 100:.3f}")
