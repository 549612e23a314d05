import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayforce_amd.engine import Engine
eng = Engine(0)
n = 1_000_000
t = {"k": eng.gen_i64(n, 4, 1000), "a": eng.gen_i64(n, 2, 1_000_000), "v": eng.gen_f64(n, 5)}
for _ in range(20):
    eng.select({"from": t, "by": "k", "s": ("sum", "v")})
