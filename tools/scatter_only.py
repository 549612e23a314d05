"""dev: time ONLY the scope pass of a group-by (the one-pass chunk scatter) -- tools/scatter_only.py <flags> [rows] [where]"""
import sys, time
sys.path.insert(0, "/root/repo")
from rayforce_amd.engine import Engine
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
eng = Engine(0)
t = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5)}
where = None
if len(sys.argv) > 3:
    t["a"] = eng.gen_i64(rows, 2, 1_000_000)
    where = ("<", "a", 100_000)
eng.tune(flags=flags)
for _ in range(2):
    eng.scope(t["k"], where, t, [("sum", "v")])
eng.sync()
t0 = time.perf_counter()
for _ in range(5):
    r = eng.scope(t["k"], where, t, [("sum", "v")])
eng.sync()
print(f"flags {flags}: scope pass {(time.perf_counter() - t0) * 200:.3f} ms  -> {r}")
