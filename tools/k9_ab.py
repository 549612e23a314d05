"""dev: K9 (sparse keys) A/B in one process: RFX_PLANE_HASH_PARTS=128 (round 3: 128 partitions, two workgroups each) | unset (round 5: 256 partitions with the interleaved key|value plane, one workgroup each) | 192."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
eng = Engine(0)
k = eng.gen_i64(rows, 4, 1_000_000); k.mul_(1_000_003).sub_(77)
t = {"k": k, "v": eng.gen_f64(rows, 5)}
for w in (None, ("<", "v", 0.5)):
    r = None
    for _ in range(2):
        r = eng.group_by("k", [("sum", "v")], w, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        r = eng.group_by("k", [("sum", "v")], w, t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    m = None if w is None else t["v"] < 0.5
    dense = (t["k"] + 77) // 1_000_003
    v = t["v"] if m is None else t["v"][m]; d = dense if m is None else dense[m]
    want = torch.zeros(1_000_000, dtype=torch.float64, device=v.device).index_add_(0, d, v)
    got = torch.zeros_like(want); got[(r["keys"] + 77) // 1_000_003] = r["results"][0]
    ok = bool(((got - want).abs() <= 1e-9 * want.abs() + 1e-300).all()) and bool((r["first"][1:] > r["first"][:-1]).all())
    print(f"PARTS={os.environ.get('RFX_PLANE_HASH_PARTS', 'default(256 kvi)')} where={w} rows={rows}: {ms:.2f} ms/query groups={r['groups']} ok={ok} scatter={eng.stat(0)} fallback={eng.stat(1)}", flush=True)
