"""dev: time Engine.where (rfx_hip_where_estimate + rfx_hip_where_once) on the w2 shape and check the ids against torch:
tools/where_ab.py <rows> <a < threshold of 1e6>; RFX_NO_RTC=1: the prebuilt kernel instead of the one compiled for the predicate list."""
import os, sys, time, torch
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
ms = (time.perf_counter() - t) * 100
ids = torch.as_tensor(ids, device="cuda")
ok = True
step = 1 << 28
pos = 0
for lo in range(0, n, step):
    want = torch.nonzero(a[lo:lo + step] < thr).flatten() + lo
    got = ids[pos:pos + want.numel()]
    ok = ok and got.numel() == want.numel() and bool((got == want).all())
    pos += want.numel()
ok = ok and pos == ids.numel()
print(f"rtc {'off' if os.environ.get('RFX_NO_RTC') else 'on'} rows {n} selected {ids.numel()} ms/query {ms:.3f} ids {'ok' if ok else 'WRONG'}", flush=True)
