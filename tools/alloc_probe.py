"""dev: why is the c3w query ~5 % slower over the library's own device copies (pinned host columns) than over torch-owned columns?
The same data in (A) torch allocations, (B) blocks of the context's pool (rfx_hip_malloc) filled by a device copy, (C) such blocks filled by the
pipelined upload from host memory, (D) plain hipMalloc blocks (torch.cuda.caching_allocator bypassed: one hipMalloc each via a fresh big torch tensor of
odd size); Engine.group_by over each, ms per query."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rayforce_amd import _lib as L
from rayforce_amd.engine import Engine, _View
eng = Engine(0); lib = eng.lib
rows = 1_000_000_000
src = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5), "a": eng.gen_i64(rows, 2, 1_000_000)}
def run(t, label):
    for _ in range(3): eng.group_by("k", [("sum", "v")], ("<", "a", 100_000), t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.group_by("k", [("sum", "v")], ("<", "a", 100_000), t)
    torch.cuda.synchronize()
    print(f"{label:<60}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
run(src, "A torch allocations (generated on the device)")
def pool_block(nbytes):
    p = C.c_void_p()
    L.check(lib.rfx_hip_malloc(eng._ctx, C.byref(p), C.c_size_t(nbytes)), "malloc")
    return p.value
def as_tensor(ptr, dt):
    return torch.as_tensor(_View(None, ptr, rows, "<i8" if dt == torch.int64 else "<f8"), device="cuda")
B = {}
for c, t in src.items():
    ptr = pool_block(rows * 8)
    B[c] = as_tensor(ptr, t.dtype); B[c].copy_(t)
run(B, "B the context's pool blocks (64 MB classes), device copy")
host = {c: t.cpu().numpy() for c, t in src.items()}
Cc = {}
for c, t in src.items():
    ptr = pool_block(rows * 8)
    L.check(lib.rfx_hip_h2d_pipelined(eng._ctx, C.c_void_p(ptr), C.c_void_p(host[c].ctypes.data), C.c_size_t(rows * 8)), "h2d")
    Cc[c] = as_tensor(ptr, t.dtype)
run(Cc, "C pool blocks filled by the pipelined upload")
del src; torch.cuda.empty_cache()
run(B, "B again, torch's own copies freed")
run(Cc, "C again, torch's own copies freed")
