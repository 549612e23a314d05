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

# which is it -- where the blocks lie, or who wrote them?  Six more blocks allocated back to back; the FIRST three filled by the upload, the LAST three by a device copy
src = {c: t.clone() for c, t in Cc.items()}
blocks = [pool_block(rows * 8) for _ in range(6)]
D, E = {}, {}
for i, c in enumerate(("k", "v", "a")):
    L.check(lib.rfx_hip_h2d_pipelined(eng._ctx, C.c_void_p(blocks[i]), C.c_void_p(host[c].ctypes.data), C.c_size_t(rows * 8)), "h2d")
    D[c] = as_tensor(blocks[i], src[c].dtype)
    E[c] = as_tensor(blocks[3 + i], src[c].dtype); E[c].copy_(src[c])
print("block addresses:", [hex(b) for b in blocks], flush=True)
run(D, "D blocks 0-2 of six, filled by the upload")
run(E, "E blocks 3-5 of six, filled by a device copy")
# interleaved: k from the first triple, v and a from the second
run({"k": D["k"], "v": E["v"], "a": E["a"]}, "F k from D, v and a from E")
run({"k": E["k"], "v": D["v"], "a": D["a"]}, "G k from E, v and a from D")

# is a block fast or slow BY ITSELF?  a one-column filter -> sum (K1) over each block's copy of the same column
for name, t in (("D.a", D["a"]), ("E.a", E["a"]), ("D.k", D["k"]), ("E.k", E["k"]), ("B.a", B["a"]), ("Cc.a", Cc["a"])):
    for _ in range(3): eng.filter_aggr([("sum", "x")], ("<", "x", 100_000), {"x": t})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.filter_aggr([("sum", "x")], ("<", "x", 100_000), {"x": t})
    torch.cuda.synchronize()
    print(f"K1 over {name:<5} at {hex(t.data_ptr())}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
