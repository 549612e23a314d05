"""dev: does the RELATIVE placement of the three columns of the c3w query matter?  One big device buffer, the columns as views at chosen byte offsets:
  (a) bases a multiple of 64 MB apart (what the library's big-block pool hands out for pinned / uploaded columns),
  (b) the same plus a different skew per column.  Engine.group_by over each, ms per query."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
rows = 1_000_000_000
span = ((rows * 8 + (64 << 20) - 1) // (64 << 20)) * (64 << 20)
buf = torch.empty(3 * span + (5 << 30), dtype=torch.uint8, device="cuda")
src = {"k": eng.gen_i64(rows, 4, 1_000_000), "v": eng.gen_f64(rows, 5), "a": eng.gen_i64(rows, 2, 1_000_000)}
def views(skews):
    out = {}
    for i, c in enumerate(("k", "v", "a")):
        off = i * span + skews[i]
        t = buf[off:off + rows * 8].view(torch.int64 if c != "v" else torch.float64)
        t.copy_(src[c])
        out[c] = t
    return out
def run(t):
    for _ in range(3): eng.group_by("k", [("sum", "v")], ("<", "a", 100_000), t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.group_by("k", [("sum", "v")], ("<", "a", 100_000), t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
print(f"torch's own allocations: {run(src):.3f} ms", flush=True)
for name, sk in (("64 MB multiples apart", (0, 0, 0)), ("+ 4 KB, 8 KB", (0, 4096, 8192)), ("+ 68 KB, 136 KB", (0, 69632, 139264)), ("+ 1 MB + 4 KB, 2 MB + 8 KB", (0, (1 << 20) + 4096, (2 << 20) + 8192)),
                 ("+ 256 B, 512 B", (0, 256, 512)), ("+ 17 MB, 34 MB", (0, 17 << 20, 34 << 20)), ("+ 64 MB, 128 MB", (0, 64 << 20, 128 << 20)),
                 ("+ 192 MB, 448 MB", (0, 192 << 20, 448 << 20)), ("+ 320 MB, 832 MB", (0, 320 << 20, 832 << 20)), ("+ 1 GB, 2 GB", (0, 1 << 30, 2 << 30)),
                 ("+ 1.5 GB + 2 MB, 3 GB + 6 MB", (0, (3 << 29) + (2 << 20), (3 << 30) + (6 << 20))), ("+ 2 MB, 4 MB", (0, 2 << 20, 4 << 20)),
                 ("+ 0.5 GB, 1.25 GB", (0, 1 << 29, 5 << 28))):
    print(f"{name:<32}: {run(views(sk)):.3f} ms", flush=True)
