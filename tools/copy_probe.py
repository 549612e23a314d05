"""dev: what a plain device copy reaches on this box (read N bytes + write N bytes) -- the bound of any pass that rewrites its input
once, e.g. the unfiltered chunk scatter (16 B/row in, 16 B/record out).  tools/copy_probe.py [GB=16]"""
import sys, json, torch
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
n = int(gb * 1e9) // 8
a = torch.ones(n, dtype=torch.int64, device="cuda:0")
b = torch.empty_like(a)
out = {}
for name, fn in (("copy_", lambda: b.copy_(a)), ("add_scalar", lambda: torch.add(a, 1, out=b)), ("read_only_sum", lambda: a.sum())):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    moved = (1 if name.startswith("read") else 2) * n * 8
    out[name] = {"ms": round(ms, 3), "TB_per_s": round(moved / ms / 1e9, 3)}
print(json.dumps({"probe": "copy", "bytes_each_way": n * 8, **out}))
