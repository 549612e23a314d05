"""tools/h2d_bench.py -- host-to-device rates on the GPU box: plain hipMemcpy from pageable memory vs the pipelined
pinned-staging path, from a heap array and from an mmapped column file (what the reference's `get` hands over)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref, rfo
from rayforce_amd.engine import Engine
from rayforce_amd import _lib as L

eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
a = rfo.gen_i64(n, 2, 1_000_000)
d = torch.empty(n, dtype=torch.int64, device=eng.device)
gb = a.nbytes / 1e9
for name, fn in (("plain h2d (pageable heap)", lambda: L.check(eng.lib.rfx_hip_h2d(eng._ctx, d.data_ptr(), a.ctypes.data, a.nbytes))),
                 ("pipelined (heap)", lambda: L.check(eng.lib.rfx_hip_h2d_pipelined(eng._ctx, d.data_ptr(), a.ctypes.data, a.nbytes)))):
    for rep in range(3):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name:32s} {gb / dt:6.1f} GB/s  ({dt * 1e3:.0f} ms for {gb:.1f} GB)")
path = "/dev/shm/rfx_h2d_col" if os.path.isdir("/dev/shm") else "/tmp/rfx_h2d_col"
ref.write_col(path, a)
for rep in range(3):
    t0 = time.perf_counter(); t = eng.load_column(path); dt = time.perf_counter() - t0
print(f"{'column file (mmap, page cache)':32s} {gb / dt:6.1f} GB/s  ({dt * 1e3:.0f} ms)")
assert int(t.sum()) == int(a.sum())
import mmap
with open(path, "rb") as fh:
    for rep in range(3):
        m = mmap.mmap(fh.fileno(), 0, prot=mmap.PROT_READ)
        buf = np.frombuffer(m, dtype=np.uint8, offset=16)
        t0 = time.perf_counter()
        L.check(eng.lib.rfx_hip_h2d(eng._ctx, d.data_ptr(), buf.ctypes.data, a.nbytes))
        dt = time.perf_counter() - t0
        del buf
        m.close()
print(f"{'plain h2d from a fresh mmap':32s} {gb / dt:6.1f} GB/s  ({dt * 1e3:.0f} ms)")
os.remove(path)
