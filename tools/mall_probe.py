"""tools/mall_probe.py -- does a buffer that was just WRITTEN come back faster than HBM (memory-side cache, 256 MB)?  Times the K1
streaming read of an n-row column right after it was generated, and again after 4 GB of unrelated reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
big = eng.gen_i64(512_000_000, 9, 1000)  # 4 GB of something else
for mb in (16, 32, 64, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 8
    col = eng.empty(n)
    res = {}
    for mode in ("warm-after-write", "cold"):
        best = 1e9
        for rep in range(5):
            eng.gen_i64(n, 3, 1_000_000, out=col)
            if mode == "cold":
                eng.filter_aggr([("sum", big)], None, None)
            eng.sync()
            eng.timer_start()
            eng.lib.rfx_hip_filter_aggr  # noqa
            part = eng.filter_aggr_partials([("sum", col)], None, None)
            ms = eng.timer_stop()
            best = min(best, ms)
        res[mode] = best
    print(f"{mb:5d} MB  read right after write {mb / 1024 / res['warm-after-write'] * 1e3 / 1e3:6.2f} TB/s ({res['warm-after-write'] * 1e3:7.1f} us)   "
          f"after 4 GB of other reads {mb / 1024 / res['cold'] * 1e3 / 1e3:6.2f} TB/s ({res['cold'] * 1e3:7.1f} us)", flush=True)
