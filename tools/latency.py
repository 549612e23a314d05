"""tools/latency.py -- per-query latency of small and medium tables through the operator ABI (rfx_select, standalone host object
model) and through the flat ABI (Engine): where the fixed costs of a GPU query sit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rfo
from rayforce_amd import hostobj as H
from rayforce_amd.engine import Engine

ops = H.lib()
ops.rfx_host_bind()
eng = Engine(0)
for n in (10_000, 100_000, 1_000_000, 10_000_000):
    host = {"k": rfo.gen_i64(n, 4, 1000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5)}
    tab = H.table(host)
    ops.rfx_host_drop(ops.rfx_pin(tab))  # resident and trusted: without it every call re-validates the columns with a full checksum
    dev = {k: eng.column(v) for k, v in host.items()}
    for name, q in (("where-sum", {"s": ("sum", "v"), "where": ("<", "a", 100_000)}), ("by-sum", {"s": ("sum", "v"), "by": "k"}),
                    ("where-by", {"s": ("sum", "v"), "c": ("count", "a"), "where": ("<", "a", 500_000), "by": "k"})):
        d = H.select_dict(q, tab)
        for _ in range(3):
            ops.rfx_host_drop(ops.rfx_select(d))
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.rfx_host_drop(ops.rfx_select(d))
        t_ops = (time.perf_counter() - t0) / reps * 1e6
        for _ in range(3):
            eng.select({"from": dev, **q})
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.select({"from": dev, **q})
        t_eng = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        for _ in range(5):
            rfo.select({"from": host, **q})
        t_cpu = (time.perf_counter() - t0) / 5 * 1e6
        print(f"n {n:>9} {name:<9} rfx_select {t_ops:8.0f} us   Engine.select {t_eng:8.0f} us   CPU oracle ({rfo.lib().rfo_get_threads()} thr) {t_cpu:8.0f} us", flush=True)
        ops.rfx_host_drop(d)
    ops.rfx_host_drop(tab)
