"""Join index / left-join timings (H2O join shape: id1,id2 keys; BASELINE.md lists the reference's published 3149 ms at 1e7 x 1e7)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rayforce_amd.engine import Engine
eng = Engine(0)
for nl, nr in ((10_000_000, 10_000_000), (1_000_000_000, 10_000_000)):
    left = {"id1": eng.gen_i64(nl, 1, 1000), "id2": eng.gen_i64(nl, 2, 20_000), "v": eng.gen_f64(nl, 3)}
    right = {"id1": eng.gen_i64(nr, 4, 1000), "id2": eng.gen_i64(nr, 5, 20_000), "w": eng.gen_f64(nr, 6)}
    for keys in (["id2"], ["id1", "id2"]):
        for fn in ("join_index", "left_join", "inner_join"):
            f = getattr(eng, fn)
            r = f(keys, left, right)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                r = f(keys, left, right)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            rows = (r.numel() if isinstance(r, torch.Tensor) else next(iter(r.values())).numel())
            print(f"left {nl:>11} x right {nr:>9} keys {'+'.join(keys):<8} {fn:<11} {ms:9.2f} ms  ({nl / ms / 1e6:6.2f} G left rows/s, {rows} result rows)", flush=True)
    del left, right
