"""One-off: run the fixed-seed fuzz generator of tests/test_gpu_fuzz.py over many more seeds (python tools/fuzz_more.py 400 3000)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import rfo
from rayforce_amd.engine import Engine
import test_gpu_fuzz as F
from test_gpu_parity import check_select

eng = Engine(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    t, q = F.make_case(rng)
    flags = int(rng.choice([0, 0, 0, 1, 2, 4, 16, 32, 64, 128, 256, 1024, 2048]))
    eng.tune(flags=flags)
    try:
        try:
            rfo.select({"from": t, **q})
        except rfo.NotPerfect:
            continue
        check_select(eng, t, q)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "flags", flags, "n", len(t["k"]), {k: v for k, v in q.items()}, "->", repr(e)[:300], flush=True)
    finally:
        eng.tune(flags=0)
print("done", hi - lo, "cases,", bad, "failures")
