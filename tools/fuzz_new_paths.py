"""More seeds for the row-hash key-tuple and join fuzz tests: python tools/fuzz_new_paths.py <lo> <hi>"""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rayforce_amd.engine import Engine
import test_gpu_fuzz as F
lo, hi = int(sys.argv[1]), int(sys.argv[2])
eng = Engine(0)
bad = 0
for seed in range(lo, hi):
    for fn in (F.test_random_key_tuples_row_hash, F.test_random_joins):
        try:
            fn.__wrapped__(eng, seed) if hasattr(fn, "__wrapped__") else fn(eng, seed)
        except Exception:  # noqa: BLE001
            bad += 1
            print("FAIL", fn.__name__, seed)
            traceback.print_exc(limit=2)
print(f"done seeds {lo}..{hi}: {bad} failures")
