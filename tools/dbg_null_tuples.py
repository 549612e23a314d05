"""One seed of tools/fuzz_null_tuples.py alone (a fresh process: no history in the residency cache), with the columns of small cases printed:
python tools/dbg_null_tuples.py <seed> ...  (DBG_COUNT_ONLY=1: a count per tuple instead of the seed's own aggregates)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import rfo
from rayforce_amd import hostobj as H
import importlib.util
spec = importlib.util.spec_from_file_location("gen_case", os.path.join(ROOT, "tools", "gen_case.py")); G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
ops = H.lib(); ops.rfx_host_bind()
for seed in map(int, sys.argv[1:]):
    n, t, by, kinds, q = G.gen(seed)
    if os.environ.get("DBG_COUNT_ONLY"): q = {"c": ("count", "a"), "by": by}
    print(q)
    want = rfo.select({"from": t, **q})
    tab = H.table(t); d = H.select_dict(q, tab)
    r = ops.rfx_select(d)
    if H.is_error(r):
        print(seed, "ERROR", H.error_text(r)); continue
    got = H.table_to_numpy(r)
    print("seed", seed, "n", n, kinds, "groups got", len(got[next(iter(by))]), "want", len(want[next(iter(by))]), "path stats", [int(ops.rfx_exec_stat(ops.rfx_ops_exec(), i)) for i in range(9)] if hasattr(ops, "rfx_exec_stat") else "")
    if n <= 4:
        for k in t: print("   col", k, t[k])
        for k in want: print("   ", k, "got", got[k], "want", want[k])
    else:
        gt = set(zip(*[got[g].tolist() for g in by])); wt = set(zip(*[want[g].tolist() for g in by]))
        print("   distinct got tuples", len(gt), "want", len(wt), "got-want", len(gt - wt), "want-got", len(wt - gt), "sum counts", len(got[next(iter(by))]))
        ex = list(gt - wt)[:3]; print("   extra examples", ex); print("   missing examples", list(wt - gt)[:3])
