"""Test plumbing for the FLAT device ABI (include/rfx_hip.h): table sets a caller of rfx_hip_group_* must provide.  The product's own
callers never build these by hand -- the planner (rfx_exec.c) does; tests of the flat entry points do."""
import ctypes as C
from typing import Optional

import torch

from rayforce_amd import _lib as L


def group_tables(eng, aggs_arr, nagg: int, kmin: int, rng: int, hashed: bool = False, store: Optional[torch.Tensor] = None):
    """Allocate (or wrap `store`) one table set.  Returns (struct, backing tensor [n_arrays, cells], layout)."""
    n_arr = C.c_int()
    L.check(eng.lib.rfx_hip_group_table_arrays(aggs_arr, nagg, C.byref(n_arr)), "group_table_arrays")
    cells = rng + 1 if hashed else rng
    total = n_arr.value + (1 if hashed else 0)
    if store is None:
        store = torch.empty((total, cells), dtype=torch.int64, device=eng.device)
    t = L.HashTables() if hashed else L.GroupTables()
    k = 0
    if hashed:
        t.capacity = rng
        t.d_keys = store[k].data_ptr(); k += 1
    else:
        t.kmin, t.range = kmin, rng
    t.nagg = nagg
    t.d_first = store[k].data_ptr(); k += 1
    layout = [("first", None)]
    for a in range(nagg):
        t.d_acc[a] = store[k].data_ptr(); k += 1
        layout.append(("acc", a))
        kind, f64 = aggs_arr[a].kind, L.agg_input_type(aggs_arr[a]) == L.RFX_F64
        if kind == L.RFX_AGG_AVG or (kind == L.RFX_AGG_SUM and not f64):
            t.d_cnt[a] = store[k].data_ptr(); k += 1
            layout.append(("cnt", a))
        else:
            t.d_cnt[a] = None
    return t, store, layout

