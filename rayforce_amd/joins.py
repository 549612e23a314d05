"""lj / ij over device columns (ray_left_join / ray_inner_join, core/join.c:158-298): the join INDEX is the planner's
(rfx_exec_join_index); what is here is the reference's column rule for the result -- keys, then the other left columns, then the
right-only ones -- assembled with one gather per column."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib as L
from ._lib import RfxError

def join_index(eng, keys, left: Dict[str, torch.Tensor], right: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Per LEFT row the FIRST right row with an equal key tuple, or null (index_left_join_obj, core/index.c:2886-2928): the planner's
    rfx_exec_join_index (dense / hashed build side, composite key or row hash + tuple check)."""
    keys = [keys] if isinstance(keys, str) else list(keys)
    lk = [eng._check_col(eng._resolve(k, left)) for k in keys]
    rk = [eng._check_col(eng._resolve(k, right)) for k in keys]
    if any(c.dtype != torch.int64 for c in lk + rk):
        raise RfxError("join keys must be i64-like columns on this path")
    nl, nr = lk[0].numel(), rk[0].numel()
    ids = eng.empty(nl)
    k = len(keys)
    col = C.c_int(0)
    rc = eng.lib.rfx_exec_join_index(eng._x, (C.c_void_p * k)(*[c.data_ptr() for c in lk]), (C.c_void_p * k)(*[c.data_ptr() for c in rk]), k, nl, nr,
                                      ids.data_ptr(), C.byref(col))
    if rc != L.RFX_OK and col.value:
        raise RfxError("join: two key tuples share one 64-bit row hash (collision); not answered on this path")
    eng._xcheck(rc, "join_index")
    return ids

def left_join(eng, keys, left: Dict[str, torch.Tensor], right: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``(lj [keys] left right)`` -- ray_left_join, core/join.c:158-198: every left row; a non-key column that the right table has takes
    the matched right row's value, else the left row's own (null when the left table lacks the column); columns: keys, then the other
    left columns, then the right-only ones.  Empty side -> the left table."""
    keys = [keys] if isinstance(keys, str) else list(keys)
    nl = next(iter(left.values())).numel() if left else 0
    nr = next(iter(right.values())).numel() if right else 0
    if nl == 0 or nr == 0:
        return dict(left)
    ids = join_index(eng, keys, left, right)
    out = {k: left[k] for k in keys}
    for name in [c for c in left if c not in keys] + [c for c in right if c not in keys and c not in left]:
        if name not in right:
            out[name] = left[name]
            continue
        rc, lc = right[name], left.get(name)
        if lc is not None and lc.dtype != rc.dtype:
            raise RfxError(f"join: column {name} has different types in the two tables")
        o = torch.empty(nl, dtype=rc.dtype, device=eng.device)
        fill = 0x7FF8000000000000 if rc.dtype == torch.float64 else (1 << 63)  # NaN / NULL_I64 bit patterns
        L.check(eng.lib.rfx_hip_gather_or(eng._ctx, rc.data_ptr(), lc.data_ptr() if lc is not None else None, ids.data_ptr(), nl, fill, o.data_ptr()), "gather_or")
        out[name] = o
    eng.sync()
    return out

def inner_join(eng, keys, left: Dict[str, torch.Tensor], right: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``(ij [keys] left right)`` -- ray_inner_join, core/join.c:200-298: the left rows that have a match, in left order, paired with
    their first matching right row; a column the right table has comes from the right row."""
    keys = [keys] if isinstance(keys, str) else list(keys)
    nl = next(iter(left.values())).numel() if left else 0
    nr = next(iter(right.values())).numel() if right else 0
    if nl == 0 or nr == 0:
        return dict(left)
    ids = join_index(eng, keys, left, right)
    lids = eng.where(("!=", ids, None))  # ascending left rows with a match
    rids = eng.at_ids(ids, lids)
    out = {}
    for name in keys + [c for c in left if c not in keys] + [c for c in right if c not in keys and c not in left]:
        if name in right:
            if name in left and left[name].dtype != right[name].dtype:
                raise RfxError(f"join: column {name} has different types in the two tables")
            out[name] = eng.at_ids(right[name], rids)
        else:
            out[name] = eng.at_ids(left[name], lids)
    return out

