"""Compile the plan kernels of recurring queries into the on-disk code-object cache WITHOUT a device (rfx_hip_rtc_prewarm_filter_aggr,
rfx_rtc.hip): `__graft_entry__.build()` does it for the BASELINE.json plans, so that the first query of a fresh process on a fresh box
loads its kernel from rayforce_amd/rtc_cache in milliseconds instead of compiling it for seconds (or running the generic kernel).

A plan is described the way Engine.filter_aggr takes it, with column TYPES in place of columns:
    filter_aggr(("<", "a", 100_000), [("sum", "b")], {"a": "i64", "b": "f64"})
Atoms' values do not matter (they are kernel arguments, not part of the compiled text); their types do."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

from . import _lib as L

_TYPES = {"i64": L.RFX_I64, "f64": L.RFX_F64}


def _flat(where):
    if where is None:
        return L.RFX_AND, []
    if where[0] in L.OPS:
        return L.RFX_AND, [(where, 0)]
    top = where[0]
    out = []
    for arm in where[1:]:
        if arm[0] in L.OPS:
            out.append((arm, 0))
        else:  # a parenthesis of the opposite operator over comparisons (rfx_pred_t::more)
            out.extend((s, 1) for s in arm[1:-1])
            out.append((arm[-1], 0))
    return (L.RFX_AND if top == "and" else L.RFX_OR), out


def _preds(where, types, ident):
    logic, preds = _flat(where)
    parr = (L.Pred * max(1, len(preds)))()
    for p, ((op, lhs, rhs), more) in zip(parr, preds):
        p.d_col, p.col_type, p.op, p.more = ident[lhs], _TYPES[types[lhs]], L.OPS[op], more
        if isinstance(rhs, str):
            p.d_rhs_col, p.rhs_type = ident[rhs], _TYPES[types[rhs]]
        elif isinstance(rhs, float):
            p.rhs_type, p.rhs_f = L.RFX_F64, rhs
        else:
            p.rhs_type, p.rhs_i = L.RFX_I64, int(rhs)
    return parr, len(preds), logic


def where(where, types: Dict[str, str]) -> bool:
    """The one-pass `where` kernel of a predicate list (rfx_hip_rtc_prewarm_where): True = its code object is in the cache."""
    lib = L.load_library()
    ident = {name: 4096 * (i + 1) for i, name in enumerate(types)}
    parr, n, logic = _preds(where, types, ident)
    return lib.rfx_hip_rtc_prewarm_where(parr, n, logic) == L.RFX_OK


def filter_aggr(where, aggs: Sequence[Tuple[str, Optional[str]]], types: Dict[str, str]) -> bool:
    """True: the plan's code object is in the cache (already, or compiled now).  False: no run-time compiler / no cache directory."""
    lib = L.load_library()
    ident = {name: 4096 * (i + 1) for i, name in enumerate(types)}  # stand-ins for device pointers: they only tell columns apart
    parr, npred, logic = _preds(where, types, ident)
    aarr = (L.Agg * max(1, len(aggs)))()
    for a, (fn, col) in zip(aarr, aggs):
        a.kind = L.AGGS[fn]
        a.d_col, a.col_type = (ident[col], _TYPES[types[col]]) if col is not None else (None, L.RFX_I64)
    return lib.rfx_hip_rtc_prewarm_filter_aggr(parr, npred, logic, aarr, len(aggs)) == L.RFX_OK


BASELINE_PLANS = {  # BASELINE.json configs (bench.py's workloads of the same names) whose hot kernel is a plan kernel
    "c1": (None, [("sum", "v")], {"v": "f64"}),
    "c2": (("<", "a", 100_000), [("sum", "a")], {"a": "i64"}),
    "c2b": (("<", "a", 100_000), [("sum", "b")], {"a": "i64", "b": "f64"}),
    "c5": (("and", ("<", "a", 0.316228), (">", "b", 0.683772), ("!=", "c", 0.25)), [("avg", "d"), ("min", "d"), ("max", "d")],
           {"a": "f64", "b": "f64", "c": "f64", "d": "f64"}),
}


WHERE_PLANS = {  # bench.py's `where` workloads
    "w2": (("<", "a", 100_000), {"a": "i64"}),
}


def baseline() -> Dict[str, bool]:
    done = {name: filter_aggr(*plan) for name, plan in BASELINE_PLANS.items()}
    done.update({name: where(*plan) for name, plan in WHERE_PLANS.items()})
    return done


def cache_stats() -> Tuple[int, int]:
    """(plans loaded from disk, code objects written) by this process."""
    a, b = C.c_int64(), C.c_int64()
    L.load_library().rfx_hip_rtc_cache_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)
