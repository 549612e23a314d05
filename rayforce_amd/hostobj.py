"""ctypes view of the operator-level C ABI (include/rfx_ops.h) and of the standalone host object model (rfx_host.c).

Used by the tests to drive ``rfx_select`` & friends exactly as the reference's evaluator would: with obj_p arguments
laid out like RayforceDB objects (include/rfx_abi.h).  Nothing here computes anything.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence

import numpy as np

from . import _lib as L

T_LIST, T_B8, T_I64, T_SYMBOL, T_F64, T_TABLE, T_DICT, T_ERR = 0, 1, 5, 6, 10, 98, 99, 127
NP_OF = {T_B8: np.int8, T_I64: np.int64, T_SYMBOL: np.int64, T_F64: np.float64, 9: np.int64}
TYPE_OF = {np.dtype(np.int8): T_B8, np.dtype(np.bool_): T_B8, np.dtype(np.int64): T_I64, np.dtype(np.float64): T_F64}

OPS_PROTOTYPES = {
    "rfx_host_bind": (C.c_int, []),
    "rfx_ops_set_device": (C.c_int, [C.c_int]),
    "rfx_ops_last_error": (C.c_char_p, []),
    "rfx_ops_set_shards": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int]),
    "rfx_ops_shards": (C.c_int, []),
    "rfx_ops_set_validation": (C.c_int, [C.c_int]),
    "rfx_ops_set_deterministic": (C.c_int, [C.c_int]),
    "rfx_ops_set_rank_slices": (C.c_int, [C.c_int]),
    "rfx_ops_dist_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "rfx_ops_dist_finalize": (C.c_int, []),
    "rfx_ops_exec": (C.c_void_p, []),
    "rfx_host_device_vector": (C.c_void_p, [C.c_int8, C.c_int64, C.POINTER(C.c_void_p), C.c_int]),
    "rfx_select": (C.c_void_p, [C.c_void_p]),
    "rfx_update": (C.c_void_p, [C.c_void_p]),
    **{f"rfx_{n}": (C.c_void_p, [C.c_void_p, C.c_void_p]) for n in ("eq", "ne", "lt", "gt", "le", "ge", "at", "add", "sub", "mul", "div", "floordiv", "mod")},
    "rfx_and": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int64]),
    "rfx_or": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int64]),
    "rfx_and_sf": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int64]),
    "rfx_or_sf": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int64]),
    "rfx_left_join": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int64]),
    "rfx_inner_join": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int64]),
    **{f"rfx_{n}": (C.c_void_p, [C.c_void_p]) for n in ("where", "sum", "avg", "min", "max", "count", "first", "pin", "unpin", "invalidate", "stats", "group")},
    "rfx_cache_clear": (None, []),
    "rfx_cache_bytes": (C.c_int64, []),
    "rfx_last_select_on_gpu": (C.c_int, []),
    "rfx_host_vector": (C.c_void_p, [C.c_int8, C.c_int64]),
    "rfx_host_i64": (C.c_void_p, [C.c_int64]),
    "rfx_host_f64": (C.c_void_p, [C.c_double]),
    "rfx_host_symbol": (C.c_void_p, [C.c_char_p]),
    "rfx_host_list": (C.c_void_p, [C.c_int64]),
    "rfx_host_table": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "rfx_host_dict": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "rfx_host_fn": (C.c_void_p, [C.c_char_p]),
    "rfx_host_clone": (C.c_void_p, [C.c_void_p]),
    "rfx_host_drop": (None, [C.c_void_p]),
    "rfx_host_intern": (C.c_int64, [C.c_char_p, C.c_int64]),
    "rfx_host_symbol_name": (C.c_char_p, [C.c_int64]),
    "rfx_host_error_text": (C.c_char_p, [C.c_void_p]),
}


class Header(C.Structure):
    _fields_ = [("mmod", C.c_uint8), ("order", C.c_uint8), ("type", C.c_int8), ("attrs", C.c_uint8), ("rc", C.c_uint32), ("len", C.c_int64)]


_bound = None


def lib() -> C.CDLL:
    global _bound
    if _bound is None:
        l = L.load_library()
        for name, (res, args) in OPS_PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _bound = l
    return _bound


def header(o: int) -> Header:
    return Header.from_address(o)


def payload(o: int) -> int:
    return o + 16


def vector(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    t = TYPE_OF[a.dtype]
    o = lib().rfx_host_vector(t, a.size)
    if a.size:
        C.memmove(payload(o), a.ctypes.data, a.nbytes)
    return o


def to_numpy(o: int) -> np.ndarray:
    h = header(o)
    dt = NP_OF[h.type]
    return np.frombuffer((C.c_char * (h.len * np.dtype(dt).itemsize)).from_address(payload(o)), dtype=dt).copy() if h.len else np.empty(0, dt)


def list_of(items: Sequence[int]) -> int:
    o = lib().rfx_host_list(len(items))
    arr = (C.c_void_p * len(items)).from_address(payload(o))
    for i, it in enumerate(items):
        arr[i] = it
    return o


def list_items(o: int):
    h = header(o)
    return list((C.c_void_p * h.len).from_address(payload(o)))


def symbols(names: Sequence[str]) -> int:
    l = lib()
    o = l.rfx_host_vector(T_SYMBOL, len(names))
    arr = (C.c_int64 * len(names)).from_address(payload(o))
    for i, n in enumerate(names):
        arr[i] = l.rfx_host_intern(n.encode(), len(n))
    return o


def table(cols: Dict[str, np.ndarray]) -> int:
    return lib().rfx_host_table(symbols(list(cols)), list_of([vector(v) for v in cols.values()]))


def device_vector(t, ptrs=None) -> int:
    """A DEVICE column handle (rfx_host_device_vector): a torch CUDA tensor's cells handed to the operators where they are.  `ptrs`: one
    device address per shard (columns kept shard by shard on several devices); default: the tensor's one allocation."""
    import torch
    tp = {torch.int64: T_I64, torch.float64: T_F64, torch.int8: T_B8}[t.dtype]
    ps = [t.data_ptr()] if ptrs is None else list(ptrs)
    return lib().rfx_host_device_vector(tp, t.numel(), (C.c_void_p * len(ps))(*ps), len(ps))


def device_table(cols) -> int:
    """dict name -> CUDA tensor as a table object whose columns are device handles (the tensors must outlive the table)."""
    return lib().rfx_host_table(symbols(list(cols)), list_of([device_vector(v) for v in cols.values()]))


def atom(x) -> int:
    l = lib()
    if isinstance(x, str):
        return l.rfx_host_symbol(x.encode())
    if isinstance(x, (int, np.integer)):
        return l.rfx_host_i64(int(x))
    return l.rfx_host_f64(float(x))


def expr(e) -> int:
    """('<', 'a', 5) / ('and', e1, e2) / ('sum', 'v') -> LIST [function object, args...] as the reference's parser builds it."""
    if isinstance(e, tuple):
        return list_of([lib().rfx_host_fn(e[0].encode())] + [expr(x) for x in e[1:]])
    return atom(e)


def select_dict(query: Dict, tab: int) -> int:
    """{name: ('sum', 'v'), 'where': (...), 'by': 'k'} + table object -> the select DICT."""
    keys, vals = [], []
    for k, v in query.items():
        keys.append(k)
        if k == "by" and isinstance(v, dict):  # by: {name: column ...} -> DICT of SYMBOL keys, LIST of symbol atoms
            vals.append(lib().rfx_host_dict(symbols(list(v.keys())), list_of([expr(c) for c in v.values()])))  # column symbol or (xbar col w)
        else:
            vals.append(atom(v) if k == "by" else expr(v))
    keys.append("from")
    vals.append(lib().rfx_host_clone(tab))
    return lib().rfx_host_dict(symbols(keys), list_of(vals))


def table_to_numpy(o: int) -> Dict[str, np.ndarray]:
    l = lib()
    keys, vals = list_items(o)
    names = [l.rfx_host_symbol_name(int(i)).decode() for i in to_numpy(keys)]
    return {n: to_numpy(c) for n, c in zip(names, list_items(vals))}


def is_error(o: int) -> bool:
    return header(o).type == T_ERR


def error_text(o: int) -> str:
    return lib().rfx_host_error_text(o).decode(errors="replace")
