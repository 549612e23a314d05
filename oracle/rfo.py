"""oracle/rfo.py -- numpy front-end of the CPU restatement (oracle/librfo.so).

*** TEST INFRASTRUCTURE ONLY ***  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
never by rayforce_amd.  It composes the C restatement in the reference's own pass order:

    where:  predicate -> B8 mask (cmp_map) -> and/or in place (logic_map) -> ops_where -> ascending ids
    no by:  every aggregated column is GATHERED through the ids (filter_collect / at_ids) and the copy is folded
            (ray_sum_partial's TYPE_MAPFILTER arm, core/math.c:1874-1890)
    by:     index_scope_i64 -> dense first-occurrence index (range <= rows) or open-addressed hash -> AGGR_ITER partials
            -> AGGR_COLLECT, group key column = key[filter[first_ids]] (core/query.c:63-75)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NULL_I64 = np.int64(-(2**63))
OPS = {"==": 0, "!=": 1, "<": 2, ">": 3, "<=": 4, ">=": 5}
_I64, _F64 = 5, 10


def build() -> str:
    path = os.path.join(_HERE, "librfo.so")
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    return path


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "librfo.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        p, i64, f64 = C.c_void_p, C.c_int64, C.c_double
        sig = {
            "rfo_set_threads": (None, [C.c_int]), "rfo_get_threads": (C.c_int, []),
            "rfo_pool_split_by_mem": (i64, [i64, i64, i64]), "rfo_pool_chunk_aligned": (i64, [i64, i64, i64]),
            "rfo_gen_i64": (None, [p, i64, C.c_uint64, i64, C.c_uint64]), "rfo_gen_f64": (None, [p, i64, C.c_uint64, i64]),
            "rfo_cmp": (C.c_int, [C.c_int, C.c_int, p, C.c_int, C.c_int, p, C.c_int, i64, p]),
            "rfo_logic": (None, [C.c_int, p, p, C.c_int, i64]),
            "rfo_where": (i64, [p, i64, p]), "rfo_at_ids": (None, [p, p, i64, p]),
            "rfo_sum_i64": (i64, [p, i64]), "rfo_sum_f64": (f64, [p, i64]),
            "rfo_min_i64": (i64, [p, i64]), "rfo_max_i64": (i64, [p, i64]),
            "rfo_min_f64": (f64, [p, i64]), "rfo_max_f64": (f64, [p, i64]),
            "rfo_cnt_i64": (i64, [p, i64]), "rfo_cnt_f64": (i64, [p, i64]),
            "rfo_avg_i64": (f64, [p, i64]), "rfo_avg_f64": (f64, [p, i64]),
            "rfo_scope_i64": (None, [p, p, i64, C.POINTER(i64), C.POINTER(i64)]),
            "rfo_group_dense": (i64, [p, p, i64, i64, i64, p, p, p]),
            "rfo_group_sparse": (i64, [p, p, i64, p, p]),
            "rfo_hash_fnv1a": (C.c_uint64, [i64]), "rfo_hash_index_u64": (C.c_uint64, [C.c_uint64, C.c_uint64]),
            "rfo_aggr_first": (None, [p, p, p, i64, p]),
            "rfo_binop": (C.c_int, [C.c_int, C.c_int, p, C.c_int, C.c_int, p, C.c_int, i64, p]),
            "rfo_aggr_fold": (None, [C.c_int, C.c_int, p, p, i64, i64, p]),
            "rfo_xbar_i64": (None, [p, i64, i64, p]),
            "rfo_composite_plan": (C.c_int, [p, p, C.c_int, p, C.POINTER(i64)]),
            "rfo_composite_key": (None, [p, p, p, C.c_int, p, i64, p]),
        }
        for fn in ("sum", "min", "max", "count", "avg"):
            for t in ("i64", "f64"):
                sig[f"rfo_aggr_{fn}_{t}"] = (None, [p, p, p, i64, i64, p])
        for name, (res, args) in sig.items():
            f = getattr(_lib, name)
            f.restype, f.argtypes = res, args
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _col(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.int64, np.float64):
        raise TypeError(f"unsupported dtype {a.dtype}")
    return a


def _t(a) -> str:
    return "i64" if a.dtype == np.int64 else "f64"


def set_threads(n: int) -> None:
    lib().rfo_set_threads(int(n))


def gen_i64(n, seed, modulus, row0=0):
    out = np.empty(n, np.int64)
    lib().rfo_gen_i64(_ptr(out), n, seed, row0, modulus)
    return out


def gen_f64(n, seed, row0=0):
    out = np.empty(n, np.float64)
    lib().rfo_gen_f64(_ptr(out), n, seed, row0)
    return out


# ---------------------------------------------------------------- operators (names of the reference)
def cmp(op: str, lhs, rhs) -> np.ndarray:
    """ray_eq..ray_ge: vector (x) atom | atom (x) vector | vector (x) vector -> B8 mask."""
    def prep(x):
        if isinstance(x, np.ndarray):
            return _col(x), 0
        if x is None:
            return np.array([NULL_I64], np.int64), 1
        if isinstance(x, (int, np.integer)):
            return np.array([x], np.int64), 1
        return np.array([x], np.float64), 1
    l, la = prep(lhs)
    r, ra = prep(rhs)
    n = len(l) if not la else (len(r) if not ra else 1)
    if not la and not ra and len(l) != len(r):
        raise ValueError("length")
    out = np.empty(n, np.int8)
    rc = lib().rfo_cmp(OPS[op], _I64 if l.dtype == np.int64 else _F64, _ptr(l), la, _I64 if r.dtype == np.int64 else _F64, _ptr(r), ra, n, _ptr(out))
    if rc:
        raise TypeError("type")
    return out


def and_(*masks):
    acc = np.array(masks[0], np.int8, copy=True)
    for m in masks[1:]:
        m = np.ascontiguousarray(m, np.int8)
        lib().rfo_logic(0, _ptr(acc), _ptr(m), 0, len(acc))
    return acc


def or_(*masks):
    acc = np.array(masks[0], np.int8, copy=True)
    for m in masks[1:]:
        m = np.ascontiguousarray(m, np.int8)
        lib().rfo_logic(1, _ptr(acc), _ptr(m), 0, len(acc))
    return acc


def where(mask) -> np.ndarray:
    mask = np.ascontiguousarray(mask, np.int8)
    n = lib().rfo_where(_ptr(mask), len(mask), None)
    ids = np.empty(n, np.int64)
    lib().rfo_where(_ptr(mask), len(mask), _ptr(ids))
    return ids


def at_ids(col, ids):
    col = _col(col)
    out = np.empty(len(ids), col.dtype)
    lib().rfo_at_ids(_ptr(col), _ptr(np.ascontiguousarray(ids, np.int64)), len(ids), _ptr(out))
    return out


def fold(fn: str, col):
    """Scalar ray_sum / ray_min / ray_max / ray_avg / ray_count of a materialised column.  None = null."""
    col = _col(col)
    t, n, L = _t(col), len(col), lib()
    if fn == "count":
        return n  # ops_count: length, nulls included (core/misc.c:43-60)
    if fn == "first":
        if n == 0:
            return None if t == "i64" else float("nan")
        v = col[0]
        return (None if v == NULL_I64 else int(v)) if t == "i64" else float(v)
    r = getattr(L, f"rfo_{fn}_{t}")(_ptr(col), n)
    if fn == "avg" or t == "f64":
        return float(r)
    return None if r == NULL_I64 else int(r)


def mask_of(where_spec, table) -> np.ndarray:
    head = where_spec[0]
    if head in OPS:
        _, lhs, rhs = where_spec
        # an operand may be an element-wise expression: the reference evaluates it first (eval -> binop_map), then compares
        lhs = table[lhs] if isinstance(lhs, str) else (eval_arg(lhs, table) if isinstance(lhs, tuple) else lhs)
        rhs = table[rhs] if isinstance(rhs, str) else (eval_arg(rhs, table) if isinstance(rhs, tuple) else rhs)
        return cmp(head, lhs, rhs)
    subs = [mask_of(w, table) for w in where_spec[1:]]
    return and_(*subs) if head == "and" else or_(*subs)


class NotPerfect(Exception):
    """index_group_list_perfect returned NULL_OBJ: the reference takes its row-hash path (group_rows below)."""


U64_HASH_SEED = np.uint64(0x9DDFEA08EB382D69)  # core/hash.h:35


def hash_index_u64(h, k):
    """hash_index_u64(h, k), core/hash.h:86-97, over uint64 arrays (wrapping arithmetic)."""
    h, k = np.asarray(h, np.uint64), np.asarray(k, np.uint64)
    with np.errstate(over="ignore"):
        a = (h ^ k) * U64_HASH_SEED
        a ^= a >> np.uint64(47)
        b = (((k << np.uint64(31)) | (k >> np.uint64(33))) ^ a) * U64_HASH_SEED
        b ^= b >> np.uint64(47)
        b *= U64_HASH_SEED
    return b


def row_hash(cols, filter_ids=None) -> np.ndarray:
    """__index_list_precalc_hash (core/index.c:274-309): start from U64_HASH_SEED, fold every key column in.  Unfiltered
    i64-like columns go through hash_index_i64_batch = hash_index_u64(running, value) (core/hash.h:130-143); filtered rows and
    f64 columns through hash_index_u64(value, running) (index_hash_obj_partial, core/index.c:155-175)."""
    cols = [_col(c) for c in cols]
    n = len(cols[0]) if filter_ids is None else len(filter_ids)
    h = np.full(n, U64_HASH_SEED, np.uint64)
    for c in cols:
        v = (c if filter_ids is None else c[filter_ids]).view(np.uint64)
        h = hash_index_u64(v, h) if (filter_ids is not None or c.dtype == np.float64) else hash_index_u64(h, v)
    return h.view(np.int64)


def group_rows(keys, filter_ids=None, order="first"):
    """index_group_list after the perfect path gave up (core/index.c:2731-2790): rows with bitwise-equal key tuples
    (__index_list_cmp_row, :59-104) form a group.  order="first": the single-threaded arm (:2760-2781), groups in first
    occurrence order; order="radix": the multi-threaded arm (index_group_list_radix, :2556-2729), groups ordered by
    (row hash & 1023, first occurrence).  Returns (group id per selected row, first selected row per group, groups)."""
    cols = [_col(k) if filter_ids is None else _col(k)[filter_ids] for k in keys]
    n = len(cols[0])
    if n == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64), 0
    rows = np.stack([c.view(np.int64) for c in cols], axis=1)
    _, first, inv = np.unique(rows, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    if order == "radix":
        h = row_hash(keys, filter_ids)
        rank = np.lexsort((first, h[first] & 1023))
    else:
        rank = np.argsort(first, kind="stable")
    new_id = np.empty(len(first), np.int64)
    new_id[rank] = np.arange(len(first))
    return new_id[inv], first[rank].astype(np.int64), len(first)


def composite_key(keys, filter_ids=None):
    """index_group_list_perfect (core/index.c:2308-2424): (composite key per SELECTED row, total_max, mins, mults)."""
    L = lib()
    cols = [_col(k) for k in keys]
    idx = None if filter_ids is None else np.ascontiguousarray(filter_ids, np.int64)
    n = len(cols[0]) if idx is None else len(idx)
    if n == 0:
        return np.empty(0, np.int64), -1, [], []
    mins, maxs = [], []
    for c in cols:
        if c.dtype != np.int64:
            raise NotPerfect("non-integer key column")
        mn, mx = C.c_int64(), C.c_int64()
        L.rfo_scope_i64(_ptr(c), _ptr(idx), n, C.byref(mn), C.byref(mx))
        mins.append(mn.value)
        maxs.append(mx.value)
    k = len(cols)
    amin, amax, amul = (C.c_int64 * k)(*mins), (C.c_int64 * k)(*maxs), (C.c_int64 * k)()
    tmax = C.c_int64()
    if not L.rfo_composite_plan(amin, amax, k, amul, C.byref(tmax)):
        raise NotPerfect("key ranges overflow the composite key")
    out = np.empty(n, np.int64)
    ptrs = (C.c_void_p * k)(*[c.ctypes.data for c in cols])
    L.rfo_composite_key(ptrs, amin, amul, k, _ptr(idx), n, _ptr(out))
    return out, tmax.value, mins, list(amul)


XOPS = {"+": 1, "-": 2, "*": 3, "div": 4, "/": 5, "%": 6}


def binop(op: str, lhs, rhs) -> np.ndarray:
    """ray_add / ray_sub / ray_mul / ray_fdiv over i64 / f64 vectors and atoms (one side must be a vector)."""
    def prep(x):
        if isinstance(x, np.ndarray):
            a = _col(x)
            return a, 0, (10 if a.dtype == np.float64 else 5)
        if isinstance(x, float):
            return np.array([x], np.float64), 1, 10
        return np.array([int(x)], np.int64), 1, 5
    la, l_atom, lt = prep(lhs)
    ra, r_atom, rt = prep(rhs)
    n = len(la) if not l_atom else len(ra)
    out_f64 = (lt == 10) if op == "/" else (op == "div" or lt == 10 or rt == 10)  # `/` keeps the left operand's type
    out = np.empty(n, np.float64 if out_f64 else np.int64)
    lib().rfo_binop(XOPS[op], lt, _ptr(la), l_atom, rt, _ptr(ra), r_atom, n, _ptr(out))
    return out


def xbar(col, width: int) -> np.ndarray:
    """(xbar col width) over an i64 column."""
    c = _col(col)
    out = np.empty(len(c), np.int64)
    lib().rfo_xbar_i64(_ptr(c), len(c), int(width), _ptr(out))
    return out


def key_column(spec, table):
    """A `by:` entry: a column name, or ("xbar", column, width)."""
    if isinstance(spec, tuple):
        assert spec[0] == "xbar", spec
        return xbar(table[spec[1]], spec[2])
    return table[spec]


def eval_arg(arg, table):
    """An aggregate's argument: a column name, or (op, lhs, rhs) with operands column names / atoms."""
    if isinstance(arg, tuple):
        op, l, r = arg
        ev = lambda x: eval_arg(x, table) if isinstance(x, (tuple, str)) else x  # nested expressions compose: temporaries, as the reference
        return binop(op, ev(l), ev(r))
    return table[arg]


def aggr_fold(fn: str, vals, gids, groups):
    """Expression argument under `by:`: per-group vectors folded with the scalar rules (rfo_aggr_fold)."""
    vals = _col(vals)
    out = np.empty(groups, np.float64 if (fn == "avg" or vals.dtype == np.float64) else np.int64)
    lib().rfo_aggr_fold({"sum": 0, "min": 1, "max": 2, "avg": 3}[fn], 10 if vals.dtype == np.float64 else 5, _ptr(vals),
                        _ptr(np.ascontiguousarray(gids, np.int64)), len(gids), groups, _ptr(out))
    return out


def group_index(key, filter_ids=None, scope=None):
    """index_group_i64: returns (gids per selected row, first positions, groups, dense?).  `scope=(min, max)` forces the
    key scope instead of scanning for it (index_group_i64_scoped called by index_group_list_perfect)."""
    key = _col(key)
    L = lib()
    idx = None if filter_ids is None else np.ascontiguousarray(filter_ids, np.int64)
    n = len(key) if idx is None else len(idx)
    if n == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64), 0, True
    mn, mx = C.c_int64(), C.c_int64()
    if scope is None:
        L.rfo_scope_i64(_ptr(key), _ptr(idx), n, C.byref(mn), C.byref(mx))
    else:
        mn.value, mx.value = scope
    rng = mx.value - mn.value + 1
    gids = np.empty(n, np.int64)
    firsts = np.empty(n, np.int64)
    if 0 < rng <= n and mn.value != int(NULL_I64):
        hk = np.empty(rng, np.int64)
        g = L.rfo_group_dense(_ptr(key), _ptr(idx), n, mn.value, rng, _ptr(hk), _ptr(firsts), _ptr(gids))
        return gids, firsts[:g].copy(), g, True
    g = L.rfo_group_sparse(_ptr(key), _ptr(idx), n, _ptr(gids), _ptr(firsts))
    return gids, firsts[:g].copy(), g, False


def aggr(fn: str, col, gids, filter_ids, groups):
    col = _col(col)
    t = _t(col)
    idx = None if filter_ids is None else np.ascontiguousarray(filter_ids, np.int64)
    out = np.empty(groups, np.float64 if (fn == "avg" or (t == "f64" and fn != "count")) else np.int64)
    getattr(lib(), f"rfo_aggr_{fn}_{t}")(_ptr(col), _ptr(gids), _ptr(idx), len(gids), groups, _ptr(out))
    return out


def join_index(keys, left: dict, right: dict) -> np.ndarray:
    """index_left_join_obj (core/index.c:2886-2928; one key: ray_find): per left row the FIRST right row whose key tuple is
    bitwise equal, else null."""
    keys = [keys] if isinstance(keys, str) else list(keys)
    lk = np.stack([_col(left[k]).view(np.int64) for k in keys], axis=1)
    rk = np.stack([_col(right[k]).view(np.int64) for k in keys], axis=1)
    nl, nr = len(lk), len(rk)
    if nl == 0 or nr == 0:
        return np.full(nl, NULL_I64, np.int64)
    _, inv = np.unique(np.concatenate([rk, lk]), axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    big = np.iinfo(np.int64).max
    first = np.full(int(inv.max()) + 1, big, np.int64)
    np.minimum.at(first, inv[:nr], np.arange(nr, dtype=np.int64))
    ids = first[inv[nr:]]
    ids[ids == big] = NULL_I64
    return ids


def _null_of(dtype):
    return np.nan if dtype == np.float64 else NULL_I64


def left_join(keys, left: dict, right: dict) -> dict:
    """ray_left_join + __left_join_inner + select_column (core/join.c:38-66,83-198)."""
    keys = [keys] if isinstance(keys, str) else list(keys)
    nl = len(next(iter(left.values()))) if left else 0
    nr = len(next(iter(right.values()))) if right else 0
    if nl == 0 or nr == 0:
        return dict(left)
    ids = join_index(keys, left, right)
    hit = ids != NULL_I64
    out = {k: _col(left[k]) for k in keys}
    for name in [c for c in left if c not in keys] + [c for c in right if c not in keys and c not in left]:
        if name not in right:
            out[name] = _col(left[name])
            continue
        rc = _col(right[name])
        lc = _col(left[name]) if name in left else np.full(nl, _null_of(rc.dtype), rc.dtype)
        o = lc.copy()
        o[hit] = rc[ids[hit]]
        out[name] = o
    return out


def inner_join(keys, left: dict, right: dict) -> dict:
    """ray_inner_join + index_inner_join_obj + get_column (core/join.c:68-81,200-298; core/index.c:2930-2990)."""
    keys = [keys] if isinstance(keys, str) else list(keys)
    nl = len(next(iter(left.values()))) if left else 0
    nr = len(next(iter(right.values()))) if right else 0
    if nl == 0 or nr == 0:
        return dict(left)
    ids = join_index(keys, left, right)
    lids = np.nonzero(ids != NULL_I64)[0]
    rids = ids[lids]
    out = {}
    for name in keys + [c for c in left if c not in keys] + [c for c in right if c not in keys and c not in left]:
        out[name] = _col(right[name])[rids] if name in right else _col(left[name])[lids]
    return out


def update(query: dict) -> dict:
    """ray_update (core/update.c:936-1106) restated over numpy columns: `where:` -> row ids (ray_where, :1001); the mappings are
    evaluated over the filtered table (remap_filter, :1036-1042: value i belongs to row ids[i]) or, with `by:`, as one aggregate per
    group of the selected rows (index_group(groupby, filters), :1021-1023); __update_table (:753-935) then writes value i to row
    ids[i] / each group's value to all of its rows (aggr_row + set_ids), a name the table lacks becoming a new column that is null
    elsewhere (nullv).  Returns the new table (the value form, `from: t`)."""
    table = query["from"]
    where_spec, by = query.get("where"), query.get("by")
    outs = [(k, v) for k, v in query.items() if k not in ("from", "where", "by")]
    n = len(next(iter(table.values())))
    ids = where(mask_of(where_spec, table)) if where_spec is not None else np.arange(n, dtype=np.int64)
    res = dict(table)
    if by is not None:
        gids, firsts, groups, _ = group_index(table[by], ids if where_spec is not None else None)
    for name, m in outs:
        if by is not None:
            fn, col = m
            if fn == "first":
                c = _col(table[col])
                per_group = c[ids[firsts]] if groups else np.empty(0, c.dtype)
            else:
                per_group = aggr(fn, table[col], gids, ids if where_spec is not None else None, groups)
            vals = per_group[gids]
        elif isinstance(m, (tuple, str)):
            vals = _col(eval_arg(m, table))[ids]
        else:
            vals = np.full(len(ids), m, np.float64 if isinstance(m, float) else np.int64)
        if name in table:
            old = _col(table[name])
            assert old.dtype == vals.dtype, "type conversion on update is not restated (the reference casts f64 into i64 columns)"
            new = old.copy()
        else:
            new = np.full(n, np.nan if vals.dtype == np.float64 else NULL_I64, vals.dtype)
        new[ids] = vals
        res[name] = new
    return res


def select(query: dict) -> dict:
    """Same contract as rayforce_amd.Engine.select, numpy in / numpy out."""
    table = query["from"]
    where_spec, by = query.get("where"), query.get("by")
    outs = [(k, v) for k, v in query.items() if k not in ("from", "where", "by", "take", "order")]
    ids = None
    if where_spec is not None:
        ids = where(mask_of(where_spec, table))
    if by is not None:
        if isinstance(by, dict):  # by: {name: col ...} -- several key columns (index_group_list, core/query.c:93-135)
            names, srcs = list(by.keys()), [key_column(c, table) for c in by.values()]
            if len(srcs) == 1:
                key = srcs[0]
                gids, firsts, groups, _ = group_index(key, ids)
            else:
                key = srcs[0]
                try:
                    comp, tmax, _, _ = composite_key(srcs, ids)
                    # the composite column is already restricted to the selected rows: no filter below this line
                    gids, firsts, groups, _ = group_index(comp, None, scope=(0, tmax))
                except NotPerfect:
                    if any(_col(c).dtype != np.int64 for c in srcs):
                        raise
                    gids, firsts, groups = group_rows(srcs, ids, query.get("order", "first"))
            pos = firsts if ids is None else ids[firsts]
            res = {nm: (_col(c)[pos] if groups else np.empty(0, np.int64)) for nm, c in zip(names, srcs)}
        else:
            key = table[by]
            gids, firsts, groups, _ = group_index(key, ids)
            pos = firsts if ids is None else ids[firsts]
            res = {by: _col(key)[pos] if groups else np.empty(0, np.int64)}
        for name, (fn, col) in outs:
            if isinstance(col, tuple):  # expression argument: scalar rules group by group
                if fn in ("count", "first"):
                    raise ValueError(f"({fn} expr) under by: is not restated")
                vals = eval_arg(col, table)
                res[name] = aggr_fold(fn, vals if ids is None else at_ids(vals, ids), gids, groups)
            elif fn == "first":
                c = _col(table[col])
                res[name] = c[pos] if groups else np.empty(0, c.dtype)
            else:
                c = table[col] if col is not None else key
                res[name] = aggr(fn, c, gids, ids, groups)
        return res
    if outs:
        res = {}
        for name, (fn, col) in outs:
            c = _col(eval_arg(col, table)) if col is not None else _col(next(iter(table.values())))
            if ids is not None:
                c = at_ids(c, ids)
            v = fold(fn, c)
            f64 = fn == "avg" or (fn != "count" and c.dtype == np.float64)
            if v is None:
                v = float("nan") if f64 else NULL_I64
            res[name] = np.array([v], np.float64 if f64 else np.int64)
        return res
    if ids is None:
        return dict(table)
    return {name: at_ids(col, ids) for name, col in table.items()}
