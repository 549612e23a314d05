"""Tuples -> the library's descriptors (rfx_pred_t, rfx_agg_t, rfx_xnode_t): what the Python host does instead of the reference's parser.
Nothing here decides how a query runs; it only spells the query in the C structures of include/rfx_hip.h."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import RfxError


def ctype_of(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return L.RFX_I64
    if t.dtype == torch.float64:
        return L.RFX_F64
    raise RfxError(f"unsupported column dtype {t.dtype} (the path handles i64 and f64 columns)")


class NotFused(Exception):
    """The where: tree has more comparisons / levels than one fused pass carries."""


def check_col(eng, t: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or t.dim() != 1 or not t.is_contiguous():
        raise RfxError("columns must be contiguous 1-D tensors")
    if t.device != eng.device:
        raise RfxError(f"column lives on {t.device}, engine on {eng.device}")
    if n is not None and t.numel() != n:
        raise RfxError("length mismatch")  # reference: err_length (core/cmp.c:633-640)
    return t

def resolve(eng, x, table):
    if isinstance(x, str):
        if table is None or x not in table:
            raise RfxError(f"unknown column {x!r}")
        return table[x]
    if isinstance(x, tuple) and x and x[0] in L.XOPS:  # an expression where a column is expected: evaluated once (k_derive), as the reference does
        col = eng.eval_expr(x, table)
        eng._keep.append(col)
        return col
    return x

def tree_leaves(eng, where) -> Tuple[int, List[tuple], List[int]]:
    """(logic, comparisons in order, rfx_pred_t.more per comparison): the where: tree's leaves with the depth of parentheses each sits
    in and the parentheses closing after it (rfx_hip.h: the two-level `more` form where it suffices, else RFX_PRED_TREE)."""
    if where is None:
        return L.RFX_AND, [], []
    head = where[0]
    if head in L.OPS:
        return L.RFX_AND, [tuple(where)], [0]
    if head not in ("and", "or"):
        raise RfxError(f"unknown predicate head {head!r}")
    leaves: List[tuple] = []
    dep: List[int] = []
    clo: List[int] = []

    def node(e, level_op, depth):
        h = e[0]
        if h in L.OPS:
            leaves.append(tuple(e))
            dep.append(depth)
            clo.append(0)
            return
        if h not in ("and", "or"):
            raise RfxError(f"unknown predicate head {h!r}")
        own = int(h != level_op)  # the opposite operator opens a parenthesis one level down; the same one is associative
        first = len(leaves)
        for sub in e[1:]:
            node(sub, h, depth + own)
        if own and len(leaves) > first:
            clo[-1] += 1

    node(where, head, 0)
    maxd = max(dep) if dep else 0
    if len(leaves) > L.RFX_MAX_PREDS or maxd > 3 or (maxd > 1 and len(leaves) < 3):
        raise NotFused()
    if maxd <= 1:
        more = [1 if (d == 1 and c == 0) else 0 for d, c in zip(dep, clo)]
    else:
        more = [L.RFX_PRED_TREE | d | (c << 4) for d, c in zip(dep, clo)]
    return (L.RFX_AND if head == "and" else L.RFX_OR), leaves, more

def preds(eng, leaves: Sequence[tuple], more: Sequence[int], table, n: Optional[int]):
    arr = (L.Pred * max(1, len(leaves)))()
    for i, (op, lhs, rhs) in enumerate(leaves):
        lhs = check_col(eng, resolve(eng, lhs, table), n)
        n = lhs.numel() if n is None else n
        p = arr[i]
        p.more = more[i]
        p.d_col, p.col_type, p.op = lhs.data_ptr(), ctype_of(lhs), L.OPS[op]
        rhs = resolve(eng, rhs, table) if isinstance(rhs, (str, tuple)) else rhs
        if isinstance(rhs, torch.Tensor):
            rhs = check_col(eng, rhs, n)
            p.d_rhs_col, p.rhs_type = rhs.data_ptr(), ctype_of(rhs)
            eng._keep.append(rhs)
        elif isinstance(rhs, bool):
            raise RfxError("boolean atoms are not comparable on this path")
        elif isinstance(rhs, int):
            p.d_rhs_col, p.rhs_type, p.rhs_i = None, L.RFX_I64, rhs
        elif isinstance(rhs, float):
            p.d_rhs_col, p.rhs_type, p.rhs_f = None, L.RFX_F64, rhs
        elif rhs is None:  # null atom compares as 0Nl
            p.d_rhs_col, p.rhs_type, p.rhs_i = None, L.RFX_I64, L.NULL_I64
        else:
            raise RfxError(f"unsupported rhs {type(rhs)}")
        eng._keep.append(lhs)
    return arr, n

def aggs(eng, aggs: Sequence[Tuple[str, object]], table, n: Optional[int]):
    if len(aggs) > L.RFX_EXEC_MAX_AGGS:
        raise RfxError(f"more than {L.RFX_EXEC_MAX_AGGS} output columns in one query")
    arr = (L.Agg * max(1, len(aggs)))()
    for i, (fn, col) in enumerate(aggs):
        a = arr[i]
        a.kind = L.AGGS[fn]
        if isinstance(col, tuple):
            n = agg_expr(eng, a, col, table, n)
            continue
        col = resolve(eng, col, table) if col is not None else None
        if col is None:
            if fn != "count":
                raise RfxError(f"{fn} needs a column")
            a.d_col, a.col_type = None, L.RFX_I64
        else:
            col = check_col(eng, col, n)
            n = col.numel() if n is None else n
            a.d_col, a.col_type = col.data_ptr(), ctype_of(col)
            eng._keep.append(col)
    return arr, n

def agg_expr(eng, a, expr, table, n):
    """``(op x y)`` with x / y columns, atoms or such expressions -> rfx_xnode_t list in evaluation order."""
    nodes = []

    def operand(x, o):
        nonlocal n
        if isinstance(x, tuple):
            o.kind, o.node = L.RFX_XK_NODE, build(x)
            return
        x = resolve(eng, x, table) if isinstance(x, (str, torch.Tensor)) else x
        if isinstance(x, torch.Tensor):
            col = check_col(eng, x, n)
            n = col.numel() if n is None else n
            o.kind, o.type, o.d_col = L.RFX_XK_COL, ctype_of(col), col.data_ptr()
            eng._keep.append(col)
        elif isinstance(x, float):
            o.kind, o.type, o.f = L.RFX_XK_ATOM, L.RFX_F64, x
        else:
            o.kind, o.type, o.i = L.RFX_XK_ATOM, L.RFX_I64, L.NULL_I64 if x is None else int(x)

    def build(e) -> int:
        if len(e) != 3 or e[0] not in L.XOPS:
            raise RfxError(f"unsupported expression {e!r}: (op lhs rhs) with op in + - * div / %")
        node = L.XNode()
        node.op = L.XOPS[e[0]]
        operand(e[1], node.l)
        operand(e[2], node.r)
        nodes.append(node)
        return len(nodes) - 1

    build(expr)
    if len(nodes) > L.RFX_MAX_XNODES:
        raise RfxError(f"expression too deep: at most {L.RFX_MAX_XNODES} operations")
    if not any(o.kind == L.RFX_XK_COL for nd in nodes for o in (nd.l, nd.r)):
        raise RfxError("an expression needs at least one column operand")
    arr = (L.XNode * len(nodes))(*nodes)
    eng._keep.append(arr)
    a.nxnodes, a.xnodes, a.d_col, a.col_type = len(nodes), arr, None, L.RFX_I64
    return n

