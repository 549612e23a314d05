"""Host-side mirror of the reference's operator surface for the select / where / by path, over HBM-resident columns.

Names follow RayforceDB (core/env.c:135-225): ``eq ne lt gt le ge`` (-> B8 byte masks), ``and_ / or_``, ``where``,
``at_ids`` (gather), ``sum min max avg count first`` and ``select`` with ``where:`` / ``by:`` clauses
(core/query.c:607-654).  Columns are 1-D ``torch.int64`` / ``torch.float64`` CUDA tensors; torch only owns the memory
and the stream -- all compute goes through librfx.so (hand-written HIP).  No CPU fallback exists.

Predicates are tuples ``(op, lhs, rhs)`` with op in ``== != < > <= >=``, lhs a column (tensor or table column name) and
rhs a Python int/float atom or another column; ``("and", p1, p2, ...)`` / ``("or", p1, ...)`` combine them flatly
(nested trees are evaluated through materialised masks, like the reference does).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib as L
from ._lib import RfxError

Column = torch.Tensor
PredSpec = tuple


def _ctype_of(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return L.RFX_I64
    if t.dtype == torch.float64:
        return L.RFX_F64
    raise RfxError(f"unsupported column dtype {t.dtype} (the path handles i64 and f64 columns)")


class Engine:
    """One GPU, one HIP stream, one librfx context."""

    def __init__(self, device: Union[int, torch.device, None] = None):
        self.lib = L.load_library()
        if not torch.cuda.is_available():
            raise RfxError("no GPU visible to torch: the MI355X engine has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        torch.cuda.set_device(self.device)
        self._ctx = C.c_void_p()
        # run on torch's current stream so that tensor ops and librfx kernels are ordered without extra syncs
        stream = torch.cuda.current_stream(self.device).cuda_stream or 1  # 0 = legacy default stream -> RFX_STREAM_LEGACY
        L.check(self.lib.rfx_hip_ctx_create(self.device.index, C.c_void_p(stream), C.byref(self._ctx)), "ctx_create")
        self._keep: List[torch.Tensor] = []

    def close(self) -> None:
        if self._ctx:
            self.lib.rfx_hip_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def sync(self) -> None:
        L.check(self.lib.rfx_hip_ctx_sync(self._ctx), "sync")

    def tune(self, blocks_per_cu: int = 0, flags: int = 0) -> None:
        L.check(self.lib.rfx_hip_ctx_tune(self._ctx, blocks_per_cu, flags), "tune")

    def stat(self, which):
        """Path counters of this context (include/rfx_hip.h RFX_STAT_*): 0 plane scatter launches, 1 plane fallbacks, 2 plane aggregate
        launches, 3 chunk scatter launches, 4 chunk aggregate launches."""
        return int(self.lib.rfx_hip_ctx_stat(self._ctx, which))

    def timer_start(self) -> None:
        L.check(self.lib.rfx_hip_timer_start(self._ctx), "timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        L.check(self.lib.rfx_hip_timer_stop(self._ctx, C.byref(ms)), "timer_stop")
        return float(ms.value)

    def profile(self, enable: bool = True) -> None:
        L.check(self.lib.rfx_hip_ctx_profile(self._ctx, int(enable)), "ctx_profile")

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        L.check(self.lib.rfx_hip_last_kernel_ms(self._ctx, C.byref(ms)), "last_kernel_ms")
        return float(ms.value)

    def empty(self, n: int, dtype=torch.int64) -> torch.Tensor:
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def column(self, host_array) -> torch.Tensor:
        """Upload a numpy int64/float64 array (host -> HBM over PCIe; not part of any timed region)."""
        t = torch.as_tensor(host_array)
        if t.dtype not in (torch.int64, torch.float64, torch.int8, torch.bool):
            raise RfxError(f"unsupported dtype {t.dtype}")
        if t.dtype == torch.bool:
            t = t.to(torch.int8)
        return t.contiguous().to(self.device)

    # ------------------------------------------------------------------ on-disk columns
    _FILE_DTYPES = {5: torch.int64, 6: torch.int64, 9: torch.int64, 10: torch.float64}  # i64, symbol ids, timestamp, f64

    def load_column(self, path: str) -> torch.Tensor:
        """A RayforceDB column file (core/binary.c:263-311) -> device column, moved with the pipelined pinned-staging path."""
        t, n = C.c_int32(), C.c_int64()
        L.check(self.lib.rfx_column_file_stat(path.encode(), C.byref(t), C.byref(n)), "column_file_stat")
        out = torch.empty(n.value, dtype=self._FILE_DTYPES[t.value], device=self.device)
        L.check(self.lib.rfx_hip_column_file_load(self._ctx, path.encode(), out.data_ptr(), n.value), "column_file_load")
        return out

    def load_splayed(self, directory: str, columns: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
        """A splayed table (core/io.c:1194-1364): `<dir>/.d` is the serialised symbol vector of the column names, every
        column is its own file.  Loads the 8-byte columns (all, or the ones asked for) into HBM."""
        import os
        names = self._splayed_names(directory)
        want = list(columns) if columns is not None else names
        missing = [c for c in want if c not in names]
        if missing:
            raise RfxError(f"no such column(s) in {directory}: {missing}")
        return {c: self.load_column(os.path.join(directory, c)) for c in want}

    def load_parted(self, root: str, table: str, columns: Optional[Sequence[str]] = None, where=None) -> Dict[str, torch.Tensor]:
        """A parted table, `(get-parted root 'table)` (core/vary.c:185-392): `<root>/<YYYY.MM.DD>/<table>/` is one splayed
        table per date (a `sym` entry beside them is skipped), partitions in ascending date order, all with the same columns;
        the result has the virtual `Date` column first (days since 2000.01.01, core/date.c:124-135, as i64 here) and every
        8-byte column concatenated over the partitions.

        `where` -- a comparison on `Date`, or and / or of such: ("==", "Date", "2024.01.02"), dates as text or day numbers --
        is the reference's partition pruning (cmp_map on the MAPCOMMON column, core/cmp.c:341-358): it is evaluated on the
        directory list, and a partition that fails it is never opened, let alone uploaded."""
        import datetime
        import os
        epoch = datetime.date(2000, 1, 1)

        def days(x) -> int:
            if isinstance(x, int):
                return x
            y, m, d = (int(p) for p in str(x).split("."))
            return (datetime.date(y, m, d) - epoch).days

        parts = []
        for name in os.listdir(root):
            if name == "sym":
                continue
            try:
                parts.append((days(name), name))
            except (ValueError, TypeError):
                raise RfxError(f"{root}: partition directory {name!r} is not a date (YYYY.MM.DD)") from None
        parts.sort()
        if not parts:
            raise RfxError(f"{root}: no partitions")

        def keep(p, d) -> bool:
            if p[0] in ("and", "or"):
                r = [keep(q, d) for q in p[1:]]
                return all(r) if p[0] == "and" else any(r)
            op, lhs, rhs = p
            if lhs != "Date":
                raise RfxError("load_parted prunes on the virtual Date column only; filter other columns in the query")
            c = days(rhs)
            return {"==": d == c, "!=": d != c, "<": d < c, ">": d > c, "<=": d <= c, ">=": d >= c}[op]

        kept = [(d, nm) for d, nm in parts if where is None or keep(where, d)]
        first_dir = os.path.join(root, (kept or parts)[0][1], table)  # schema: first partition that is read at all
        names = list(self._splayed_names(first_dir))
        want = list(columns) if columns is not None else names
        missing = [c for c in want if c not in names]
        if missing:
            raise RfxError(f"no such column(s) in {first_dir}: {missing}")
        # lengths and types from the headers only, then one device column per name and every file straight into its slice
        lens, types = [], {}
        for d, nm in kept:
            n_here = None
            for c in want:
                t, n = C.c_int32(), C.c_int64()
                L.check(self.lib.rfx_column_file_stat(os.path.join(root, nm, table, c).encode(), C.byref(t), C.byref(n)), "column_file_stat")
                if types.setdefault(c, t.value) != t.value:
                    raise RfxError(f"column {c} changes type between partitions")
                if n_here is not None and n.value != n_here:
                    raise RfxError(f"columns of partition {nm} differ in length")
                n_here = n.value
            lens.append(n_here or 0)
        total = sum(lens)
        out = {"Date": self.empty(total)}
        for c in want:
            out[c] = torch.empty(total, dtype=self._FILE_DTYPES[types[c]] if c in types else torch.int64, device=self.device)
        row = 0
        for (d, nm), n_here in zip(kept, lens):
            if n_here:
                out["Date"][row:row + n_here].fill_(d)  # plumbing: a constant per partition
                for c in want:
                    L.check(self.lib.rfx_hip_column_file_load(self._ctx, os.path.join(root, nm, table, c).encode(),
                                                               out[c].data_ptr() + row * 8, n_here), "column_file_load")
            row += n_here
        return out

    @staticmethod
    def _splayed_names(directory: str):
        import os
        raw = open(os.path.join(directory, ".d"), "rb").read()
        # serialised object: 16-byte IPC header (magic fa de fa ce, version, payload size), then type, attrs, len:i64, strings
        if len(raw) < 26 or raw[:4] != bytes.fromhex("fadeface") or raw[16] != 6:
            raise RfxError(f"{directory}/.d is not a serialised symbol vector")
        cnt = int.from_bytes(raw[18:26], "little")
        return [b.decode() for b in raw[26:].split(b"\0")[:cnt]]

    def upload(self, host_array) -> torch.Tensor:
        """Host numpy array (i64 / f64) -> device column through the pipelined path (what rfx_ops.c's residency cache uses)."""
        import numpy as np
        a = np.ascontiguousarray(host_array)
        if a.ndim != 1 or a.dtype not in (np.int64, np.float64):
            raise RfxError(f"upload: a 1-d int64 / float64 array is expected, not {a.dtype} with shape {a.shape}")
        out = torch.empty(a.shape[0], dtype=torch.float64 if a.dtype == np.float64 else torch.int64, device=self.device)
        L.check(self.lib.rfx_hip_h2d_pipelined(self._ctx, out.data_ptr(), a.ctypes.data, a.nbytes), "h2d_pipelined")
        return out

    def gen_i64(self, n: int, seed: int, modulus: int, row0: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = self.empty(n, torch.int64) if out is None else out
        L.check(self.lib.rfx_hip_gen_i64(self._ctx, out.data_ptr(), n, seed, row0, modulus), "gen_i64")
        return out

    def gen_f64(self, n: int, seed: int, row0: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = self.empty(n, torch.float64) if out is None else out
        L.check(self.lib.rfx_hip_gen_f64(self._ctx, out.data_ptr(), n, seed, row0), "gen_f64")
        return out

    # ------------------------------------------------------------------ descriptors
    def _check_col(self, t: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
        if not isinstance(t, torch.Tensor) or t.dim() != 1 or not t.is_contiguous():
            raise RfxError("columns must be contiguous 1-D tensors")
        if t.device != self.device:
            raise RfxError(f"column lives on {t.device}, engine on {self.device}")
        if n is not None and t.numel() != n:
            raise RfxError("length mismatch")  # reference: err_length (core/cmp.c:633-640)
        return t

    def _flatten(self, where, table) -> Tuple[int, List[tuple]]:
        """Return (logic, [simple predicates]) or raise NotFlat for trees deeper than two levels.  A two-level tree --
        ``(and (or A B) C)``, ``(or A (and B C))`` -- stays ONE fused pass: the comparisons of a parenthesis of the opposite operator
        are marked ``_More`` (rfx_pred_t::more) on all but the last; the same operator nested in itself is associative and flattened."""
        if where is None:
            return L.RFX_AND, []
        head = where[0]
        if head in L.OPS:
            return L.RFX_AND, [where]
        if head not in ("and", "or"):
            raise RfxError(f"unknown predicate head {head!r}")
        out: List[tuple] = []

        def arm(e):
            h = e[0]
            if h in L.OPS:
                out.append(tuple(e))
            elif h == head:
                for s in e[1:]:
                    arm(s)
            elif h in ("and", "or") and len(e) > 1 and all(s[0] in L.OPS for s in e[1:]):
                out.extend(_More(s) for s in e[1:-1])
                out.append(tuple(e[-1]))
            else:
                raise _NotFlat()

        for s in where[1:]:
            arm(s)
        if len(out) > L.RFX_MAX_PREDS:
            raise _NotFlat()
        return (L.RFX_AND if head == "and" else L.RFX_OR), out

    def _resolve(self, x, table):
        if isinstance(x, str):
            if table is None or x not in table:
                raise RfxError(f"unknown column {x!r}")
            return table[x]
        if isinstance(x, tuple) and x and x[0] in L.XOPS:  # an expression where a column is expected: evaluate it once (k_derive)
            col = self.eval_expr(x, table)
            self._keep.append(col)
            return col
        return x

    def eval_expr(self, expr, table=None) -> torch.Tensor:
        """``(op x y)`` / an expression tree over columns and atoms as a device column: ray_add / ray_sub / ray_mul / ray_div
        (binop_map, core/math.c:2280-2345) in ONE pass whatever the depth.  Also how `where:` takes predicates over expressions."""
        a = L.Agg()
        n = self._agg_expr(a, "sum", expr, table, None)
        if n is None:
            raise RfxError("an expression needs at least one column operand")
        out = torch.empty(n, dtype=torch.float64 if L.agg_input_type(a) == L.RFX_F64 else torch.int64, device=self.device)
        t = C.c_int32()
        L.check(self.lib.rfx_hip_eval_expr(self._ctx, C.byref(a), n, out.data_ptr(), C.byref(t)), "eval_expr")
        return out

    def _preds(self, preds: Sequence[tuple], table, n: Optional[int]):
        if len(preds) > L.RFX_MAX_PREDS:
            raise RfxError("too many predicates for one fused pass")
        arr = (L.Pred * max(1, len(preds)))()
        for i, pr in enumerate(preds):
            op, lhs, rhs = pr
            lhs = self._check_col(self._resolve(lhs, table), n)
            n = lhs.numel() if n is None else n
            p = arr[i]
            p.more = 1 if isinstance(pr, _More) else 0
            p.d_col = lhs.data_ptr()
            p.col_type = _ctype_of(lhs)
            p.op = L.OPS[op]
            rhs = self._resolve(rhs, table) if isinstance(rhs, (str, tuple)) else rhs
            if isinstance(rhs, torch.Tensor):
                rhs = self._check_col(rhs, n)
                p.d_rhs_col = rhs.data_ptr()
                p.rhs_type = _ctype_of(rhs)
                self._keep.append(rhs)
            elif isinstance(rhs, bool):
                raise RfxError("boolean atoms are not comparable on this path")
            elif isinstance(rhs, int):
                p.d_rhs_col = None
                p.rhs_type = L.RFX_I64
                p.rhs_i = rhs
            elif isinstance(rhs, float):
                p.d_rhs_col = None
                p.rhs_type = L.RFX_F64
                p.rhs_f = rhs
            elif rhs is None:  # null atom compares as 0Nl
                p.d_rhs_col = None
                p.rhs_type = L.RFX_I64
                p.rhs_i = L.NULL_I64
            else:
                raise RfxError(f"unsupported rhs {type(rhs)}")
            self._keep.append(lhs)
        return arr, n

    def _aggs(self, aggs: Sequence[Tuple[str, Optional[torch.Tensor]]], table, n: Optional[int]):
        if len(aggs) > L.RFX_MAX_AGGS:
            raise RfxError("too many aggregates for one fused pass")
        arr = (L.Agg * max(1, len(aggs)))()
        for i, (fn, col) in enumerate(aggs):
            a = arr[i]
            a.kind = L.AGGS[fn]
            if isinstance(col, tuple):
                n = self._agg_expr(a, fn, col, table, n)
                continue
            col = self._resolve(col, table) if col is not None else None
            if col is None:
                if fn != "count":
                    raise RfxError(f"{fn} needs a column")
                a.d_col = None
                a.col_type = L.RFX_I64
            else:
                col = self._check_col(col, n)
                n = col.numel() if n is None else n
                a.d_col = col.data_ptr()
                a.col_type = _ctype_of(col)
                self._keep.append(col)
        return arr, n

    def _agg_expr(self, a, fn: str, expr, table, n):
        """``(fn (op lhs rhs))``: lhs / rhs are columns or atoms, at least one a column."""
        if len(expr) != 3 or expr[0] not in L.XOPS:
            raise RfxError(f"unsupported expression {expr!r}: (op lhs rhs) with op in + - * div")
        if isinstance(expr[1], tuple) or isinstance(expr[2], tuple):
            return self._agg_expr_tree(a, expr, table, n)
        op, lhs, rhs = expr
        l = self._resolve(lhs, table) if isinstance(lhs, (str, torch.Tensor)) else lhs
        r = self._resolve(rhs, table) if isinstance(rhs, (str, torch.Tensor)) else rhs
        lcol, rcol = isinstance(l, torch.Tensor), isinstance(r, torch.Tensor)
        if not (lcol or rcol):
            raise RfxError("an expression needs at least one column operand")
        a.xop = L.XOPS[op]
        col, other, swap = (l, r, False) if lcol else (r, l, True)
        col = self._check_col(col, n)
        n = col.numel() if n is None else n
        a.d_col, a.col_type = col.data_ptr(), _ctype_of(col)
        a.xflags = L.RFX_XF_SWAP if swap else 0
        self._keep.append(col)
        if isinstance(other, torch.Tensor):
            other = self._check_col(other, n)
            a.d_xrhs_col, a.xrhs_type = other.data_ptr(), _ctype_of(other)
            self._keep.append(other)
        elif isinstance(other, float):
            a.d_xrhs_col, a.xrhs_type, a.xrhs_f = None, L.RFX_F64, other
        else:
            a.d_xrhs_col, a.xrhs_type, a.xrhs_i = None, L.RFX_I64, L.NULL_I64 if other is None else int(other)
        return n

    @staticmethod
    def _agg_chunks(aggs):
        """Split an output list into launches: <= RFX_MAX_AGGS aggregates, <= RFX_MAX_EXPRS expressions and a handful of
        distinct argument columns each (predicate and key columns need plan slots too)."""
        chunks, cur, nx, cols = [], [], 0, set()
        for fn, col in aggs:
            def leaves(e):
                return [y for x in e[1:] for y in (leaves(x) if isinstance(x, tuple) else [x])]
            ops = [x for x in leaves(col) if isinstance(x, (str, torch.Tensor))] if isinstance(col, tuple) else ([col] if col is not None else [])
            ids = {x if isinstance(x, str) else x.data_ptr() for x in ops}
            x = 1 if isinstance(col, tuple) else 0
            if cur and (len(cur) >= L.RFX_MAX_AGGS or nx + x > L.RFX_MAX_EXPRS or len(cols | ids) > 4):
                chunks.append(cur)
                cur, nx, cols = [], 0, set()
            cur.append((fn, col))
            nx += x
            cols |= ids
        if cur:
            chunks.append(cur)
        return chunks or [[]]

    def _agg_expr_tree(self, a, expr, table, n):
        """Nested expression -> rfx_xnode_t list in evaluation order (operands: column / atom / earlier node)."""
        nodes = []

        def operand(x, o):
            nonlocal n
            if isinstance(x, tuple):
                o.kind, o.node = L.RFX_XK_NODE, build(x)
                return
            x = self._resolve(x, table) if isinstance(x, (str, torch.Tensor)) else x
            if isinstance(x, torch.Tensor):
                col = self._check_col(x, n)
                n = col.numel() if n is None else n
                o.kind, o.type, o.d_col = L.RFX_XK_COL, _ctype_of(col), col.data_ptr()
                self._keep.append(col)
            elif isinstance(x, float):
                o.kind, o.type, o.f = L.RFX_XK_ATOM, L.RFX_F64, x
            else:
                o.kind, o.type, o.i = L.RFX_XK_ATOM, L.RFX_I64, L.NULL_I64 if x is None else int(x)

        def build(e) -> int:
            if len(e) != 3 or e[0] not in L.XOPS:
                raise RfxError(f"unsupported expression {e!r}: (op lhs rhs) with op in + - * div")
            node = L.XNode()
            node.op = L.XOPS[e[0]]
            operand(e[1], node.l)
            operand(e[2], node.r)
            nodes.append(node)
            return len(nodes) - 1

        build(expr)
        if len(nodes) > L.RFX_MAX_XNODES:
            raise RfxError(f"expression too deep: at most {L.RFX_MAX_XNODES} operations")
        arr = (L.XNode * len(nodes))(*nodes)
        self._keep.append(arr)
        a.nxnodes, a.xnodes = len(nodes), arr
        a.d_col, a.col_type = None, L.RFX_I64
        return n

    def _arg_f64(self, col, table) -> bool:
        """Element type of an aggregate's argument: a column, or (op lhs rhs) with the reference's promotion."""
        if isinstance(col, tuple):
            def f(x):
                if isinstance(x, tuple):
                    return self._arg_f64(x, table)
                x = self._resolve(x, table) if isinstance(x, (str, torch.Tensor)) else x
                return x.dtype == torch.float64 if isinstance(x, torch.Tensor) else isinstance(x, float)
            if col[0] == "/":  # ray_div keeps the left operand's type (infer_div_type, core/math.c:149-188)
                return f(col[1])
            return col[0] == "div" or f(col[1]) or f(col[2])
        return self._resolve(col, table).dtype == torch.float64

    @staticmethod
    def _value(v: L.Value):
        if v.type == L.RFX_F64:
            return float("nan") if v.is_null else float(v.f)
        return None if v.is_null else int(v.i)

    # ------------------------------------------------------------------ K1/K5: fused filter -> aggregates
    def filter_aggr_partials(self, aggs, where=None, table=None, nrows: Optional[int] = None, row0: int = 0,
                             out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Device-resident partials ((nagg+1) x 64 bytes as a uint8 tensor) -- the multi-GPU exchange payload."""
        self._keep.clear()
        logic, flat = self._flatten(where, table)
        parr, n = self._preds(flat, table, nrows)
        aarr, n = self._aggs(aggs, table, n)
        if n is None:
            raise RfxError("cannot infer the row count (count without column and without predicate needs nrows=)")
        out = torch.empty((len(aggs) + 1) * 64, dtype=torch.uint8, device=self.device) if out is None else out
        L.check(self.lib.rfx_hip_filter_aggr(self._ctx, parr, len(flat), logic, aarr, len(aggs), n, row0, out.data_ptr()),
                "filter_aggr")
        return out

    def filter_aggr(self, aggs, where=None, table=None, nrows: Optional[int] = None):
        """``select {aggs} from t where p`` without ``by:`` -> ([values], selected_rows).  (syncs)"""
        try:
            logic, flat = self._flatten(where, table)
        except _NotFlat:
            return self._filter_aggr_via_ids(aggs, where, table)
        chunks = self._agg_chunks(aggs)
        if len(chunks) > 1:  # more outputs than one fused pass carries: several passes, same selection
            vals, sel = [], 0
            for ch in chunks:
                v, sel = self.filter_aggr(ch, where, table, nrows)
                vals += v
            return vals, sel
        self._keep.clear()
        parr, n = self._preds(flat, table, nrows)
        aarr, n = self._aggs(aggs, table, n)
        if n is None:
            raise RfxError("cannot infer the row count")
        vals = (L.Value * max(1, len(aggs)))()
        sel = C.c_int64()
        L.check(self.lib.rfx_hip_filter_aggr_host(self._ctx, parr, len(flat), logic, aarr, len(aggs), n, vals, C.byref(sel)),
                "filter_aggr")
        return [self._value(vals[i]) for i in range(len(aggs))], int(sel.value)

    def filter_aggr_dist(self, aggs, where=None, table=None, nrows: Optional[int] = None, row0: int = 0):
        """The same over the row-range SHARDED table, through the C exchange (rfx_dist_filter_aggr_host: local fused pass, one
        ncclAllGather of the partials, rank-ordered fold).  Flat predicates and at most RFX_MAX_AGGS aggregates.  (syncs)"""
        logic, flat = self._flatten(where, table)
        self._keep.clear()
        parr, n = self._preds(flat, table, nrows)
        aarr, n = self._aggs(aggs, table, n)
        if n is None:
            raise RfxError("cannot infer the row count")
        vals = (L.Value * max(1, len(aggs)))()
        sel = C.c_int64()
        L.check(self.lib.rfx_dist_filter_aggr_host(self._ctx, parr, len(flat), logic, aarr, len(aggs), n, row0, vals, C.byref(sel)), "dist_filter_aggr")
        return [self._value(vals[i]) for i in range(len(aggs))], int(sel.value)

    def _filter_aggr_via_ids(self, aggs, where, table):
        # nested boolean tree: masks -> where -> gather -> plain folds (the reference's own plan, on the GPU)
        ids = self.where(where, table)
        gathered = []
        def pick(x):
            x = self._resolve(x, table) if isinstance(x, (str, torch.Tensor)) else x
            return self.at_ids(x, ids) if isinstance(x, torch.Tensor) else x

        for fn, col in aggs:
            if isinstance(col, tuple):
                def pick_tree(e):
                    return (e[0],) + tuple(pick_tree(x) if isinstance(x, tuple) else pick(x) for x in e[1:])
                gathered.append((fn, pick_tree(col)))
                continue
            col = self._resolve(col, table) if col is not None else None
            gathered.append((fn, self.at_ids(col, ids) if col is not None else None))
        vals, _ = self.filter_aggr(gathered, None, None, nrows=int(ids.numel()))
        return vals, int(ids.numel())

    # scalar verbs of the reference (core/math.c:2388-2526, core/misc.c:43-60)
    def sum(self, col, where=None, table=None):
        return self.filter_aggr([("sum", col)], where, table)[0][0]

    def min(self, col, where=None, table=None):
        return self.filter_aggr([("min", col)], where, table)[0][0]

    def max(self, col, where=None, table=None):
        return self.filter_aggr([("max", col)], where, table)[0][0]

    def avg(self, col, where=None, table=None):
        return self.filter_aggr([("avg", col)], where, table)[0][0]

    def count(self, col, where=None, table=None):
        return self.filter_aggr([("count", col)], where, table)[0][0]

    def first(self, col, where=None, table=None):
        return self.filter_aggr([("first", col)], where, table)[0][0]

    # ------------------------------------------------------------------ K2: masks
    def cmp(self, op: str, lhs, rhs, table=None) -> torch.Tensor:
        """ray_eq .. ray_ge on a column: B8 byte mask (int8 tensor of 0/1)."""
        self._keep.clear()
        parr, n = self._preds([(op, lhs, rhs)], table, None)
        out = torch.empty(n, dtype=torch.int8, device=self.device)
        L.check(self.lib.rfx_hip_cmp_mask(self._ctx, parr, n, out.data_ptr()), "cmp_mask")
        return out

    def eq(self, a, b): return self.cmp("==", a, b)
    def ne(self, a, b): return self.cmp("!=", a, b)
    def lt(self, a, b): return self.cmp("<", a, b)
    def gt(self, a, b): return self.cmp(">", a, b)
    def le(self, a, b): return self.cmp("<=", a, b)
    def ge(self, a, b): return self.cmp(">=", a, b)

    def _logic(self, logic: int, masks) -> torch.Tensor:
        if not masks:
            raise RfxError("and/or need at least one argument")
        acc = masks[0].clone()
        for m in masks[1:]:
            if isinstance(m, (bool, int)):
                L.check(self.lib.rfx_hip_mask_logic(self._ctx, logic, acc.data_ptr(), None, int(bool(m)), acc.numel()), "mask_logic")
                continue
            if m.numel() != acc.numel():
                raise RfxError("length mismatch")  # reference: err_type on unequal lengths (core/logic.c:122-126)
            L.check(self.lib.rfx_hip_mask_logic(self._ctx, logic, acc.data_ptr(), m.data_ptr(), 0, acc.numel()), "mask_logic")
        return acc

    def and_(self, *masks) -> torch.Tensor:
        return self._logic(L.RFX_AND, masks)

    def or_(self, *masks) -> torch.Tensor:
        return self._logic(L.RFX_OR, masks)

    def mask_of(self, where, table=None) -> torch.Tensor:
        """Materialise any predicate tree as a byte mask (the reference's evaluation order, on the GPU)."""
        head = where[0]
        if head in L.OPS:
            return self.cmp(head, where[1], where[2], table)
        subs = [self.mask_of(w, table) for w in where[1:]]
        return self.and_(*subs) if head == "and" else self.or_(*subs)

    # ------------------------------------------------------------------ K3: where
    def where(self, where, table=None, row0: int = 0) -> torch.Tensor:
        """Ascending row ids of the selected rows.  `where` = int8/bool mask tensor or a predicate tree.  (syncs)"""
        cnt = C.c_int64()
        if isinstance(where, torch.Tensor):
            mask = self._check_col(where.view(torch.int8) if where.dtype == torch.bool else where)
            if mask.dtype != torch.int8:
                raise RfxError("where expects a B8 mask")  # reference: err_type (core/items.c:1395)
            L.check(self.lib.rfx_hip_where_begin(self._ctx, None, 0, L.RFX_AND, mask.data_ptr(), mask.numel(), C.byref(cnt)), "where_begin")
        else:
            try:
                logic, flat = self._flatten(where, table)
            except _NotFlat:
                return self.where(self.mask_of(where, table), row0=row0)
            self._keep.clear()
            parr, n = self._preds(flat, table, None)
            # one pass (rfx_where_once.hip): the buffer is sized by a sampled estimate, the count comes back exact; a selection the
            # sample underestimated (clustered rows) says so and runs again with the exact size
            est = C.c_int64()
            L.check(self.lib.rfx_hip_where_estimate(self._ctx, parr, len(flat), logic, n, C.byref(est)), "where_estimate")
            out = torch.empty(int(est.value), dtype=torch.int64, device=self.device)
            rc = self.lib.rfx_hip_where_once(self._ctx, parr, len(flat), logic, n, row0, out.data_ptr(), out.numel(), C.byref(cnt))
            if rc == L.RFX_ELIMIT and int(cnt.value) > out.numel():
                out = torch.empty(int(cnt.value), dtype=torch.int64, device=self.device)
                rc = self.lib.rfx_hip_where_once(self._ctx, parr, len(flat), logic, n, row0, out.data_ptr(), out.numel(), C.byref(cnt))
            L.check(rc, "where_once")
            return out[:int(cnt.value)]
        out = torch.empty(int(cnt.value), dtype=torch.int64, device=self.device)
        L.check(self.lib.rfx_hip_where_emit(self._ctx, row0, out.data_ptr()), "where_emit")
        return out

    # ------------------------------------------------------------------ K4: gather
    def at_ids(self, col: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        self._check_col(col)
        self._check_col(ids)
        if ids.dtype != torch.int64:
            raise RfxError("ids must be i64")
        _ctype_of(col)
        out = torch.empty(ids.numel(), dtype=col.dtype, device=self.device)
        L.check(self.lib.rfx_hip_gather(self._ctx, col.data_ptr(), ids.data_ptr(), ids.numel(), out.data_ptr()), "gather")
        return out

    # ------------------------------------------------------------------ K6: scope
    def scope(self, key: torch.Tensor, where=None, table=None, aggs=None) -> Tuple[int, int, int]:
        """index_scope_i64: (min, max, rows_seen).  (syncs)  With `aggs` (the group-by's aggregates) the pass may also leave the
        rows radix-partitioned for the group_dense_accumulate call that follows (rfx_hip_group_scope)."""
        self._check_col(key)
        if key.dtype != torch.int64:
            raise RfxError("group key must be i64 on this path")
        logic, flat = self._flatten(where, table)
        self._keep.clear()
        parr, n = self._preds(flat, table, key.numel())
        mn, mx, cnt = C.c_int64(), C.c_int64(), C.c_int64()
        if aggs:
            aarr, _ = self._aggs(aggs, table, key.numel())
            L.check(self.lib.rfx_hip_group_scope(self._ctx, key.data_ptr(), parr, len(flat), logic, aarr, len(aggs), key.numel(),
                                                 C.byref(mn), C.byref(mx), C.byref(cnt)), "group_scope")
        else:
            L.check(self.lib.rfx_hip_scope_i64(self._ctx, key.data_ptr(), parr, len(flat), logic, key.numel(), C.byref(mn), C.byref(mx),
                                               C.byref(cnt)), "scope_i64")
        return int(mn.value), int(mx.value), int(cnt.value)

    # ------------------------------------------------------------------ K7/K8/K10 dense group-by, K9 hashed
    def group_tables(self, aggs_arr, nagg: int, kmin: int, rng: int, hashed: bool = False, store: Optional[torch.Tensor] = None):
        """Allocate (or wrap `store`) one table set.  Returns (struct, backing tensor [n_arrays, cells], layout)."""
        n_arr = C.c_int()
        L.check(self.lib.rfx_hip_group_table_arrays(aggs_arr, nagg, C.byref(n_arr)), "group_table_arrays")
        cells = rng + 1 if hashed else rng
        total = n_arr.value + (1 if hashed else 0)
        if store is None:
            store = torch.empty((total, cells), dtype=torch.int64, device=self.device)
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

    spec_retries = 0  # sampled scopes that a pass reported as too small (the query then ran again under the exact scope)

    def group_by(self, key, aggs, where=None, table=None, total_rows: Optional[int] = None, row0: int = 0, _collective=None,
                 order: str = "first", _cap_hint: int = 0, _probe_first: Optional[torch.Tensor] = None, _spec: bool = True):
        """``select {aggs} from t [where p] by key`` -> dict(keys=, first=, results=[...], groups=n) of device tensors.

        Group order is first occurrence (`order="radix"`: for key tuples that take the row-hash path, the order of the
        reference's multi-threaded radix grouping instead -- hash & 1023, then first occurrence).  `_collective(store, layout, aggs_arr, hashed)` -- if given -- is called
        between the local scatter pass and the ranking step so that several GPUs can merge their tables.  (syncs)
        """
        if where is not None and not isinstance(where, torch.Tensor):
            try:
                self._flatten(where, table)
            except _NotFlat:
                # a nested boolean tree: the reference's own plan -- masks, where, gather (filter_collect), then group the gathered
                # columns (core/query.c:607-654) -- on the device; `first` is translated back through the ids
                if _collective is not None:
                    raise RfxError("nested boolean trees are not fused with `by:` across GPUs; pass ids via where() + at_ids()")
                names = set()

                def leaves(e):
                    for x in e[1:]:
                        if isinstance(x, tuple):
                            leaves(x)
                        elif isinstance(x, str):
                            names.add(x)
                        elif isinstance(x, torch.Tensor):
                            raise RfxError("nested boolean trees with `by:` need columns given by name")

                for k in (key if isinstance(key, list) else [key]):
                    leaves(("k", k[1]) if isinstance(k, tuple) else ("k", k))
                for _, col in aggs:
                    if col is not None:
                        leaves(col if isinstance(col, tuple) else ("a", col))
                ids = self.where(where, table)
                sub = {nm: self.at_ids(self._check_col(self._resolve(nm, table)), ids) for nm in names}
                r = self.group_by(key, aggs, None, sub, None, 0, None, order)
                if r["groups"]:
                    r["first"] = self.at_ids(ids, r["first"])
                return r
        chunks = self._agg_chunks(aggs)
        if len(chunks) > 1:  # more outputs than one table set carries: several passes (same groups, same order)
            r = None
            for ch in chunks:  # (a later launch starts from the hashed-table capacity the first one ended with: same groups)
                part = self.group_by(key, ch, where, table, total_rows, row0, _collective, order, r.get("cap", 0) if r else 0,
                                     _probe_first if r is None else None, _spec)
                if r is None:
                    r = part
                else:
                    r["results"] += part["results"]
            return r
        multi = None
        is_xbar = lambda k: isinstance(k, tuple) and len(k) == 3 and k[0] == "xbar"
        if isinstance(key, (list, tuple)) and not is_xbar(key) and len(key) == 1:
            key = key[0]
        if isinstance(key, (list, tuple)) and not is_xbar(key):
            # several key columns -> one composite dense key (index_group_list_perfect, core/index.c:2308-2424)
            kcols = [self._key_col(k, table) for k in key]
            try:
                spec_keys = _spec and self._may_speculate(kcols[0].numel(), _collective, kcols[0])
                tmax, seen, multi = self._composite_plan(kcols, where, table, _collective, sampled=spec_keys)
                if spec_keys and multi is None:
                    spec_keys = False
                    tmax, seen, multi = self._composite_plan(kcols, where, table, _collective)
            except _NotPerfect as e:  # ranges overflow 64 bits / null keys: the reference's row-hash path (core/index.c:2731-2790)
                return self._group_by_row_hash(kcols, e.scopes, aggs, where, table, total_rows, row0, _collective, order)
            key = kcols[0]
        else:
            key = self._key_col(key, table)
        if key.dtype != torch.int64:
            raise RfxError("group key must be i64 on this path (f64 keys group on their bit pattern: view as int64)")
        n = key.numel()
        try:
            logic, flat = self._flatten(where, table)
        except _NotFlat:
            raise RfxError("nested boolean trees are not fused with `by:`; pass ids via where() + at_ids()")
        spec = False  # the scope below was SAMPLED: the pass reports keys outside it, a report sends the query through the exact scope
        if multi is None:
            if _spec and self._may_speculate(n, _collective, key):
                smn, smx = self.scope_sample(key)
                if smn != L.NULL_I64 and 0 < smx - smn + 1 <= self.SPEC_MAX_SLOTS:
                    kmin, kmax, seen, spec = smn, smx, n, True
            if not spec:
                kmin, kmax, seen = self.scope(key, where, table, aggs)
                if _collective is not None:
                    kmin, kmax, seen = _collective("scope", (kmin, kmax, seen, self.device))
        else:
            kmin, kmax = 0, tmax  # forced scope, core/index.c:2421
            spec = bool(locals().get("spec_keys", False))
        self._keep.clear()
        parr, _ = self._preds(flat, table, n)
        aarr, _ = self._aggs(aggs, table, n)
        nagg = len(aggs)
        total_rows = n if total_rows is None else total_rows
        out_dtypes = []
        for i, (fn, col) in enumerate(aggs):
            if fn in ("avg",):
                out_dtypes.append(torch.float64)
            elif fn == "count":
                out_dtypes.append(torch.int64)
            else:
                out_dtypes.append(torch.float64 if L.agg_input_type(aarr[i]) == L.RFX_F64 else torch.int64)
        if seen == 0:
            r = dict(groups=0, keys=self.empty(0), first=self.empty(0), results=[self.empty(0, d) for d in out_dtypes])
            if multi is not None:
                r["key_columns"] = [self.empty(0) for _ in multi[0]]
            return r
        rng = kmax - kmin + 1
        # index_group_i64_scoped: dense "perfect hash" iff range <= rows (core/index.c:2013); else open addressing
        dense = 0 < rng <= max(seen, 1) and kmin != L.NULL_I64
        ng = C.c_int64()
        if spec and not dense:
            return self.group_by(key if multi is None else kcols, aggs, where, table, total_rows, row0, _collective, order, _cap_hint, _probe_first, False)
        if dense:
            t, store, layout = self.group_tables(aarr, nagg, kmin, rng)
            L.check(self.lib.rfx_hip_group_tables_init(self._ctx, aarr, C.byref(t)), "group_tables_init")
            if spec:
                L.check(self.lib.rfx_hip_ctx_speculative(self._ctx, 1), "ctx_speculative")
            try:
                if multi is None:
                    rc = self.lib.rfx_hip_group_dense_accumulate(self._ctx, key.data_ptr(), parr, len(flat), logic, aarr, n, row0, C.byref(t))
                else:
                    k = len(kcols)
                    ptrs = (C.c_void_p * k)(*[kc.data_ptr() for kc in kcols])
                    rc = self.lib.rfx_hip_group_dense_accumulate_keys(self._ctx, ptrs, (C.c_int64 * k)(*multi[0]), (C.c_int64 * k)(*multi[1]), k, parr,
                                                                      len(flat), logic, aarr, n, row0, C.byref(t))
            finally:
                if spec:
                    self.lib.rfx_hip_ctx_speculative(self._ctx, 0)  # (the report stays readable until the next speculative(1))
            redo = False
            if spec:
                if rc == L.RFX_ESTATE:  # the pass would have taken a path that cannot report out-of-scope keys: nothing ran
                    redo = True
                else:
                    L.check(rc, "group_dense_accumulate")
                    bad = C.c_int(0)
                    L.check(self.lib.rfx_hip_group_out_of_scope(self._ctx, C.byref(bad)), "group_out_of_scope")
                    redo = bool(bad.value)
            else:
                L.check(rc, "group_dense_accumulate")
            if redo:  # an outlier, a null key, a range the sample missed: the exact scope and the pass again
                del t, store
                self.spec_retries += 1
                kk = key if multi is None else kcols[0]
                self.__dict__.setdefault("_spec_failed", set()).add((kk.data_ptr(), kk.numel()))
                return self.group_by(key if multi is None else kcols, aggs, where, table, total_rows, row0, _collective, order, _cap_hint, _probe_first, False)
            if _collective is not None:  # the kinds of THIS launch's aggregates (a chunk of the query's, or the row-hash path's extras)
                _collective("tables", (store, layout, [int(aarr[i].kind) for i in range(nagg)],
                                       [L.agg_input_type(aarr[i]) == L.RFX_F64 for i in range(nagg)], t, aarr))
            if rng <= L.RFX_RANK_SMALL and _collective is None:
                # few slots: rank + emit in ONE launch, results sliced out of its block after one host round trip (the group count)
                block = self.empty(1 + (2 + nagg) * rng)
                L.check(self.lib.rfx_hip_group_rank_emit_small(self._ctx, aarr, C.byref(t), row0, 0, block.data_ptr()), "group_rank_emit_small")
                g = int(block[0])
                r = dict(groups=g, keys=block[1:1 + g], first=block[1 + rng:1 + rng + g], dense=True, cap=0,
                         results=[block[1 + (2 + a) * rng:1 + (2 + a) * rng + g].view(out_dtypes[a]) for a in range(nagg)])
                if multi is not None:
                    mins, mults, ranges = multi
                    r["key_columns"] = []
                    for mn, mu, rg in zip(mins, mults, ranges):
                        kc = self.empty(g)
                        L.check(self.lib.rfx_hip_composite_decode(self._ctx, r["keys"].data_ptr(), g, mn, mu, rg, kc.data_ptr()), "composite_decode")
                        r["key_columns"].append(kc)
                return r
            L.check(self.lib.rfx_hip_group_rank(self._ctx, C.byref(t), total_rows, C.byref(ng)), "group_rank")
        else:
            if multi is not None:  # sparse composite: the hashed path keys on the materialised column (core/index.c:2421 -> :2092)
                k = len(kcols)
                key = self.empty(n)
                ptrs = (C.c_void_p * k)(*[kc.data_ptr() for kc in kcols])
                L.check(self.lib.rfx_hip_composite_key(self._ctx, ptrs, (C.c_int64 * k)(*multi[0]), (C.c_int64 * k)(*multi[1]), k, n, key.data_ptr()),
                        "composite_key")
            # capacity: the reference sizes its table by the row count (ht_oa_create(len), core/index.c:1805); the distinct keys
            # are usually far fewer, so start at 4 M slots (>= 2 M distinct keys) and grow x16 whenever the table reports full
            cap_max = 1 << max(4, math.ceil(math.log2(max(2 * seen, 16))))
            cap = min(cap_max, max(1 << 22, _cap_hint))
            while True:
                t, store, layout = self.group_tables(aarr, nagg, 0, cap, hashed=True)
                L.check(self.lib.rfx_hip_hash_tables_init(self._ctx, aarr, C.byref(t)), "hash_tables_init")
                rc = self.lib.rfx_hip_group_hash_accumulate(self._ctx, key.data_ptr(), parr, len(flat), logic, aarr, n, row0, C.byref(t))
                full = rc == L.RFX_ELIMIT
                if _collective is not None:
                    full = bool(_collective("flag", int(full)))  # every rank grows together: merged tables share one capacity
                if not full:
                    L.check(rc, "group_hash_accumulate")
                    break
                if cap >= cap_max:
                    L.check(rc, "group_hash_accumulate")
                del t, store
                cap = cap_max  # the launch gave up at 3/4 load, early: far more distinct keys than the first guess, take the reference's size
            if _collective is not None:
                def make_tables(other_store):
                    return self.group_tables(aarr, nagg, 0, cap, hashed=True, store=other_store)[0]

                def merge(other):
                    L.check(self.lib.rfx_hip_hash_tables_merge(self._ctx, aarr, C.byref(t), C.byref(other)), "hash_tables_merge")

                _collective("hash_tables", (self, make_tables, store, merge))
            if _probe_first is not None:  # per row the first row of its group (K11's probe against the group-by's own table)
                L.check(self.lib.rfx_hip_join_probe_hash(self._ctx, key.data_ptr(), n, C.byref(t), _probe_first.data_ptr()), "join_probe_hash")
            L.check(self.lib.rfx_hip_hash_rank(self._ctx, C.byref(t), total_rows, C.byref(ng)), "hash_rank")
        g = int(ng.value)
        keys = self.empty(g)
        first = self.empty(g)
        results = [self.empty(g, d) for d in out_dtypes]
        ptrs = (C.c_void_p * max(1, nagg))(*[r.data_ptr() for r in results])
        if dense:
            L.check(self.lib.rfx_hip_group_emit_sharded(self._ctx, aarr, C.byref(t), row0, n if _collective is not None else 0, keys.data_ptr(),
                                                        first.data_ptr(), ptrs), "group_emit")
        else:
            L.check(self.lib.rfx_hip_hash_emit_sharded(self._ctx, aarr, C.byref(t), row0, n if _collective is not None else 0, keys.data_ptr(),
                                                       first.data_ptr(), ptrs), "hash_emit")
        if _collective is not None:  # FIRST: only the rank that owns a group's first row had its value; the others emitted 0
            firsts = [results[i] for i in range(nagg) if int(aarr[i].kind) == L.RFX_AGG_FIRST]
            if firsts:
                _collective("first_values", firsts)
        r = dict(groups=g, keys=keys, first=first, results=results, dense=dense, cap=0 if dense else cap)
        if multi is not None:
            mins, mults, ranges = multi
            r["key_columns"] = []
            for mn, mu, rg in zip(mins, mults, ranges):
                kc = self.empty(g)
                L.check(self.lib.rfx_hip_composite_decode(self._ctx, keys.data_ptr(), g, mn, mu, rg, kc.data_ptr()), "composite_decode")
                r["key_columns"].append(kc)
        self.sync()
        return r

    def row_hash(self, kcols, value_first: bool = False) -> torch.Tensor:
        """The reference's row hash of the key tuples (__index_list_precalc_hash, core/index.c:274-309) as one i64 column."""
        k, n = len(kcols), kcols[0].numel()
        out = self.empty(n)
        ptrs = (C.c_void_p * k)(*[kc.data_ptr() for kc in kcols])
        L.check(self.lib.rfx_hip_row_hash(self._ctx, ptrs, k, n, int(value_first), out.data_ptr()), "row_hash")
        return out

    def _group_by_row_hash(self, kcols, scopes, aggs, where, table, total_rows, row0, _collective, order):
        """Key tuples that do not fold into one 64-bit composite (H2O Q7: six keys; or null keys): group on the reference's
        64-bit row hash with the sparse-key machinery, and PROVE the grouping: every key column rides along as a (min, max)
        aggregate pair, and a group whose rows all agree on every key column is exactly one tuple.  A group with min != max
        is a hash collision (two tuples, one hash; probability ~ groups^2 / 2^65) and raises instead of answering wrongly.
        Nulls are keys like any other there (the reference compares tuples bitwise): min / max skip nulls, so a column that
        holds nulls is checked through a copy with the nulls replaced by a value above its maximum."""
        n = kcols[0].numel()
        h = self.row_hash(kcols, value_first=where is not None)
        if _collective is None and row0 == 0:
            # One GPU: the reference's own proof shape -- the tuple comparison of every row with its group's first row
            # (__index_list_cmp_row, core/index.c:2731-2790) -- done once after grouping instead of on every probe: per row the first
            # row of its group (the join probe against the group-by's own hashed table), then one gather + compare per key column;
            # the key columns of the result are the tuples at the groups' first rows.  Twelve (min, max) proof aggregates cost
            # 1.2e9 device atomics per 1e8 rows (two launches, 94 ms); this costs seven random reads per row.
            ids = self.empty(n)
            r = self.group_by(h, list(aggs), where, table, total_rows, row0, None, _probe_first=ids)
            if r["dense"]:
                raise RfxError("row-hash group-by: the hash column took the dense path")
            chk = self.empty(n)
            for kc in kcols:
                L.check(self.lib.rfx_hip_gather_or(self._ctx, kc.data_ptr(), kc.data_ptr(), ids.data_ptr(), n, 0, chk.data_ptr()), "gather_or")
                if not bool(torch.equal(chk, kc)):  # an unselected row probes nothing (null id) and compares with itself
                    raise RfxError("row-hash group-by: two key tuples share one 64-bit row hash (collision); not answered on this path")
            del chk, ids
            r["key_columns"] = [self.at_ids(kc, r["first"]) if r["groups"] else self.empty(0) for kc in kcols]
            if order == "radix" and r["groups"] > 1:  # (hash & 1023, first occurrence): core/index.c:2465-2729
                perm = torch.argsort((r["keys"] & 1023) * (1 << 40) + torch.argsort(torch.argsort(r["first"])), stable=True)
                r["keys"], r["first"] = r["keys"][perm], r["first"][perm]
                r["results"] = [x[perm] for x in r["results"]]
                r["key_columns"] = [x[perm] for x in r["key_columns"]]
            return r
        checks, repl = [], []
        for kc, (mn, mx) in zip(kcols, scopes):
            if mn == L.NULL_I64:  # scope saw a null (INT64_MIN sorts lowest)
                if mx == 2**63 - 1:
                    raise RfxError("row-hash group-by: a key column holds both nulls and INT64_MAX; no spare value for the collision proof")
                c2 = self.empty(n)
                L.check(self.lib.rfx_hip_replace_null_i64(self._ctx, kc.data_ptr(), n, mx + 1, c2.data_ptr()), "replace_null_i64")
                checks.append(c2)
                repl.append(mx + 1)
            else:
                checks.append(kc)
                repl.append(None)
        extra = [(fn, c) for c in checks for fn in ("min", "max")]
        r = self.group_by(h, list(aggs) + extra, where, table, total_rows, row0, _collective)
        res, ex = r["results"][:len(aggs)], r["results"][len(aggs):]
        key_columns = []
        for i, rp in enumerate(repl):
            mn, mx = ex[2 * i], ex[2 * i + 1]
            if r["groups"] and not bool(torch.equal(mn, mx)):
                raise RfxError("row-hash group-by: two key tuples share one 64-bit row hash (collision); not answered on this path")
            key_columns.append(mx if rp is None else torch.where(mx == rp, torch.full_like(mx, L.NULL_I64), mx))
        r["results"], r["key_columns"] = res, key_columns
        if order == "radix" and r["groups"] > 1:  # (hash & 1023, first occurrence): core/index.c:2465-2729
            perm = torch.argsort((r["keys"] & 1023) * (1 << 40) + torch.argsort(torch.argsort(r["first"])), stable=True)
            r["keys"], r["first"] = r["keys"][perm], r["first"][perm]
            r["results"] = [x[perm] for x in r["results"]]
            r["key_columns"] = [x[perm] for x in r["key_columns"]]
        return r

    # ------------------------------------------------------------------ equi-joins: lj / ij, core/join.c:158-298
    def join_index(self, keys, left: Dict[str, torch.Tensor], right: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Per LEFT row the FIRST right row with an equal key tuple, or null (index_left_join_obj, core/index.c:2886-2928;
        one key column: ray_find).  Build = the group-by's first-occurrence table over the right keys (no aggregates), probe =
        one pass over the left keys.  Several key columns probe on the reference's row hash of both sides and are then
        compared column by column at the matched rows; a mismatch (two tuples, one hash) raises."""
        keys = [keys] if isinstance(keys, str) else list(keys)
        lk = [self._check_col(self._resolve(k, left)) for k in keys]
        rk = [self._check_col(self._resolve(k, right)) for k in keys]
        if any(c.dtype != torch.int64 for c in lk + rk):
            raise RfxError("join keys must be i64-like columns on this path")
        nl, nr = lk[0].numel(), rk[0].numel()
        ids = self.empty(nl)
        if nl == 0:
            return ids
        if nr == 0:
            return ids.fill_(L.NULL_I64)
        exact = True
        if len(keys) == 1:
            lkey, rkey = lk[0], rk[0]
        else:
            # ranges (over BOTH sides) that multiply into 64 bits: one injective composite key per side, as the group-by's
            # "perfect" path builds it -- exact, no hashing; else the reference's own route, the row hash
            k = len(keys)
            mins, maxs = [], []
            for lc, rc_ in zip(lk, rk):
                a, b = self.scope(lc), self.scope(rc_)
                mins.append(min(a[0], b[0]))
                maxs.append(max(a[1], b[1]))
            amin, amax, amul, tmax = (C.c_int64 * k)(*mins), (C.c_int64 * k)(*maxs), (C.c_int64 * k)(), C.c_int64()
            if self.lib.rfx_composite_plan(amin, amax, k, amul, C.byref(tmax)) == L.RFX_OK:
                lkey, rkey = self.empty(nl), self.empty(nr)
                for cols, n_, out in ((lk, nl, lkey), (rk, nr, rkey)):
                    ptrs = (C.c_void_p * k)(*[c.data_ptr() for c in cols])
                    L.check(self.lib.rfx_hip_composite_key(self._ctx, ptrs, amin, amul, k, n_, out.data_ptr()), "composite_key")
            else:
                lkey, rkey, exact = self.row_hash(lk), self.row_hash(rk), False
        kmin, kmax, seen = self.scope(rkey)
        rng = kmax - kmin + 1
        aarr = (L.Agg * 1)()
        # dense first-occurrence table where the group-by would choose one (range <= rows), and beyond that while it stays small
        # next to the right side (8 B per slot, <= 4 x rows or 16 M slots): a table fill is cheaper than hashing every right row
        if 0 < rng <= max(seen, 4 * nr, 1 << 24) and rng <= (1 << 29) and kmin != L.NULL_I64:
            t, store, _ = self.group_tables(aarr, 0, kmin, rng)
            L.check(self.lib.rfx_hip_group_tables_init(self._ctx, aarr, C.byref(t)), "group_tables_init")
            L.check(self.lib.rfx_hip_group_dense_accumulate(self._ctx, rkey.data_ptr(), None, 0, L.RFX_AND, aarr, nr, 0, C.byref(t)), "group_dense_accumulate")
            L.check(self.lib.rfx_hip_join_probe_dense(self._ctx, lkey.data_ptr(), nl, kmin, rng, t.d_first, ids.data_ptr()), "join_probe_dense")
        else:
            cap_max = 1 << max(4, math.ceil(math.log2(max(2 * nr, 16))))
            cap = min(cap_max, 1 << 22)
            while True:
                t, store, _ = self.group_tables(aarr, 0, 0, cap, hashed=True)
                L.check(self.lib.rfx_hip_hash_tables_init(self._ctx, aarr, C.byref(t)), "hash_tables_init")
                rc = self.lib.rfx_hip_group_hash_accumulate(self._ctx, rkey.data_ptr(), None, 0, L.RFX_AND, aarr, nr, 0, C.byref(t))
                if rc != L.RFX_ELIMIT or cap >= cap_max:
                    L.check(rc, "group_hash_accumulate")
                    break
                del t, store
                cap = min(cap_max, cap << 4)
            L.check(self.lib.rfx_hip_join_probe_hash(self._ctx, lkey.data_ptr(), nl, C.byref(t), ids.data_ptr()), "join_probe_hash")
        if not exact:  # the tuple comparison the reference does on every probe (__index_list_cmp_row), done once on the result
            chk = self.empty(nl)
            for lc, rc_ in zip(lk, rk):
                L.check(self.lib.rfx_hip_gather_or(self._ctx, rc_.data_ptr(), lc.data_ptr(), ids.data_ptr(), nl, 0, chk.data_ptr()), "gather_or")
                if not bool(torch.equal(chk, lc)):
                    raise RfxError("join: two key tuples share one 64-bit row hash (collision); not answered on this path")
        self.sync()
        return ids

    def _join_fill(self, col: torch.Tensor) -> int:
        return 0x7FF8000000000000 if col.dtype == torch.float64 else (1 << 63)  # NaN / NULL_I64 bit patterns

    def left_join(self, keys, left: Dict[str, torch.Tensor], right: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """``(lj [keys] left right)`` -- ray_left_join, core/join.c:158-198: every left row; a non-key column that the right
        table has takes the matched right row's value, else the left row's own (null when the left table lacks the column);
        columns: keys, then the other left columns, then the right-only ones.  Empty side -> the left table."""
        keys = [keys] if isinstance(keys, str) else list(keys)
        nl = next(iter(left.values())).numel() if left else 0
        nr = next(iter(right.values())).numel() if right else 0
        if nl == 0 or nr == 0:
            return dict(left)
        ids = self.join_index(keys, left, right)
        out = {k: left[k] for k in keys}
        for name in [c for c in left if c not in keys] + [c for c in right if c not in keys and c not in left]:
            if name not in right:
                out[name] = left[name]
                continue
            rc, lc = right[name], left.get(name)
            if lc is not None and lc.dtype != rc.dtype:
                raise RfxError(f"join: column {name} has different types in the two tables")
            o = torch.empty(nl, dtype=rc.dtype, device=self.device)
            L.check(self.lib.rfx_hip_gather_or(self._ctx, rc.data_ptr(), lc.data_ptr() if lc is not None else None, ids.data_ptr(), nl, self._join_fill(rc),
                                               o.data_ptr()), "gather_or")
            out[name] = o
        self.sync()
        return out

    def inner_join(self, keys, left: Dict[str, torch.Tensor], right: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """``(ij [keys] left right)`` -- ray_inner_join, core/join.c:200-298: the left rows that have a match, in left order,
        paired with their first matching right row; a column the right table has comes from the right row."""
        keys = [keys] if isinstance(keys, str) else list(keys)
        nl = next(iter(left.values())).numel() if left else 0
        nr = next(iter(right.values())).numel() if right else 0
        if nl == 0 or nr == 0:
            return dict(left)
        ids = self.join_index(keys, left, right)
        lids = self.where(("!=", ids, None))  # ascending left rows with a match
        rids = self.at_ids(ids, lids)
        out = {}
        for name in keys + [c for c in left if c not in keys] + [c for c in right if c not in keys and c not in left]:
            if name in right:
                if name in left and left[name].dtype != right[name].dtype:
                    raise RfxError(f"join: column {name} has different types in the two tables")
                out[name] = self.at_ids(right[name], rids)
            else:
                out[name] = self.at_ids(left[name], lids)
        return out

    def _key_col(self, spec, table) -> torch.Tensor:
        """A `by:` entry: a column, or ("xbar", column, width) -- the bucketed key is evaluated once into a scratch column
        (ray_xbar, core/math.c:1635; the reference does the same before grouping)."""
        if isinstance(spec, tuple) and len(spec) == 3 and spec[0] == "xbar":
            col = self._check_col(self._resolve(spec[1], table))
            if col.dtype != torch.int64:
                raise RfxError("xbar over a non-integer column is not on this path")
            out = self.empty(col.numel())
            L.check(self.lib.rfx_hip_xbar_i64(self._ctx, col.data_ptr(), col.numel(), int(spec[2]), out.data_ptr()), "xbar_i64")
            return out
        return self._check_col(self._resolve(spec, table))

    def _scopes_fused(self, kcols, where, table, n):
        """(min, max, rows seen) of two to four key columns in ONE pass: K1 with a min and a max aggregate per column reads every
        key and predicate column once (one index_scope_i64 pass per column re-reads the predicate columns each time).  K1's min / max
        skip nulls where the reference's scope takes INT64_MIN as the smallest key: a column whose non-null count is below the
        selected-row count holds a null, so its minimum is null.  None when the shape does not fit one K1 launch."""
        try:
            part = self.filter_aggr_partials([(fn, kc) for kc in kcols for fn in ("min", "max")], where, table, nrows=n)
        except RfxError:
            return None
        p = part.view(torch.int64).reshape(-1, 8).cpu().tolist()  # rfx_partial_t: isum, fsum, cnt, ext, pos, ...
        seen = p[2 * len(kcols)][2]
        out = []
        for i in range(len(kcols)):
            nonnull, mn, mx = p[2 * i][2], p[2 * i][3], p[2 * i + 1][3]
            if seen == 0:
                out.append((0, -1, 0))
            elif nonnull == 0:
                out.append((L.NULL_I64, L.NULL_I64, seen))
            else:
                out.append((L.NULL_I64 if nonnull < seen else mn, mx, seen))
        return out

    SPEC_MAX_SLOTS = 1 << 14  # sampled scopes only for ranges whose tables are LDS-sized (the kernels that report out-of-scope keys)
    # ... and at most a tenth of the sample (2^18 rows): a uniformly drawn extreme value is then missed with probability e^-10

    def _may_speculate(self, n: int, _collective, key: Optional[torch.Tensor] = None) -> bool:
        """A sampled key scope instead of the full scope pass?  Large single-GPU inputs only; not for a key column whose sampled
        scope was reported too small before (a rare extreme value: the sample would miss it again); RFX_NO_SAMPLED_SCOPE=1 turns it off."""
        if key is not None and (key.data_ptr(), key.numel()) in self.__dict__.setdefault("_spec_failed", set()):  # (per engine)
            return False
        return n >= (1 << 24) and _collective is None and not os.environ.get("RFX_NO_SAMPLED_SCOPE")

    def scope_sample(self, key: torch.Tensor) -> Tuple[int, int]:
        """[min, max] of 2^14 strided rows + the column's first and last 2^11 (rfx_hip_scope_sample_i64).  (syncs)"""
        mn, mx = C.c_int64(), C.c_int64()
        L.check(self.lib.rfx_hip_scope_sample_i64(self._ctx, key.data_ptr(), key.numel(), C.byref(mn), C.byref(mx)), "scope_sample")
        return int(mn.value), int(mx.value)

    def _composite_plan(self, kcols, where, table, _collective, sampled: bool = False):
        """Scopes of every key column (through the predicates) and the reference's multiplier plan (core/index.c:2340-2383).
        Returns (composite max, rows seen, (mins, mults, ranges))."""
        if len(kcols) > L.RFX_MAX_KEYS:
            raise RfxError(f"at most {L.RFX_MAX_KEYS} key columns")
        n = kcols[0].numel()
        mins, maxs, seen = [], [], 0
        for kc in kcols:
            if kc.dtype != torch.int64 or kc.numel() != n:
                raise RfxError("key columns must be equally long i64 columns on this path")
        # (only under a filter: unfiltered, one 1.2 ms scope pass per key column beats K1 with 2 x keys min / max aggregates --
        #  two keys, 1e9 rows: 2.4 against 3.8 ms -- while with predicates every separate pass re-reads the predicate columns)
        if sampled:  # every column's scope from a sample; (None: not usable -- a null key, or tables beyond the LDS forms)
            prod = 1
            for kc in kcols:
                mn, mx = self.scope_sample(kc)
                if mn == L.NULL_I64 or mx < mn:
                    return -1, 0, None
                prod *= mx - mn + 1
                mins.append(mn)
                maxs.append(mx)
            if prod > self.SPEC_MAX_SLOTS:
                return -1, 0, None
            k = len(kcols)
            amin, amax, amul, tmax = (C.c_int64 * k)(*mins), (C.c_int64 * k)(*maxs), (C.c_int64 * k)(), C.c_int64()
            if self.lib.rfx_composite_plan(amin, amax, k, amul, C.byref(tmax)) != L.RFX_OK:
                return -1, 0, None
            return int(tmax.value), n, (mins, list(amul), [mx - mn + 1 for mn, mx in zip(mins, maxs)])
        local = self._scopes_fused(kcols, where, table, n) if (where is not None and 2 <= len(kcols) <= 4) else None
        for i, kc in enumerate(kcols):
            mn, mx, seen = local[i] if local is not None else self.scope(kc, where, table)
            if _collective is not None:
                mn, mx, seen = _collective("scope", (mn, mx, seen, self.device))
            mins.append(mn)
            maxs.append(mx)
        k = len(kcols)
        if seen == 0:
            return -1, 0, ([0] * k, [1] * k, [1] * k)
        amin, amax, amul = (C.c_int64 * k)(*mins), (C.c_int64 * k)(*maxs), (C.c_int64 * k)()
        tmax = C.c_int64()
        if self.lib.rfx_composite_plan(amin, amax, k, amul, C.byref(tmax)) == L.RFX_ELIMIT:
            raise _NotPerfect(list(zip(mins, maxs)), seen)
        return int(tmax.value), seen, (mins, list(amul), [mx - mn + 1 for mn, mx in zip(mins, maxs)])

    # ------------------------------------------------------------------ the select surface (core/query.c:607-654)
    def select(self, query: Dict) -> Dict[str, torch.Tensor]:
        """``(select {name: (fn col) ... from: t where: p by: k})`` with t a dict of equally long columns.

        Keys ``from`` (required), ``where``, ``by`` (column name) are clauses; every other key is an output column
        given as ``(fn, colname)``.  Without aggregates the filtered (and ungrouped) columns are returned, like
        select_collect_fields (core/query.c:474-557).  Result: dict name -> device tensor, group key first.
        """
        if "from" not in query:
            raise RfxError("'select' expects 'from' param")  # core/query.c:281
        table = query["from"]
        where, by = query.get("where"), query.get("by")
        outs = [(k, v) for k, v in query.items() if k not in ("from", "where", "by", "take", "order")]
        lens = {int(c.numel()) for c in table.values()}
        if len(lens) > 1:
            raise RfxError("table columns differ in length")
        n = lens.pop() if lens else 0
        if by is not None:
            aggs = [(fn, col) for _, (fn, col) in outs]
            if isinstance(by, dict):  # by: {name: column ...}
                r = self.group_by(list(by.values()), aggs, where, table, order=query.get("order", "first"))
                res = dict(zip(by.keys(), r["key_columns"])) if len(by) > 1 else {next(iter(by)): r["keys"]}
            else:
                r = self.group_by(by, aggs, where, table)
                res = {by if isinstance(by, str) else "by": r["keys"]}
            for (name, _), col in zip(outs, r["results"]):
                res[name] = col
            return res
        if outs:
            aggs = [(fn, col) for _, (fn, col) in outs]
            vals, _ = self.filter_aggr(aggs, where, table, nrows=n)
            res = {}
            for (name, (fn, col)), v in zip(outs, vals):
                f64 = fn == "avg" or (col is not None and fn != "count" and self._arg_f64(col, table))
                if v is None:
                    v = float("nan") if f64 else L.NULL_I64
                res[name] = torch.tensor([v], dtype=torch.float64 if f64 else torch.int64, device=self.device)
            return res
        if where is None:
            return dict(table)
        ids = self.where(where, table)
        return {name: self.at_ids(col, ids) for name, col in table.items()}


class _NotFlat(Exception):
    pass


class _More(tuple):
    """A comparison that shares its parenthesis with the next one (rfx_pred_t::more)."""


class _NotPerfect(Exception):
    """The key ranges do not multiply into a 64-bit composite key (index_group_list_perfect gives up, core/index.c:2364-2383)."""

    def __init__(self, scopes, seen):
        super().__init__("key ranges overflow the composite key")
        self.scopes, self.seen = scopes, seen
