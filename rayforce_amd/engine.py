"""Python host of librfx.so for the tests and bench.py: the reference's select / where / by surface over HBM-resident columns.

There is NO planning here.  Every query goes to the library's planner (include/rfx_exec.h: rfx_exec_filter_aggr / rfx_exec_where /
rfx_exec_group_by / rfx_exec_join_index) or, for `select`, through the C operator door itself (rfx_select on device-column handles);
this file only marshals: tuples -> rfx_pred_t / rfx_agg_t descriptors, torch tensors -> device addresses, result blocks -> tensors.
Columns are 1-D ``torch.int64`` / ``torch.float64`` CUDA tensors (torch owns memory and the stream); there is no CPU fallback.

Predicates are tuples ``(op, lhs, rhs)`` with op in ``== != < > <= >=``, lhs a column (tensor or table column name) or an element-wise
expression ``(+|-|*|div|/|% x y)``, rhs a Python int / float atom, a column or an expression; ``("and", ...)`` / ``("or", ...)`` nest
freely: up to eight comparisons in four levels run fused in the query's one pass (rfx_pred_t's tree form), anything wider is evaluated
into a B8 mask first (mask_of) and handed to the planner as the selection.  Names follow RayforceDB (core/env.c:135-225).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib as L
from . import marshal as M
from ._lib import RfxError
from .marshal import NotFused as _NotFused, ctype_of as _ctype_of

Column = torch.Tensor


class _Owner:
    """Holds a result struct of the planner (rfx_groups_t / rfx_ids_t) until no tensor views its device blocks any more."""

    def __init__(self, eng, struct, free):
        self.eng, self.struct, self.free = eng, struct, free

    def __del__(self):
        try:
            if self.eng._x:
                self.free(self.eng._x, C.byref(self.struct))
        except Exception:  # pragma: no cover -- interpreter shutdown
            pass


class _View:
    def __init__(self, owner, ptr: int, n: int, typestr: str):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class Engine:
    """One GPU.  ``shards=k`` splits every query row-range over k contexts on that GPU (each on its own stream and host thread, merged by
    the planner's device kernels): the logic one evaluator process runs over the GPUs of a node, testable on one.

    An Engine is driven by ONE Python thread at a time (the planner and its contexts are not re-entrant).  Results hold their device blocks
    through an owner object whose finaliser may run on whatever thread the garbage collector picks: that is safe -- it only hands blocks
    back to the contexts' pools, which share one process-wide lock (rfx_ctx.hip) -- but it is the only cross-thread call that is."""

    def __init__(self, device: Union[int, torch.device, None] = None, shards: int = 1):
        self.lib = L.load_library()
        if not torch.cuda.is_available():
            raise RfxError("no GPU visible to torch: the MI355X engine has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        torch.cuda.set_device(self.device)
        self.shards = int(shards)
        self._ctxs = (C.c_void_p * self.shards)()
        # shard 0 runs on torch's current stream (tensor ops and librfx kernels are ordered without extra syncs); further shards own theirs
        stream = torch.cuda.current_stream(self.device).cuda_stream or 1  # 0 = legacy default stream -> RFX_STREAM_LEGACY
        for s in range(self.shards):
            c = C.c_void_p()
            L.check(self.lib.rfx_hip_ctx_create(self.device.index, C.c_void_p(stream if s == 0 else 0), C.byref(c)), "ctx_create")
            self._ctxs[s] = c.value
        self._ctx = C.c_void_p(self._ctxs[0])
        self._x = C.c_void_p()
        L.check(self.lib.rfx_exec_create(self._ctxs, self.shards, C.byref(self._x)), "exec_create")
        self._keep: List = []

    def close(self) -> None:
        if self._x:
            self.lib.rfx_exec_destroy(self._x)
            self._x = C.c_void_p()
            for s in range(self.shards):
                self.lib.rfx_hip_ctx_destroy(C.c_void_p(self._ctxs[s]))
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _xcheck(self, rc: int, what: str) -> None:
        if rc != L.RFX_OK:
            raise RfxError(f"{what} failed with code {rc}: {self.lib.rfx_exec_last_error(self._x).decode(errors='replace')}")

    def sync(self) -> None:
        L.check(self.lib.rfx_hip_ctx_sync(self._ctx), "sync")

    def trim(self) -> None:
        """Give the blocks the contexts keep for reuse back to the device (before a phase that needs the memory for columns)."""
        for s in range(self.shards):
            L.check(self.lib.rfx_hip_ctx_trim(C.c_void_p(self._ctxs[s])), "ctx_trim")

    def tune(self, blocks_per_cu: int = 0, flags: int = 0) -> None:
        for s in range(self.shards):
            L.check(self.lib.rfx_hip_ctx_tune(C.c_void_p(self._ctxs[s]), blocks_per_cu, flags), "tune")

    def stat(self, which: int) -> int:
        """Path counters (include/rfx_hip.h RFX_STAT_*), summed over the shards."""
        return sum(int(self.lib.rfx_hip_ctx_stat(C.c_void_p(self._ctxs[s]), which)) for s in range(self.shards))

    def xstat(self, which: int) -> int:
        """The planner's counters (include/rfx_exec.h RFX_XSTAT_*)."""
        return int(self.lib.rfx_exec_stat(self._x, which))

    @property
    def spec_retries(self) -> int:
        return self.xstat(L.RFX_XSTAT_SCOPE_RETRIED)

    def forget_scopes(self) -> None:
        self.lib.rfx_exec_forget_scopes(self._x)

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
        """Upload a numpy int64 / float64 / int8 / bool array (host -> HBM over PCIe; not part of any timed region)."""
        t = torch.as_tensor(host_array)
        if t.dtype not in (torch.int64, torch.float64, torch.int8, torch.bool):
            raise RfxError(f"unsupported dtype {t.dtype}")
        return (t.to(torch.int8) if t.dtype == torch.bool else t).contiguous().to(self.device)

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

    # on-disk columns: rayforce_amd/colfiles.py
    def load_column(self, path: str) -> torch.Tensor:
        from . import colfiles
        return colfiles.load_column(self, path)

    def load_splayed(self, directory: str, columns: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
        from . import colfiles
        return colfiles.load_splayed(self, directory, columns)

    def load_parted(self, root: str, table: str, columns: Optional[Sequence[str]] = None, where=None) -> Dict[str, torch.Tensor]:
        from . import colfiles
        return colfiles.load_parted(self, root, table, columns, where)

    # ------------------------------------------------------------------ marshalling: tuples -> descriptors (rayforce_amd/marshal.py)
    def _check_col(self, t, n=None):
        return M.check_col(self, t, n)

    def _resolve(self, x, table):
        return M.resolve(self, x, table)

    def _query(self, where, aggs, table, nrows: Optional[int], keys=None, flags: int = 0) -> Tuple[L.Query, int]:
        """The planner's query over this engine's shards: descriptors in shard 0's addresses + every column's address per shard."""
        self._keep.clear()
        q = L.Query()
        n = nrows
        if isinstance(where, torch.Tensor):  # a B8 mask as the selection
            m = self._check_col(where.view(torch.int8) if where.dtype == torch.bool else where, n)
            if m.dtype != torch.int8:
                raise RfxError("where expects a B8 mask")  # reference: err_type (core/items.c:1395)
            q.d_mask, n = m.data_ptr(), m.numel()
            self._keep.append(m)
        else:
            try:
                logic, leaves, more = M.tree_leaves(self, where)
            except _NotFused:
                return self._query(self.mask_of(where, table), aggs, table, nrows, keys, flags)
            parr, n = M.preds(self, leaves, more, table, n)
            q.preds, q.npred, q.logic = parr, len(leaves), logic
            self._keep.append(parr)
        aarr, n = M.aggs(self, aggs or [], table, n)
        q.aggs, q.nagg = aarr, len(aggs or [])
        self._keep.append(aarr)
        if keys:
            kx = (C.c_int64 * len(keys))()
            kp = (C.c_void_p * len(keys))()
            for i, k in enumerate(keys):
                if isinstance(k, tuple) and len(k) == 3 and k[0] == "xbar":  # bucketed key: evaluated by the planner (ray_xbar, core/math.c:1635)
                    if int(k[2]) <= 0:
                        raise RfxError("xbar: width must be positive")
                    col, kx[i] = self._check_col(self._resolve(k[1], table), n), int(k[2])
                else:
                    col = self._check_col(self._resolve(k, table), n)
                if col.dtype != torch.int64:
                    raise RfxError("group key must be i64 on this path (f64 keys group on their bit pattern: view as int64)")
                n = col.numel() if n is None else n
                kp[i] = col.data_ptr()
                self._keep.append(col)
            q.d_keys, q.kxbar, q.nkeys = kp, kx, len(keys)
            self._keep += [kp, kx]
        if n is None:
            raise RfxError("cannot infer the row count (count without column and without predicate needs nrows=)")
        q.nrows, q.flags = n, flags
        if self.shards > 1:  # shards of one device: every column's slice by the planner's own split rule
            ptrs = {}
            for t in self._keep:
                if isinstance(t, torch.Tensor) and t.numel() == n:
                    ptrs[t.data_ptr()] = t.element_size()
            cols = (L.QCol * max(1, len(ptrs)))()
            r0 = C.c_int64()
            for i, (p, esz) in enumerate(ptrs.items()):
                for s in range(self.shards):
                    self.lib.rfx_exec_split(n, self.shards, s, C.byref(r0), None)
                    cols[i].d[s] = p + r0.value * esz
            q.cols, q.ncols = cols, len(ptrs)
            self._keep.append(cols)
            torch.cuda.current_stream(self.device).synchronize()  # the other shards run on their own streams
        return q, n

    @staticmethod
    def _value(v: L.Value):
        if v.type == L.RFX_F64:
            return float("nan") if v.is_null else float(v.f)
        return None if v.is_null else int(v.i)

    def _view(self, owner, d_src: int, n: int, dtype) -> torch.Tensor:
        """A planner-owned device column as a torch tensor WITHOUT a copy: torch aliases the memory through __cuda_array_interface__ and keeps
        `owner` alive; the planner's blocks are released when the last tensor over them dies."""
        if not n or not d_src:
            return self.empty(0, dtype)
        return torch.as_tensor(_View(owner, d_src, n, "<f8" if dtype == torch.float64 else "<i8"), device=self.device)

    # ------------------------------------------------------------------ scalar aggregates (K1/K5)
    def filter_aggr(self, aggs, where=None, table=None, nrows: Optional[int] = None):
        """``select {aggs} from t where p`` without ``by:`` -> ([values], selected_rows).  (syncs)"""
        q, _ = self._query(where, aggs, table, nrows)
        vals = (L.Value * max(1, len(aggs)))()
        sel = C.c_int64()
        self._xcheck(self.lib.rfx_exec_filter_aggr(self._x, C.byref(q), vals, C.byref(sel)), "filter_aggr")
        return [self._value(vals[i]) for i in range(len(aggs))], int(sel.value)

    # scalar verbs of the reference (core/math.c:2388-2526, core/misc.c:43-60)
    def sum(self, col, where=None, table=None): return self.filter_aggr([("sum", col)], where, table)[0][0]
    def min(self, col, where=None, table=None): return self.filter_aggr([("min", col)], where, table)[0][0]
    def max(self, col, where=None, table=None): return self.filter_aggr([("max", col)], where, table)[0][0]
    def avg(self, col, where=None, table=None): return self.filter_aggr([("avg", col)], where, table)[0][0]
    def count(self, col, where=None, table=None): return self.filter_aggr([("count", col)], where, table)[0][0]
    def first(self, col, where=None, table=None): return self.filter_aggr([("first", col)], where, table)[0][0]

    # ------------------------------------------------------------------ K2: masks (API parity with the reference's B8 results)
    def cmp(self, op: str, lhs, rhs, table=None) -> torch.Tensor:
        """ray_eq .. ray_ge on a column: B8 byte mask (int8 tensor of 0/1)."""
        self._keep.clear()
        parr, n = M.preds(self, [(op, lhs, rhs)], [0], table, None)
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

    def and_(self, *masks) -> torch.Tensor: return self._logic(L.RFX_AND, masks)
    def or_(self, *masks) -> torch.Tensor: return self._logic(L.RFX_OR, masks)

    def mask_of(self, where, table=None) -> torch.Tensor:
        """Materialise any predicate tree as a byte mask (the reference's own evaluation order, on the GPU)."""
        head = where[0]
        if head in L.OPS:
            return self.cmp(head, where[1], where[2], table)
        subs = [self.mask_of(w, table) for w in where[1:]]
        return self.and_(*subs) if head == "and" else self.or_(*subs)

    # ------------------------------------------------------------------ K3: where, K4: gather
    def where(self, where, table=None, row0: int = 0) -> torch.Tensor:
        """Ascending row ids (row0 + row) of the selected rows.  `where` = int8/bool mask tensor or a predicate tree.  (syncs)"""
        q, _ = self._query(where, [], table, None)
        q.row0 = row0
        ids = L.Ids()
        self._xcheck(self.lib.rfx_exec_where(self._x, C.byref(q), C.byref(ids)), "where")
        owner = _Owner(self, ids, self.lib.rfx_exec_ids_free)
        if ids.nshards == 1:
            return self._view(owner, ids.d_ids[0], int(ids.total), torch.int64)
        out, at = self.empty(int(ids.total)), 0
        for s in range(ids.nshards):  # shard order = row order
            if ids.count[s]:
                L.check(self.lib.rfx_hip_d2d(self._ctx, out.data_ptr() + at * 8, C.c_void_p(ids.d_ids[s]), ids.count[s] * 8), "d2d")
                at += ids.count[s]
        self.sync()
        return out

    def at_ids(self, col: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        self._check_col(col)
        self._check_col(ids)
        if ids.dtype != torch.int64:
            raise RfxError("ids must be i64")
        _ctype_of(col)
        out = torch.empty(ids.numel(), dtype=col.dtype, device=self.device)
        L.check(self.lib.rfx_hip_gather(self._ctx, col.data_ptr(), ids.data_ptr(), ids.numel(), out.data_ptr()), "gather")
        return out

    def eval_expr(self, expr, table=None) -> torch.Tensor:
        """``(op x y)`` / an expression tree over columns and atoms as a device column: ray_add .. ray_mod (binop_map,
        core/math.c:2280-2345) in ONE pass whatever the depth."""
        a = L.Agg()
        keep, self._keep = self._keep, []
        n = M.agg_expr(self, a, expr, table, None)
        out = torch.empty(n, dtype=torch.float64 if L.agg_input_type(a) == L.RFX_F64 else torch.int64, device=self.device)
        t = C.c_int32()
        L.check(self.lib.rfx_hip_eval_expr(self._ctx, C.byref(a), n, out.data_ptr(), C.byref(t)), "eval_expr")
        self._keep = keep
        return out

    def scope(self, key: torch.Tensor, where=None, table=None) -> Tuple[int, int, int]:
        """index_scope_i64 (core/index.c:376-435): (min, max, rows seen) of a key column through the predicates.  (syncs)"""
        self._keep.clear()
        logic, leaves, more = M.tree_leaves(self, where)
        parr, n = M.preds(self, leaves, more, table, self._check_col(key).numel())
        mn, mx, cnt = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.lib.rfx_hip_scope_i64(self._ctx, key.data_ptr(), parr, len(leaves), logic, n, C.byref(mn), C.byref(mx), C.byref(cnt)), "scope_i64")
        return int(mn.value), int(mx.value), int(cnt.value)

    def row_hash(self, kcols, value_first: bool = False) -> torch.Tensor:
        """The reference's row hash of the key tuples (__index_list_precalc_hash, core/index.c:274-309) as one i64 column."""
        k, n = len(kcols), kcols[0].numel()
        out = self.empty(n)
        ptrs = (C.c_void_p * k)(*[kc.data_ptr() for kc in kcols])
        L.check(self.lib.rfx_hip_row_hash(self._ctx, ptrs, k, n, int(value_first), out.data_ptr()), "row_hash")
        return out

    # ------------------------------------------------------------------ group-by (K6-K10): the planner's
    def group_by(self, key, aggs, where=None, table=None, order: str = "first", flags: int = 0, probe_first: bool = False):
        """``select {aggs} from t [where p] by key`` -> dict(groups=, keys=, first=, results=[...], dense=, cap=, path=[, key_columns=]) of
        device tensors, groups in first-occurrence order (`order="radix"`, key tuples on the row-hash path only: the order of the
        reference's multi-threaded radix grouping -- hash & 1023, then first occurrence, core/index.c:2465-2729).  (syncs)"""
        keys = list(key) if isinstance(key, (list, tuple)) and not (isinstance(key, tuple) and len(key) == 3 and key[0] == "xbar") else [key]
        q, n = self._query(where, aggs, table, None, keys, flags | L.RFX_Q_WANT_FIRST | (L.RFX_Q_PROBE_FIRST if probe_first else 0))
        g = L.Groups()
        self._xcheck(self.lib.rfx_exec_group_by(self._x, C.byref(q), C.byref(g)), "group_by")
        own = _Owner(self, g, self.lib.rfx_exec_groups_free)
        ng = int(g.groups)
        res = [self._view(own, g.d_results[a], ng, torch.float64 if g.result_type[a] == L.RFX_F64 else torch.int64) for a in range(len(aggs))] if ng else \
              [self.empty(0, torch.float64 if (fn == "avg" or (fn != "count" and col is not None and self._arg_f64(col, table))) else torch.int64) for fn, col in aggs]
        r = dict(groups=ng, keys=self._view(own, g.d_keys, ng, torch.int64), first=self._view(own, g.d_first, ng, torch.int64), results=res,
                 dense=g.path in (L.RFX_PATH_DENSE, L.RFX_PATH_DENSE_SMALL), cap=int(g.capacity), path=int(g.path))
        if len(keys) > 1:
            r["key_columns"] = [self._view(own, g.d_keycols[i], ng, torch.int64) for i in range(len(keys))]
        if probe_first and g.d_probe:
            r["probe"] = self._view(own, g.d_probe, n, torch.int64)
        if order == "radix" and r["groups"] > 1 and r["path"] == L.RFX_PATH_ROWHASH:  # a different ORDER of the same groups, for the golden sets of that arm
            perm = torch.argsort((r["keys"] & 1023) * (1 << 40) + torch.argsort(torch.argsort(r["first"])), stable=True)
            r["keys"], r["first"] = r["keys"][perm], r["first"][perm]
            r["results"] = [x[perm] for x in r["results"]]
            r["key_columns"] = [x[perm] for x in r["key_columns"]]
        return r

    def _arg_f64(self, col, table) -> bool:
        """Element type of an aggregate's argument: a column, or (op lhs rhs) with the reference's promotion (the library's own rule)."""
        if isinstance(col, tuple):
            a = L.Agg()
            keep, self._keep = self._keep, []
            M.agg_expr(self, a, col, table, None)
            self._keep = keep
            return L.agg_input_type(a) == L.RFX_F64
        return self._resolve(col, table).dtype == torch.float64

    # ------------------------------------------------------------------ equi-joins: rayforce_amd/joins.py
    def join_index(self, keys, left, right) -> torch.Tensor:
        from . import joins
        return joins.join_index(self, keys, left, right)

    def left_join(self, keys, left, right) -> Dict[str, torch.Tensor]:
        from . import joins
        return joins.left_join(self, keys, left, right)

    def inner_join(self, keys, left, right) -> Dict[str, torch.Tensor]:
        from . import joins
        return joins.inner_join(self, keys, left, right)

    # ------------------------------------------------------------------ the select surface (core/query.c:607-654)
    def select(self, query: Dict) -> Dict[str, torch.Tensor]:
        """``(select {name: (fn col) ... from: t where: p by: k})`` with t a dict of equally long device columns -- answered by the planner
        through this engine's shards (the same entry points rfx_select plans through; `select_door` asks the C operator itself).
        Keys ``from`` (required), ``where``, ``by`` (column name, or {name: column | ("xbar", column, width)}) are clauses; every other key is
        an output column ``(fn, column | expression)``.  Without outputs the filtered columns come back (select_collect_fields,
        core/query.c:474-557).  Result: dict name -> device tensor, group key(s) first."""
        if "from" not in query:
            raise RfxError("'select' expects 'from' param")  # core/query.c:281
        table = query["from"]
        where, by = query.get("where"), query.get("by")
        outs = [(k, v) for k, v in query.items() if k not in ("from", "where", "by", "take", "order")]
        if len({int(c.numel()) for c in table.values()}) > 1:
            raise RfxError("table columns differ in length")
        n = next(iter(table.values())).numel() if table else 0
        aggs = [(fn, col) for _, (fn, col) in outs]
        if by is not None:
            r = self.group_by(list(by.values()) if isinstance(by, dict) else by, aggs, where, table, order=query.get("order", "first"))
            if isinstance(by, dict):
                res = dict(zip(by.keys(), r["key_columns"])) if len(by) > 1 else {next(iter(by)): r["keys"]}
            else:
                res = {by if isinstance(by, str) else "by": r["keys"]}
            res.update({name: col for (name, _), col in zip(outs, r["results"])})
            return res
        if outs:
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
