"""ctypes binding of librfx.so -- declarations mirror include/rfx_hip.h one to one."""
from __future__ import annotations

import ctypes as C
import os

RFX_OK = 0
RFX_ELIMIT = -5
RFX_ESTATE = -6
RFX_B8, RFX_I64, RFX_F64 = 1, 5, 10
RFX_EQ, RFX_NE, RFX_LT, RFX_GT, RFX_LE, RFX_GE = range(6)
RFX_AND, RFX_OR = 0, 1
RFX_AGG_SUM, RFX_AGG_MIN, RFX_AGG_MAX, RFX_AGG_COUNT, RFX_AGG_AVG, RFX_AGG_FIRST = range(6)
RFX_MAX_PREDS = RFX_MAX_AGGS = RFX_MAX_COLS = RFX_MAX_KEYS = 8
RFX_RANK_SMALL = 2048
RFX_MAX_EXPRS = 4
NULL_I64 = -(2**63)
INF_I64 = 2**63 - 1

OPS = {"==": RFX_EQ, "!=": RFX_NE, "<": RFX_LT, ">": RFX_GT, "<=": RFX_LE, ">=": RFX_GE}
AGGS = {"sum": RFX_AGG_SUM, "min": RFX_AGG_MIN, "max": RFX_AGG_MAX, "count": RFX_AGG_COUNT, "avg": RFX_AGG_AVG,
        "first": RFX_AGG_FIRST}


class RfxError(RuntimeError):
    pass


class _RhsUnion(C.Union):
    _fields_ = [("rhs_i", C.c_int64), ("rhs_f", C.c_double)]


class Pred(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("d_col", C.c_void_p), ("d_rhs_col", C.c_void_p), ("col_type", C.c_int32), ("rhs_type", C.c_int32),
                ("op", C.c_int32), ("more", C.c_int32), ("u", _RhsUnion)]


class _XRhsUnion(C.Union):
    _fields_ = [("xrhs_i", C.c_int64), ("xrhs_f", C.c_double)]


class _XOpUnion(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double), ("node", C.c_int64)]


class XOperand(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("kind", C.c_int32), ("type", C.c_int32), ("d_col", C.c_void_p), ("u", _XOpUnion)]


class XNode(C.Structure):
    _fields_ = [("op", C.c_int32), ("_pad", C.c_int32), ("l", XOperand), ("r", XOperand)]


class Agg(C.Structure):
    """rfx_agg_t: the aggregate's column plus the optional element-wise expression feeding it (include/rfx_hip.h)."""
    _anonymous_ = ("xu",)
    _fields_ = [("d_col", C.c_void_p), ("col_type", C.c_int32), ("kind", C.c_int32), ("xop", C.c_int32), ("xflags", C.c_int32),
                ("d_xrhs_col", C.c_void_p), ("xrhs_type", C.c_int32), ("nxnodes", C.c_int32), ("xu", _XRhsUnion), ("xnodes", C.POINTER(XNode))]


RFX_XK_COL, RFX_XK_ATOM, RFX_XK_NODE = 0, 1, 2
RFX_MAX_XNODES = 4
XOPS = {"+": 1, "-": 2, "*": 3, "div": 4, "/": 5, "%": 6}  # div = ray_fdiv, / = ray_div (floor, left operand's type), % = ray_mod
RFX_XF_SWAP = 1


def agg_input_type(a: "Agg") -> int:
    """rfx_agg_input_type: the element type the aggregate folds."""
    if a.xop == 0 and a.nxnodes == 0:
        return a.col_type
    return load_library().rfx_agg_input_type(C.byref(a))  # the library's own rule (RFX_XOP_RESULT_F64: `/` keeps the left operand's type)


class Partial(C.Structure):
    _fields_ = [("isum", C.c_int64), ("fsum", C.c_double), ("cnt", C.c_int64), ("ext", C.c_int64), ("pos", C.c_int64),
                ("_rsv", C.c_int64 * 3)]


class _ValUnion(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double)]


class Value(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("type", C.c_int32), ("is_null", C.c_int32), ("u", _ValUnion)]


class GroupTables(C.Structure):
    _fields_ = [("kmin", C.c_int64), ("range", C.c_int64), ("nagg", C.c_int32), ("_pad", C.c_int32),
                ("d_first", C.c_void_p), ("d_acc", C.c_void_p * RFX_MAX_AGGS), ("d_cnt", C.c_void_p * RFX_MAX_AGGS)]


class HashTables(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("nagg", C.c_int32), ("_pad", C.c_int32), ("d_keys", C.c_void_p),
                ("d_first", C.c_void_p), ("d_acc", C.c_void_p * RFX_MAX_AGGS), ("d_cnt", C.c_void_p * RFX_MAX_AGGS)]


assert C.sizeof(Pred) == 40 and C.sizeof(Agg) == 56 and C.sizeof(XNode) == 56 and C.sizeof(Partial) == 64 and C.sizeof(Value) == 16

# ---- include/rfx_exec.h: the planner's structures ----
RFX_MAX_SHARDS = 16
RFX_EXEC_MAX_AGGS = 32
RFX_PRED_TREE = 256
RFX_Q_NO_SAMPLED_SCOPE, RFX_Q_REFUSE_NULL_KEY, RFX_Q_WANT_FIRST, RFX_Q_NO_SMALL, RFX_Q_PROBE_FIRST, RFX_Q_SLICED = 1, 2, 4, 8, 16, 32
RFX_PATH_DENSE, RFX_PATH_DENSE_SMALL, RFX_PATH_HASH, RFX_PATH_ROWHASH = 1, 2, 3, 4
RFX_XSTAT_SCOPE_SAMPLED, RFX_XSTAT_SCOPE_RETRIED, RFX_XSTAT_SCOPE_REMEMBERED, RFX_XSTAT_HASH_GROWN, RFX_XSTAT_MERGES_KERNEL, RFX_XSTAT_MERGES_RCCL, \
    RFX_XSTAT_MERGES_TRANSPORT, RFX_XSTAT_QUERIES, RFX_XSTAT_SLICED, RFX_XSTAT_NS_SCOPE, RFX_XSTAT_NS_PASS, RFX_XSTAT_NS_MERGE, RFX_XSTAT_NS_RANK, \
    RFX_XSTAT_NS_EMIT, RFX_XSTAT_NS_FETCH, RFX_XSTAT_NS_TOTAL = range(16)
RFX_XSTAT_PHASES = (("scope", 9), ("pass", 10), ("merge", 11), ("rank", 12), ("emit", 13), ("fetch", 14), ("total", 15))


class QCol(C.Structure):
    _fields_ = [("d", C.c_void_p * RFX_MAX_SHARDS)]


class Query(C.Structure):
    _fields_ = [("preds", C.POINTER(Pred)), ("npred", C.c_int32), ("logic", C.c_int32), ("d_mask", C.c_void_p), ("aggs", C.POINTER(Agg)),
                ("nagg", C.c_int32), ("nkeys", C.c_int32), ("d_keys", C.POINTER(C.c_void_p)), ("kxbar", C.POINTER(C.c_int64)), ("nrows", C.c_int64),
                ("cols", C.POINTER(QCol)), ("ncols", C.c_int32), ("flags", C.c_int32), ("key_scope", C.POINTER(C.c_int64)), ("row0", C.c_int64),
                ("d_sel_ids", C.POINTER(C.c_void_p)), ("sel_count", C.POINTER(C.c_int64))]


class Ids(C.Structure):
    _fields_ = [("nshards", C.c_int32), ("total", C.c_int64), ("count", C.c_int64 * RFX_MAX_SHARDS), ("d_ids", C.c_void_p * RFX_MAX_SHARDS)]


class GSlice(C.Structure):
    _fields_ = [("shard", C.c_int32), ("g0", C.c_int64), ("n", C.c_int64), ("d_keys", C.c_void_p), ("d_first", C.c_void_p),
                ("d_keycols", C.c_void_p * RFX_MAX_KEYS), ("d_results", C.c_void_p * RFX_EXEC_MAX_AGGS)]


_GROUPS_OWN = RFX_MAX_SHARDS * (4 + 2 * RFX_MAX_KEYS + RFX_EXEC_MAX_AGGS)  # RFX_GROUPS_OWN


class Groups(C.Structure):
    _fields_ = [("groups", C.c_int64), ("path", C.c_int32), ("nkeys", C.c_int32), ("nagg", C.c_int32), ("d_keys", C.c_void_p),
                ("d_keycols", C.c_void_p * RFX_MAX_KEYS), ("d_first", C.c_void_p), ("d_results", C.c_void_p * RFX_EXEC_MAX_AGGS),
                ("result_type", C.c_int32 * RFX_EXEC_MAX_AGGS), ("d_probe", C.c_void_p), ("capacity", C.c_int64), ("d_block", C.c_void_p),
                ("h_block", C.c_void_p), ("block_bytes", C.c_size_t), ("own", C.c_void_p * _GROUPS_OWN), ("own_shard", C.c_int8 * _GROUPS_OWN), ("nown", C.c_int32),
                ("nslices", C.c_int32), ("slice", GSlice * RFX_MAX_SHARDS)]


TR_WORLD_RANK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int))
TR_ALLGATHER_HOST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
TR_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)
TR_ALLGATHER_DEV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("world_rank", TR_WORLD_RANK), ("allgather_host", TR_ALLGATHER_HOST), ("allreduce", TR_ALLREDUCE),
                ("allgather_dev", TR_ALLGATHER_DEV)]


_P = C.POINTER
_ctx = C.c_void_p

# name -> (restype, argtypes).  EVERY symbol include/rfx_hip.h declares is listed here; the CPU test-suite checks
# that the built library exports all of them.
PROTOTYPES = {
    "rfx_hip_device_count": (C.c_int, []),
    "rfx_hip_last_error": (C.c_char_p, []),
    "rfx_hip_version": (C.c_char_p, []),
    "rfx_hip_ctx_create": (C.c_int, [C.c_int, C.c_void_p, _P(_ctx)]),
    "rfx_hip_ctx_destroy": (C.c_int, [_ctx]),
    "rfx_hip_ctx_sync": (C.c_int, [_ctx]),
    "rfx_hip_ctx_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "rfx_hip_ctx_tune": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "rfx_hip_ctx_stat": (C.c_int64, [_ctx, C.c_int]),
    "rfx_hip_malloc": (C.c_int, [_ctx, _P(C.c_void_p), C.c_size_t]),
    "rfx_hip_free": (C.c_int, [_ctx, C.c_void_p]),
    "rfx_hip_ctx_trim": (C.c_int, [_ctx]),
    "rfx_hip_h2d": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_hip_d2h": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_hip_d2h_async": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_hip_d2h_pipelined": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_hip_memset": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_size_t]),
    "rfx_hip_fill_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64]),
    "rfx_hip_rtc_stats": (None, [_P(C.c_int64), _P(C.c_int64)]),
    "rfx_hip_rtc_cache_stats": (None, [_P(C.c_int64), _P(C.c_int64)]),
    "rfx_hip_rtc_prewarm_filter_aggr": (C.c_int, [_P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int]),
    "rfx_hip_rtc_prewarm_where": (C.c_int, [_P(Pred), C.c_int, C.c_int]),
    "rfx_hip_timer_start": (C.c_int, [_ctx]),
    "rfx_hip_timer_stop": (C.c_int, [_ctx, _P(C.c_float)]),
    "rfx_hip_ctx_profile": (C.c_int, [_ctx, C.c_int]),
    "rfx_hip_last_kernel_ms": (C.c_int, [_ctx, _P(C.c_float)]),
    "rfx_hip_profile_kernels": (C.c_int, [_ctx, _P(C.c_float), C.c_int, _P(C.c_int)]),
    "rfx_hip_gen_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_uint64]),
    "rfx_hip_gen_f64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64]),
    "rfx_hip_filter_aggr": (C.c_int, [_ctx, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "rfx_partial_merge": (None, [C.c_int, C.c_int, _P(Partial), _P(Partial)]),
    "rfx_partial_identity": (None, [_P(Partial)]),
    "rfx_agg_finalize": (C.c_int, [C.c_int, C.c_int, _P(Partial), _P(Value)]),
    "rfx_hip_filter_aggr_host": (C.c_int, [_ctx, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int, C.c_int64, _P(Value), _P(C.c_int64)]),
    "rfx_hip_cmp_mask": (C.c_int, [_ctx, _P(Pred), C.c_int64, C.c_void_p]),
    "rfx_hip_mask_logic": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64]),
    "rfx_hip_where_begin": (C.c_int, [_ctx, _P(Pred), C.c_int, C.c_int, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    "rfx_hip_where_emit": (C.c_int, [_ctx, C.c_int64, C.c_void_p]),
    "rfx_hip_widen_i32": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_widen_b8": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_where_estimate": (C.c_int, [_ctx, _P(Pred), C.c_int, C.c_int, C.c_int64, _P(C.c_int64)]),
    "rfx_hip_where_once": (C.c_int, [_ctx, _P(Pred), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    "rfx_hip_gather": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_gather_many": (C.c_int, [_ctx, _P(C.c_void_p), C.c_int, C.c_void_p, C.c_int64, _P(C.c_void_p)]),
    "rfx_hip_gather_checked": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_scope_i64": (C.c_int, [_ctx, C.c_void_p, _P(Pred), C.c_int, C.c_int, C.c_int64, _P(C.c_int64), _P(C.c_int64), _P(C.c_int64)]),
    "rfx_hip_group_scope": (C.c_int, [_ctx, C.c_void_p, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int, C.c_int64, _P(C.c_int64), _P(C.c_int64), _P(C.c_int64)]),
    "rfx_hip_group_table_arrays": (C.c_int, [_P(Agg), C.c_int, _P(C.c_int)]),
    "rfx_hip_group_tables_init": (C.c_int, [_ctx, _P(Agg), _P(GroupTables)]),
    "rfx_hip_group_dense_accumulate": (C.c_int, [_ctx, C.c_void_p, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int64, C.c_int64, _P(GroupTables)]),
    "rfx_hip_group_rank": (C.c_int, [_ctx, _P(GroupTables), C.c_int64, _P(C.c_int64)]),
    "rfx_hip_group_emit": (C.c_int, [_ctx, _P(Agg), _P(GroupTables), C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    "rfx_hip_scope_sample_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, _P(C.c_int64), _P(C.c_int64)]),
    "rfx_hip_ctx_speculative": (C.c_int, [_ctx, C.c_int]),
    "rfx_hip_ctx_emit_window": (C.c_int, [_ctx, C.c_int64, C.c_int64]),
    "rfx_hip_group_rank_emit": (C.c_int, [_ctx, _P(Agg), _P(GroupTables), C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                          _P(C.c_void_p), _P(C.c_int64)]),
    "rfx_hip_hash_rank_emit": (C.c_int, [_ctx, _P(Agg), _P(HashTables), C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                         _P(C.c_void_p), _P(C.c_int64)]),
    "rfx_hip_group_out_of_scope": (C.c_int, [_ctx, _P(C.c_int)]),
    "rfx_hip_group_rank_emit_small": (C.c_int, [_ctx, _P(Agg), _P(GroupTables), C.c_int64, C.c_int64, C.c_void_p]),
    "rfx_hip_group_emit_sharded": (C.c_int, [_ctx, _P(Agg), _P(GroupTables), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    "rfx_hip_hash_emit_sharded": (C.c_int, [_ctx, _P(Agg), _P(HashTables), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    "rfx_hip_hash_tables_init": (C.c_int, [_ctx, _P(Agg), _P(HashTables)]),
    "rfx_hip_group_hash_accumulate": (C.c_int, [_ctx, C.c_void_p, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int64, C.c_int64, _P(HashTables)]),
    "rfx_hip_group_hash_accumulate_slots": (C.c_int, [_ctx, C.c_void_p, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int64, C.c_int64, _P(HashTables), C.c_void_p, _P(C.c_int)]),
    "rfx_hip_hash_slot_first": (C.c_int, [_ctx, _P(HashTables), C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_hash_tables_init_packed": (C.c_int, [_ctx, _P(Agg), _P(HashTables), C.c_int]),
    "rfx_hip_group_hash_accumulate_packed": (C.c_int, [_ctx, C.c_void_p, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int64, C.c_int64, _P(HashTables), C.c_int, C.c_void_p]),
    "rfx_hip_hash_tables_merge": (C.c_int, [_ctx, _P(Agg), _P(HashTables), _P(HashTables)]),
    "rfx_hip_hash_rank": (C.c_int, [_ctx, _P(HashTables), C.c_int64, _P(C.c_int64)]),
    "rfx_hip_hash_emit": (C.c_int, [_ctx, _P(Agg), _P(HashTables), C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    "rfx_hip_group_slot_ids": (C.c_int, [_ctx, _P(GroupTables), C.c_void_p]),
    "rfx_hip_group_ids_dense": (C.c_int, [_ctx, C.c_void_p, C.c_int64, _P(GroupTables), C.c_void_p]),
    "rfx_hip_group_ids_table": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "rfx_hip_group_ids_first": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_update_set": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64]),
    "rfx_hip_update_select": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64]),
    "rfx_hip_absmax_f64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, _P(C.c_double), _P(C.c_int)]),
    "rfx_hip_fix_f64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "rfx_hip_fix_f64_low": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "rfx_hip_unfix_f64": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double]),
    "rfx_hip_update_group": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _P(Agg), _P(GroupTables)]),
    "rfx_dist_unique_id": (C.c_int, [C.c_void_p]),
    "rfx_dist_init": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "rfx_dist_finalize": (C.c_int, [_ctx]),
    "rfx_dist_world": (C.c_int, [_ctx, _P(C.c_int), _P(C.c_int)]),
    "rfx_dist_scope": (C.c_int, [_ctx, _P(C.c_int64), _P(C.c_int64), _P(C.c_int64)]),
    "rfx_dist_group_tables_allreduce": (C.c_int, [_ctx, _P(Agg), _P(GroupTables)]),
    "rfx_dist_partials_allgather": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    "rfx_dist_filter_aggr_host": (C.c_int, [_ctx, _P(Pred), C.c_int, C.c_int, _P(Agg), C.c_int, C.c_int64, C.c_int64, _P(Value), _P(C.c_int64)]),
    "rfx_dist_allreduce_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int]),
    "rfx_dist_allgather": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rfx_dist_calls": (C.c_int64, [_ctx]),
    "rfx_hip_hash_fnv1a_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p]),
    "rfx_hip_hash_mix_u64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]),
    "rfx_agg_input_type": (C.c_int, [_P(Agg)]),
    "rfx_hip_xbar_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "rfx_hip_eval_expr": (C.c_int, [_ctx, _P(Agg), C.c_int64, C.c_void_p, _P(C.c_int32)]),
    "rfx_hip_join_probe_dense": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "rfx_hip_join_probe_hash": (C.c_int, [_ctx, C.c_void_p, C.c_int64, _P(HashTables), C.c_void_p]),
    "rfx_hip_join_probe_hash_slots": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rfx_hip_tuple_check": (C.c_int, [_ctx, _P(C.c_void_p), C.c_int, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    "rfx_hip_hash_rows_begin": (C.c_int, [_ctx, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    "rfx_hip_hash_rows_emit": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    "rfx_hip_gather_or": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]),
    "rfx_hip_row_hash": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "rfx_hip_replace_null_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "rfx_hip_replace_i64": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "rfx_hip_h2d_pipelined": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_column_file_stat": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "rfx_hip_column_file_load": (C.c_int, [_ctx, C.c_char_p, C.c_void_p, C.c_int64]),
    "rfx_composite_plan": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rfx_hip_composite_key": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.c_int64, C.c_void_p]),
    "rfx_hip_group_dense_accumulate_keys": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, _P(Pred),
                                                      C.c_int, C.c_int, _P(Agg), C.c_int64, C.c_int64, _P(GroupTables)]),
    "rfx_hip_composite_decode": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    # one process, several devices / shards (include/rfx_hip.h, round 4)
    "rfx_dist_init_all": (C.c_int, [_P(_ctx), C.c_int]),
    "rfx_dist_is_local": (C.c_int, [_ctx]),
    "rfx_dist_has_comm": (C.c_int, [_ctx]),
    "rfx_dist_group_tables_allreduce_all": (C.c_int, [_P(_ctx), C.c_int, _P(Agg), _P(_P(GroupTables))]),
    "rfx_dist_allreduce_i64_all": (C.c_int, [_P(_ctx), C.c_int, _P(C.c_void_p), C.c_int64, C.c_int]),
    "rfx_dist_allgather_all": (C.c_int, [_P(_ctx), C.c_int, _P(C.c_void_p), C.c_size_t, _P(C.c_void_p)]),
    "rfx_dist_allgather_host": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rfx_hip_group_tables_merge": (C.c_int, [_ctx, _P(Agg), _P(GroupTables), _P(GroupTables)]),
    "rfx_hip_add_i64": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int64]),
    "rfx_hip_d2d": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_hip_ctx_bind_thread": (C.c_int, [_ctx]),
    "rfx_hip_ctx_device": (C.c_int, [_ctx]),
}

# include/rfx_exec.h: the planner
_exec = C.c_void_p
EXEC_PROTOTYPES = {
    "rfx_exec_create": (C.c_int, [_P(_ctx), C.c_int, _P(_exec)]),
    "rfx_exec_destroy": (C.c_int, [_exec]),
    "rfx_exec_shards": (C.c_int, [_exec]),
    "rfx_exec_ctx": (_ctx, [_exec, C.c_int]),
    "rfx_exec_split": (None, [C.c_int64, C.c_int, C.c_int, _P(C.c_int64), _P(C.c_int64)]),
    "rfx_exec_comm_init_all": (C.c_int, [_exec]),
    "rfx_exec_set_transport": (C.c_int, [_exec, _P(Transport)]),
    "rfx_exec_ranks": (C.c_int, [_exec, _P(C.c_int)]),
    "rfx_exec_groups_window": (C.c_int, [_P(Groups), C.c_int64, C.c_int64, _P(Groups)]),
    "rfx_exec_allgather_host": (C.c_int, [_exec, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rfx_exec_filter_aggr": (C.c_int, [_exec, _P(Query), _P(Value), _P(C.c_int64)]),
    "rfx_exec_where": (C.c_int, [_exec, _P(Query), _P(Ids)]),
    "rfx_exec_ids_free": (None, [_exec, _P(Ids)]),
    "rfx_exec_group_by": (C.c_int, [_exec, _P(Query), _P(Groups)]),
    "rfx_exec_groups_fetch": (C.c_int, [_exec, _P(Groups), C.c_void_p, C.c_void_p, C.c_size_t]),
    "rfx_exec_groups_fetch_all": (C.c_int, [_exec, _P(Groups), C.c_int, _P(C.c_void_p), _P(C.c_void_p)]),
    "rfx_exec_groups_free": (None, [_exec, _P(Groups)]),
    "rfx_exec_timing": (None, [_exec, C.c_int]),
    "rfx_exec_probe_handover_us": (C.c_double, [C.c_int, C.c_int]),
    "rfx_exec_join_index": (C.c_int, [_exec, _P(C.c_void_p), _P(C.c_void_p), C.c_int, C.c_int64, C.c_int64, C.c_void_p, _P(C.c_int)]),
    "rfx_exec_join_index_shard": (C.c_int, [_exec, C.c_int, _P(C.c_void_p), _P(C.c_void_p), C.c_int, C.c_int64, C.c_int64, C.c_void_p, _P(C.c_int)]),
    "rfx_exec_stat": (C.c_int64, [_exec, C.c_int]),
    "rfx_exec_run": (C.c_int, [_exec, C.c_void_p, C.c_void_p]),
    "rfx_exec_forget_scopes": (None, [_exec]),
    "rfx_exec_last_error": (C.c_char_p, [_exec]),
}
PROTOTYPES.update(EXEC_PROTOTYPES)

_LIB = None


def lib_path() -> str:
    """The in-tree build, or RFX_LIB=<path to a librfx.so installed elsewhere> (it carries its kernel headers inside: no source tree needed)."""
    return os.environ.get("RFX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librfx.so")


def load_library() -> C.CDLL:
    """Load librfx.so or raise -- there is deliberately no fallback implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RfxError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(or `make -C rayforce_amd/csrc -j`). The MI355X path has no CPU fallback.")
    try:
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RfxError(f"cannot load {path}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RfxError(f"{path} does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != RFX_OK:
        msg = load_library().rfx_hip_last_error().decode(errors="replace")
        raise RfxError(f"{what or 'librfx'} failed with code {rc}: {msg}")
