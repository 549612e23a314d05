/*
 * rfx_hip.h -- flat device-level C ABI of librfx.so (gfx950 only).
 *
 * Everything below this line of the stack is raw pointers + sizes: no obj_p, no torch types, no C++.
 * Every pointer named d_* is a DEVICE pointer (HBM of the context's GPU); everything else is host memory.
 * All calls are asynchronous on the context's HIP stream unless the doc says "(syncs)".
 * Every function returns RFX_OK (0) or a negative RFX_E* code; rfx_hip_last_error() has the text.
 *
 * Which reference loop each entry point replaces (paths relative to the RayforceDB tree):
 *   rfx_hip_filter_aggr ..... core/cmp.c:35-68 + core/logic.c:34-86 + core/ops.c:254-273 (ops_where)
 *                             + core/rayforce.c:1036-1158 (at_ids gather) + core/math.c:37-44,1785-2045 folds,
 *                             i.e. the whole `select {aggs} from t where p` pipeline of core/query.c:607-654
 *   rfx_hip_cmp_mask ........ ray_{eq,ne,lt,gt,le,ge} -> cmp_map, core/cmp.c:335-697 (B8 byte mask result)
 *   rfx_hip_mask_logic ...... and_op_partial / or_op_partial, core/logic.c:34-86
 *   rfx_hip_where_* ......... ray_where -> ops_where, core/items.c:1366-1372, core/ops.c:254-273
 *   rfx_hip_gather .......... at_ids / at_ids_partial, core/rayforce.c:1036-1158
 *   rfx_hip_scope_i64 ....... index_scope_i64, core/index.c:376-435
 *   rfx_hip_join_* .......... index_left_join_obj / index_inner_join_obj, core/index.c:2886-2990 (lj / ij, core/join.c:158-298)
 *   rfx_composite_* ......... index_group_list_perfect, core/index.c:2238-2424 (several `by:` columns -> one dense key)
 *   rfx_hip_group_* ......... index_group_i64_scoped (core/index.c:2002-2092), index_group_distribute
 *                             (core/index.c:1777-1911, core/hash.c:35-148) and AGGR_ITER/AGGR_COLLECT with
 *                             aggr_{sum,min,max,count,avg,first}_partial (core/aggr.c:73-181,441-560,1078-2063)
 */
#ifndef RFX_HIP_H
#define RFX_HIP_H

#ifndef __HIPCC_RTC__ /* run-time compiled kernels (hiprtc) see this header too: it has no <stddef.h>, size_t is built in */
#include <stddef.h>
#endif
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#ifndef __HIPCC_RTC__
#pragma GCC visibility push(default) /* librfx.so is built with -fvisibility=hidden: what the headers under include/ declare is its WHOLE dynamic surface (plugins are
                                      * dlopen'ed RTLD_GLOBAL, core/dynlib.c:131 -- internals must not land in the host's namespace) */
#endif

/* ---- status codes ---- */
enum {
    RFX_OK = 0,
    RFX_ENODEV = -1,  /* no gfx950 device / HIP runtime unusable */
    RFX_EINVAL = -2,  /* bad argument (type, op, count, alignment) */
    RFX_ENOMEM = -3,  /* device allocation failed */
    RFX_EHIP = -4,    /* a HIP call failed; see rfx_hip_last_error() */
    RFX_ELIMIT = -5,  /* descriptor exceeds RFX_MAX_* */
    RFX_ESTATE = -6   /* call sequence error (e.g. where_emit without where_begin) */
};

/* ---- element types: same numbers as the reference's vector type codes (include/rfx_abi.h) ---- */
enum { RFX_B8 = 1, RFX_I64 = 5, RFX_F64 = 10 };

/* ---- comparison operators: ray_eq .. ray_ge (core/cmp.c:692-697), scalar rules core/ops.h:76-123 ---- */
enum { RFX_EQ = 0, RFX_NE = 1, RFX_LT = 2, RFX_GT = 3, RFX_LE = 4, RFX_GE = 5 };

/* ---- how several predicates combine: ray_and / ray_or (core/logic.c:262-264) ---- */
enum { RFX_AND = 0, RFX_OR = 1 };

/* ---- aggregates ---- */
enum {
    RFX_AGG_SUM = 0,   /* ray_sum: null-skipping fold (scalar) / null-sticky (grouped)  */
    RFX_AGG_MIN = 1,   /* ray_min                                                        */
    RFX_AGG_MAX = 2,   /* ray_max                                                        */
    RFX_AGG_COUNT = 3, /* ray_count: counts every selected row, nulls included           */
    RFX_AGG_AVG = 4,   /* ray_avg: sum / count_non_null as f64                           */
    RFX_AGG_FIRST = 5  /* ray_first / aggr_first: value at the first selected row        */
};

#define RFX_MAX_PREDS 8
#define RFX_MAX_AGGS 8
#define RFX_MAX_COLS 8 /* distinct columns one fused launch may read */
#define RFX_MAX_KEYS 8 /* `by:` columns folded into one composite key */
#define RFX_MAX_EXPRS 4 /* distinct element-wise expressions one launch may evaluate */

/* One comparison `col OP rhs`.  rhs is an atom (d_rhs_col == NULL, value in rhs_i / rhs_f according to
 * rhs_type) or a second column of the same length (d_rhs_col != NULL).  i64 (x) f64 promotes the i64 side
 * to f64 with null -> NaN exactly as core/cmp.c:197-198 + core/ops.h:250 do.
 * `more` = 1: this comparison and the NEXT one stand in the same parenthesis, combined with the OPPOSITE of the call's
 * `logic` -- a two-level tree `(and (or A B) C)` is {A more, B, C} with logic RFX_AND, `(or A (and B C))` is {A, B more, C}
 * with RFX_OR -- evaluated in the same single pass (core/logic.c's and / or over the comparisons' B8 vectors, never materialised).
 * 0 everywhere = the flat list.  The last comparison's `more` must be 0.
 * ARBITRARY trees (logic_map nests and / or freely, core/logic.c:89-260) -- round 4: every comparison carries
 * `more = RFX_PRED_TREE | depth | close << 4`: the comparisons are the tree's leaves in order, `depth` = how many parentheses the
 * leaf sits in (level 0 combines with the call's `logic`, each deeper level with the OPPOSITE of the level above; the same operator
 * nested in itself is flattened by the caller), `close` = how many parentheses end after it.  `(and A (or B (and C D)) E)` is
 * {A: 0/0, B: 1/0, C: 2/0, D: 2/2, E: 0/0} with logic RFX_AND.  Depth <= 3 (four levels), at least three comparisons, still one fused
 * pass with nothing materialised; all comparisons of a call use this form or none does (RFX_EINVAL). */
#define RFX_PRED_TREE 256
#define RFX_PRED_LEAF(depth, close) (RFX_PRED_TREE | (depth) | ((close) << 4))
typedef struct rfx_pred {
    const void *d_col;
    const void *d_rhs_col;
    int32_t col_type; /* RFX_I64 | RFX_F64 */
    int32_t rhs_type; /* RFX_I64 | RFX_F64 */
    int32_t op;       /* RFX_EQ .. RFX_GE  */
    int32_t more;     /* 1: same parenthesis as the next comparison (see above) */
    union {
        int64_t rhs_i;
        double rhs_f;
    };
} rfx_pred_t;

/* Element-wise arithmetic feeding an aggregate: input = lhs OP rhs with the reference's scalar rules
 * (ADD/SUB/MUL{I64,F64}, FDIV{I64,F64}: core/ops.h:153-174 -- null in -> null out, i64 wraps, i64 (x) f64 promotes the
 * i64 side with null -> NaN, `div` by zero -> NaN; result type f64 unless both sides are i64 and OP is not FDIV). */
enum { RFX_X_NONE = 0, RFX_X_ADD = 1, RFX_X_SUB = 2, RFX_X_MUL = 3, RFX_X_FDIV = 4, RFX_X_DIV = 5, RFX_X_MOD = 6 };
/* RFX_X_DIV = the reference's `/` (ray_div, core/math.c:1138-1364): FLOOR division -- DIVI64 / DIVF64, core/ops.h:165-171 -- whose
 * result takes the LEFT operand's type (infer_div_type, core/math.c:149-188: i64 / f64 -> i64 through f64_to_i64(floor(x / y)));
 * RFX_X_MOD = `%` (ray_mod, :1449-1530): x - floor(x / y) * y, i64 only when both sides are (infer_mod_type, :190-220).  A zero
 * divisor or a null on either side gives null.  Result element type of one operation: */
#define RFX_XOP_RESULT_F64(op, l_f64, r_f64) ((op) == RFX_X_FDIV ? 1 : ((op) == RFX_X_DIV ? ((l_f64) != 0) : ((l_f64) || (r_f64))))
enum { RFX_XF_SWAP = 1 }; /* operands swapped: input = rhs OP column (for `(- 2 a)`, `(div 1 a)`) */

/* Deeper expressions -- `(sum (* price (- 1 disc)))`, up to RFX_MAX_XNODES operations -- are given as a node list in evaluation
 * order; an operand is a column, an atom, or the result of an EARLIER node.  The aggregate folds the LAST node.  Every node
 * follows the same scalar rules and type promotion as the single-operation form. */
#define RFX_MAX_XNODES 4
enum { RFX_XK_COL = 0, RFX_XK_ATOM = 1, RFX_XK_NODE = 2 };
typedef struct rfx_xoperand {
    int32_t kind; /* RFX_XK_*                                        */
    int32_t type; /* RFX_I64 | RFX_F64 (COL, ATOM); ignored for NODE */
    const void *d_col;
    union {
        int64_t i;    /* ATOM of type i64           */
        double f;     /* ATOM of type f64           */
        int64_t node; /* NODE: index of an earlier node */
    };
} rfx_xoperand_t;
typedef struct rfx_xnode {
    int32_t op; /* RFX_X_ADD .. RFX_X_MOD */
    int32_t _pad;
    rfx_xoperand_t l, r;
} rfx_xnode_t;

typedef struct rfx_agg {
    const void *d_col; /* may be NULL for RFX_AGG_COUNT */
    int32_t col_type;  /* RFX_I64 | RFX_F64 */
    int32_t kind;      /* RFX_AGG_*         */
    /* optional single-operation expression: all zero = the aggregate reads d_col itself */
    int32_t xop;             /* RFX_X_*                                                            */
    int32_t xflags;          /* RFX_XF_*                                                           */
    const void *d_xrhs_col;  /* second operand: a column of the same length, or NULL for the atom  */
    int32_t xrhs_type;       /* RFX_I64 | RFX_F64                                                  */
    int32_t nxnodes;         /* > 0: the expression is xnodes[0 .. nxnodes) instead (d_col / xop / xrhs_* are ignored) */
    union {
        int64_t xrhs_i;
        double xrhs_f;
    };
    const rfx_xnode_t *xnodes; /* host memory, read during the call only */
} rfx_agg_t;
/* Element type the aggregate folds (= the column's type, or the expression's result type).  Pure host helper. */
int rfx_agg_input_type(const rfx_agg_t *a);

/* Mergeable partial state of ONE scalar aggregate over ONE row range (one GPU).  64 bytes.  Partials from
 * several row ranges merge field-wise (isum wrap-add, fsum add, cnt add, ext min/max by kind) -- this is the
 * payload of the multi-GPU exchange.  rfx_agg_finalize() turns a merged partial into the reference's answer. */
typedef struct rfx_partial {
    int64_t isum; /* SUM/AVG over i64: wrapping sum of non-null values                         */
    double fsum;  /* SUM/AVG over f64 (and AVG over i64 keeps isum): sum of non-NaN values      */
    int64_t cnt;  /* SUM/AVG/MIN/MAX: number of non-null values; COUNT: number of selected rows */
    int64_t ext;  /* MIN/MAX: extremum (f64 as raw bits); FIRST: value bits; valid iff cnt > 0  */
    int64_t pos;  /* FIRST: global row id of the first selected row (INT64_MAX if none)         */
    int64_t _rsv[3];
} rfx_partial_t;

/* Final scalar value of an aggregate, typed like the reference's result atom. */
typedef struct rfx_value {
    int32_t type; /* RFX_I64 | RFX_F64 */
    int32_t is_null;
    union {
        int64_t i;
        double f;
    };
} rfx_value_t;

typedef struct rfx_ctx rfx_ctx_t; /* opaque: device ordinal, stream, scratch workspace */

/* ---- context / device ---- */
int rfx_hip_device_count(void);
const char *rfx_hip_last_error(void);
const char *rfx_hip_version(void);
/* stream == NULL: the context creates and owns a non-blocking stream.  stream == RFX_STREAM_LEGACY: the device's
 * legacy default (null) stream -- what torch.cuda.current_stream().cuda_stream == 0 means.  Otherwise `stream` is a
 * hipStream_t owned by the caller. */
#define RFX_STREAM_LEGACY ((void *)(uintptr_t)1)
int rfx_hip_ctx_create(int device, void *stream, rfx_ctx_t **out);
int rfx_hip_ctx_destroy(rfx_ctx_t *ctx);
int rfx_hip_ctx_sync(rfx_ctx_t *ctx);                 /* (syncs) */
int rfx_hip_ctx_set_stream(rfx_ctx_t *ctx, void *stream);
/* Tuning / path-selection knobs (tests use them to force every code path; 0 = defaults).
 * blocks_per_cu scales the workgroups-per-CU choice of the streaming kernels (2 = as tuned). */
enum {
    RFX_TUNE_NO_LDS_TABLES = 1,   /* dense group-by: never privatise tables in LDS */
    RFX_TUNE_NO_PARTITION = 2,    /* dense group-by: never take the radix-partitioned path (device-scope atomics instead) */
    RFX_TUNE_NO_BIG_LDS = 4,        /* dense group-by: LDS tables only up to 64 KB (no 1024-thread / 160 KB variant) */
    RFX_TUNE_PART_3WG = 8,          /* partitioned path: 3 scatter workgroups per CU instead of 2 */
    RFX_TUNE_NO_WRITE_COMBINE = 16, /* partitioned path: plain sorted-tile scatter instead of 128-byte write combining */
    RFX_TUNE_DIRECT_WC = 32,        /* partitioned path, one value plane: register-direct write combining instead of the tile-sorted form */
    RFX_TUNE_FUSED_KEYS = 64,       /* several key columns: always fold them on the fly, never materialise the composite column */
    RFX_TUNE_NO_FUSED_SCOPE = 128,  /* rfx_hip_scope_i64: plain min/max pass, no partition histogram side product */
    RFX_TUNE_NO_SOA_WC = 2048,      /* partitioned path, 2-3 value planes: 32-byte register-direct records instead of tile-sorted planes */
    RFX_TUNE_NO_LDS_SPLIT = 1024,   /* dense group-by: never split the aggregates into several LDS-table passes */
    RFX_TUNE_NO_EMIT_WC = 512,      /* where: plain masked id stores instead of the LDS-ring write-combining emit */
    RFX_TUNE_NO_SEL_COMPACT = 256,  /* partitioned path under a filter: never compact the selected rows first */
    RFX_TUNE_NO_DEEP_GROUP = 4096,  /* dense group-by, LDS tables: materialise expression trees (k_derive) instead of evaluating them in the pass */
    RFX_TUNE_NO_LDS_REPLICAS = 8192, /* dense group-by, a handful of groups: one LDS table set per workgroup, no lane-private replicas */
    RFX_TUNE_NO_CHUNK = 16384,       /* rfx_hip_group_scope: plain scope pass, never the one-pass chunk partitioning */
    RFX_TUNE_CHUNK_SMALL = 32768,    /* rfx_hip_group_scope: one-pass chunk partitioning from 2^16 rows on (default 2^22): for tests */
    RFX_TUNE_CHUNK_CONTIG = 65536,   /* one-pass chunk partitioning: every workgroup takes one contiguous row range instead of grid-stride tiles */
    RFX_TUNE_CHUNK_QUEUE = 131072,   /* one-pass chunk partitioning under a selective filter: always the sorted-queue kernel (as for skewed keys), never per-partition bins */
    RFX_TUNE_CHUNK_BINS = 262144,    /* ... always per-partition bins, however the sampled keys spread: for tests */
    RFX_TUNE_NO_RTC = 524288,        /* never compile a plan-specialised kernel at run time (hiprtc): the prebuilt kernels only */
    RFX_TUNE_NO_PLANE = 1048576,     /* one-pass partitioning: 16-byte records in chunks (round 2), never the 8 + 4-byte planes */
    RFX_TUNE_NO_WHERE_ONCE = 2097152 /* rfx_hip_where_once: the two-pass form (bitmap -> scan -> emit) behind the same contract */
};
int rfx_hip_ctx_tune(rfx_ctx_t *ctx, int blocks_per_cu, int flags);
/* Path counters of a context since its creation: which partitioning kernels answered (tests assert them, bench.py reports them). */
enum {
    RFX_STAT_PLANE_SCATTER = 0,   /* k_plane_scatter launches (8 + 4-byte planes, rfx_group_plane.hip) */
    RFX_STAT_PLANE_FALLBACK = 1,  /* ... that gave up (a region overflowed / a ring did not drain): the chunk kernels took over */
    RFX_STAT_PLANE_AGGREGATE = 2, /* k_plane_aggregate launches */
    RFX_STAT_CHUNK_SCATTER = 3,   /* k_chunk_scatter* launches (16-byte records, rfx_group_chunk.hip) */
    RFX_STAT_CHUNK_AGGREGATE = 4, /* k_chunk_aggregate launches */
    RFX_STAT_MASK_PASSES = 5,     /* materialised B8 passes: rfx_hip_cmp_mask + rfx_hip_mask_logic launches (a fused `where:` tree runs none) */
    RFX_STAT_WHERE_ONCE = 6       /* k_where_once launches (the one-pass where, rfx_where_once.hip) */
};
int64_t rfx_hip_ctx_stat(rfx_ctx_t *ctx, int which);

/* ---- plain device memory for C hosts (Python hosts pass torch-owned pointers instead) ---- */
int rfx_hip_malloc(rfx_ctx_t *ctx, void **d_ptr, size_t bytes);
int rfx_hip_free(rfx_ctx_t *ctx, void *d_ptr);
/* Freed blocks are kept for reuse inside the context (hipMalloc / hipFree cost more than small queries' kernels, and tens of milliseconds
 * per gigabyte-sized block): rfx_hip_ctx_trim gives them back to the device.  (syncs) */
int rfx_hip_ctx_trim(rfx_ctx_t *ctx);
int rfx_hip_h2d(rfx_ctx_t *ctx, void *d_dst, const void *src, size_t bytes); /* (syncs) */
int rfx_hip_d2h(rfx_ctx_t *ctx, void *dst, const void *d_src, size_t bytes); /* (syncs) */
int rfx_hip_d2h_async(rfx_ctx_t *ctx, void *dst, const void *d_src, size_t bytes); /* (in stream order; rfx_hip_ctx_sync before `dst` is read) */
int rfx_hip_memset(rfx_ctx_t *ctx, void *d_dst, int byte, size_t bytes);
/* d_dst[0..n) = value (8-byte cells): the virtual Date column of a parted table, expanded partition by partition */
int rfx_hip_fill_i64(rfx_ctx_t *ctx, int64_t *d_dst, int64_t n, int64_t value);
/* Kernels compiled at run time for one plan (hiprtc, loaded on first use; group-bys over at most 8 slots take one): how many launches
 * went through such a kernel and how many compilations ran so far in this process.  RFX_TUNE_NO_RTC / RFX_NO_RTC=1 keep to the prebuilt
 * kernels; so does a box without libhiprtc.so.  The kernel headers are embedded in librfx.so (no source tree needed beside it). */
void rfx_hip_rtc_stats(int64_t *launches, int64_t *compiles);
/* Code objects persist on disk -- <dir of librfx.so>/rtc_cache, RFX_RTC_CACHE=<dir> elsewhere, RFX_RTC_CACHE=0 off -- keyed by the
 * generated text, the embedded headers, the compiler version and options: a plan compiled by ANY earlier process is loaded at first
 * sight.  Counters of this process: plans loaded from disk / code objects written. */
void rfx_hip_rtc_cache_stats(int64_t *loaded_from_disk, int64_t *written_to_disk);
/* Compile the fused filter + aggregate kernel of a plan into that cache WITHOUT a device (the build step pre-warms the BASELINE plans;
 * a deployment can do the same for its recurring queries).  d_col pointers only tell columns apart (any distinct non-NULL values).
 * RFX_OK: the code object is on disk; RFX_ESTATE: no compiler / no cache directory / the plan does not compile. */
int rfx_hip_rtc_prewarm_filter_aggr(const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg);
/* The same for the one-pass `where` kernel of a predicate list / tree (rfx_hip_where_once takes the plan's kernel when there is one). */
int rfx_hip_rtc_prewarm_where(const rfx_pred_t *preds, int npred, int logic);

/* Host-to-device at link speed from ANY host memory (heap vector, mmapped column file): chunks are staged through pinned
 * buffers by worker threads while the previous chunk is in flight.  (syncs) */
int rfx_hip_h2d_pipelined(rfx_ctx_t *ctx, void *d_dst, const void *src, size_t bytes);
/* device -> host for LARGE blocks (>= 64 MB; smaller ones take rfx_hip_d2h): pinned staging, up to four 32 MB chunks in flight, the destination written by
 * several host threads -- the first touch of a freshly allocated result vector's pages happens in parallel instead of inside one pageable copy.  (syncs) */
int rfx_hip_d2h_pipelined(rfx_ctx_t *ctx, void *dst, const void *d_src, size_t bytes);
/* RayforceDB column files (16-byte header {mmod 0xfd, order, type, attrs, rc, len} + raw payload, core/binary.c:263-311):
 * rfx_column_file_stat reads type (the reference's vector type code) and length -- host only, no device needed;
 * rfx_hip_column_file_load maps the file and moves its nrows 8-byte elements into d_dst with the pipelined path.  (syncs) */
int rfx_column_file_stat(const char *path, int32_t *type, int64_t *len);
int rfx_hip_column_file_load(rfx_ctx_t *ctx, const char *path, void *d_dst, int64_t nrows);
/* A 4-byte integer column (I32 / DATE / TIME) as an 8-byte device column: d_out[i] = i32_to_i64(d_in[i]) (core/ops.h:240: NULL_I32 ->
 * NULL_I64, else sign-extended) -- order and equality survive, nulls included, so the 8-byte comparison kernels answer what the
 * reference's i32 comparison arms answer (core/cmp.c:150-166). */
int rfx_hip_widen_i32(rfx_ctx_t *ctx, const int32_t *d_in, int64_t n, int64_t *d_out);
/* a B8 mask as an i64 column of 0 / 1 (a mask-only selection as one comparison of the fused pass) */
int rfx_hip_widen_b8(rfx_ctx_t *ctx, const int8_t *d_in, int64_t n, int64_t *d_out);

/* ---- timing on the context's stream (bench.py measures kernels with these HIP events) ---- */
int rfx_hip_timer_start(rfx_ctx_t *ctx);
int rfx_hip_timer_stop(rfx_ctx_t *ctx, float *ms); /* (syncs) */
/* Kernel-level profile: when enabled, every entry point brackets its DOMINANT kernel (the one streaming pass over the
 * columns: k_filter_aggr, k_sel_bitmap, k_group_dense, k_part_scatter + k_part_aggregate ...) with HIP events on
 * the context's stream; rfx_hip_last_kernel_ms returns that kernel time of the most recent call.  (syncs) */
int rfx_hip_ctx_profile(rfx_ctx_t *ctx, int enable);
int rfx_hip_last_kernel_ms(rfx_ctx_t *ctx, float *ms);
/* ... and of EVERY bracketed kernel launched since the last call of this function (or since profiling was switched on), in launch order -- a
 * group-by through the plane path: [k_plane_scatter, k_plane_aggregate].  At most the last eight; *n = how many were written.  (syncs) */
int rfx_hip_profile_kernels(rfx_ctx_t *ctx, float *ms, int cap, int *n);

/* ---- synthetic columns: counter-based splitmix64, element r = mix(seed + (r+1)*0x9E3779B97F4A7C15) ----
 * i64: value % modulus (modulus > 0);  f64: (value >> 11) * 2^-53 in [0,1).  row0 = global id of d_out[0]. */
int rfx_hip_gen_i64(rfx_ctx_t *ctx, int64_t *d_out, int64_t n, uint64_t seed, int64_t row0, uint64_t modulus);
int rfx_hip_gen_f64(rfx_ctx_t *ctx, double *d_out, int64_t n, uint64_t seed, int64_t row0);

/* ---- K1/K5: fused filter -> scalar aggregates (one pass over each distinct column) ----
 * d_out receives nagg + 1 partials: [0..nagg) one per aggregate, [nagg].cnt = number of selected rows.
 * npred == 0 selects every row.  row0 = global row id of local row 0 (only FIRST uses it). */
int rfx_hip_filter_aggr(rfx_ctx_t *ctx, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                        int nagg, int64_t nrows, int64_t row0, rfx_partial_t *d_out);
/* Host-side helpers (pure C, no device work). */
void rfx_partial_merge(int kind, int col_type, rfx_partial_t *into, const rfx_partial_t *from);
void rfx_partial_identity(rfx_partial_t *p);
/* `grouped` = 0: scalar rules of core/math.c (sum skips nulls, empty min/max -> null, avg none -> NaN). */
int rfx_agg_finalize(int kind, int col_type, const rfx_partial_t *p, rfx_value_t *out);
/* Convenience: run + copy back + finalize.  (syncs) */
int rfx_hip_filter_aggr_host(rfx_ctx_t *ctx, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                             int nagg, int64_t nrows, rfx_value_t *values, int64_t *selected);

/* ---- K2: byte masks (API parity with the reference's materialised B8 results) ---- */
int rfx_hip_cmp_mask(rfx_ctx_t *ctx, const rfx_pred_t *pred, int64_t nrows, int8_t *d_mask);
/* acc[i] = acc[i] && next[i]  (RFX_AND) / || (RFX_OR); next may be NULL with `scalar` used instead. */
int rfx_hip_mask_logic(rfx_ctx_t *ctx, int logic, int8_t *d_acc, const int8_t *d_next, int scalar, int64_t nrows);

/* ---- K3: ordered stream compaction ----
 * where_begin evaluates either the fused predicates (d_mask == NULL) or a byte mask (npred == 0), records a
 * 1-bit-per-row selection bitmap in the context and returns the number of selected rows.  (syncs)
 * where_emit then writes the ascending global row ids  row0 + i  of the selected rows into d_ids[0..count). */
int rfx_hip_where_begin(rfx_ctx_t *ctx, const rfx_pred_t *preds, int npred, int logic, const int8_t *d_mask,
                        int64_t nrows, int64_t *count);
int rfx_hip_where_emit(rfx_ctx_t *ctx, int64_t row0, int64_t *d_ids);
/* The same in ONE pass over the predicate columns (rfx_where_once.hip: ballots -> decoupled look-back -> ids; no bitmap round trip, no
 * separate scan): writes the ascending ids row0 + i of the first min(*count, cap) selected rows into d_ids, *count = the exact number of
 * selected rows whatever `cap` is.  RFX_ELIMIT when *count > cap: call again with cap >= *count.  (syncs)
 * rfx_hip_where_estimate sizes the buffer: an upper guess from 2^15 strided rows (sampled fraction + 4 sigma + 1 % of the column;
 * nrows itself for small inputs).  (syncs) */
int rfx_hip_where_estimate(rfx_ctx_t *ctx, const rfx_pred_t *preds, int npred, int logic, int64_t nrows, int64_t *upper);
int rfx_hip_where_once(rfx_ctx_t *ctx, const rfx_pred_t *preds, int npred, int logic, int64_t nrows, int64_t row0, int64_t *d_ids,
                       int64_t cap, int64_t *count);

/* ---- K4: gather of 8-byte elements: d_out[i] = d_col[d_ids[i]] ---- */
int rfx_hip_gather(rfx_ctx_t *ctx, const void *d_col, const int64_t *d_ids, int64_t m, void *d_out);
/* ... up to RFX_MAX_KEYS 8-byte columns at the same ids in ONE launch (the ids are read once): d_outs[k][i] = d_cols[k][d_ids[i]] */
int rfx_hip_gather_many(rfx_ctx_t *ctx, const void *const *d_cols, int ncols, const int64_t *d_ids, int64_t m, void *const *d_outs);
/* The same for ids that come from OUTSIDE (the `at` operator, a MAPFILTER pair handed over by the host): an id that is null,
 * negative or >= col_len yields the typed null (NULL_I64 / NaN) as at_vec_*_by_i64 does (core/items.c:53-72) -- never a read
 * beyond the column.  rfx_hip_gather stays unchecked for ids the library produced itself. */
int rfx_hip_gather_checked(rfx_ctx_t *ctx, const void *d_col, int64_t col_len, int32_t col_type, const int64_t *d_ids, int64_t m, void *d_out);

/* ---- K6: key scope (min / max of an i64 column, optionally only over rows passing the predicates) ----
 * (syncs)  *count = rows seen; min/max undefined when *count == 0. */
int rfx_hip_scope_i64(rfx_ctx_t *ctx, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic,
                      int64_t nrows, int64_t *min, int64_t *max, int64_t *count);

/* Scope pass of a group-by whose aggregates are known: the same answers as rfx_hip_scope_i64, and for key ranges whose tables
 * overflow one workgroup's LDS the SAME streaming read also radix-partitions the (key, value) rows into context scratch
 * (rfx_group_chunk.hip), so that the rfx_hip_group_dense_accumulate call that follows with the same key / predicates /
 * aggregates only has to aggregate the partitions: the columns are read once instead of three times.  (syncs) */
int rfx_hip_group_scope(rfx_ctx_t *ctx, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic,
                        const rfx_agg_t *aggs, int nagg, int64_t nrows, int64_t *min, int64_t *max, int64_t *count);

/* ---- K7/K8/K10: dense group-by over [kmin, kmin + range) ----
 * Table layout (device, caller-visible so that several GPUs can all-reduce them):
 *   d_first[range]            i64  global row id of the first selected row with that key, INT64_MAX if none (MIN-merge)
 *   d_acc[a][range]           8-byte accumulators of aggregate a; per kind:
 *        SUM   i64: wrap sum (SUM-merge) ; f64: IEEE sum, NaN sticky (SUM-merge)
 *        MIN/MAX  : order-preserving u64 image of the value, see rfx_hip.h notes in DESIGN.md (MIN/MAX-merge)
 *        COUNT    : i64 rows (SUM-merge)
 *        AVG      : f64 sum of non-null values cast to f64 (SUM-merge) ; d_cnt[a] holds the non-null count
 *   d_cnt[a][range]           i64 non-null count (AVG) / null count (SUM over i64: null-sticky flag) (SUM-merge)
 */
typedef struct rfx_group_tables {
    int64_t kmin;
    int64_t range;
    int32_t nagg;
    int32_t _pad;
    int64_t *d_first;
    void *d_acc[RFX_MAX_AGGS];
    int64_t *d_cnt[RFX_MAX_AGGS];
} rfx_group_tables_t;

/* Bytes a caller must provide for one table set (all arrays are `range` * 8 bytes; this returns how many). */
int rfx_hip_group_table_arrays(const rfx_agg_t *aggs, int nagg, int *n_arrays);
/* Set every table to its identity. */
int rfx_hip_group_tables_init(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *t);
/* One pass over (key, predicate columns, aggregate columns): scatter-aggregate local rows [0,nrows). */
int rfx_hip_group_dense_accumulate(rfx_ctx_t *ctx, const int64_t *d_key, const rfx_pred_t *preds, int npred,
                                   int logic, const rfx_agg_t *aggs, int64_t nrows, int64_t row0,
                                   const rfx_group_tables_t *t);
/* Rank occupied slots by first row (first-occurrence order, core/index.c:2037-2055).  total_rows = global
 * number of rows the row ids in d_first range over.  Returns the group count.  (syncs) */
int rfx_hip_group_rank(rfx_ctx_t *ctx, const rfx_group_tables_t *t, int64_t total_rows, int64_t *ngroups);
/* A SAMPLED key scope and the report that keeps it exact.  rfx_hip_scope_sample_i64: [min, max] of 2^18 strided rows + the first and
 * last 2^11 (one tiny launch instead of index_scope_i64's pass over the column, core/index.c:376-435; a null key shows as INT64_MIN).
 * rfx_hip_ctx_speculative(ctx, 1) before rfx_hip_group_dense_accumulate(_keys): LDS-table kernels then REPORT a selected row whose key
 * lies outside the tables' scope (paths that cannot, answer RFX_ESTATE without running); rfx_hip_group_out_of_scope afterwards: nothing
 * reported = the pass is the exact pass (a sampled range is never too large); reported = take rfx_hip_scope_i64 and run again. */
#define RFX_SCOPE_SAMPLE_ROWS (1 << 18) /* strided rows of the sample; trust it for ranges <= RFX_SCOPE_SAMPLE_ROWS / 10 (extreme missed: e^-10) */
#define RFX_SCOPE_SAMPLE_MAX_RANGE 16384 /* ... and only for ranges whose tables are LDS-sized */
int rfx_hip_scope_sample_i64(rfx_ctx_t *ctx, const int64_t *d_key, int64_t nrows, int64_t *min, int64_t *max);
int rfx_hip_ctx_speculative(rfx_ctx_t *ctx, int on);
int rfx_hip_group_out_of_scope(rfx_ctx_t *ctx, int *violated);
/* Few slots (a dense table of at most RFX_RANK_SMALL): rfx_hip_group_rank + rfx_hip_group_emit as ONE launch without a host round
 * trip.  d_block receives [group count][keys][first rows][one array per aggregate], every array `t->range` cells long (the first
 * `group count` of them valid, in first-occurrence order): 1 + (2 + nagg) * range cells, copied to the host in one piece.
 * Replaces the groups++ loop of core/index.c:2037-2055 + the collection of core/aggr.c:163-181 for small key ranges. */
#define RFX_RANK_SMALL 2048
int rfx_hip_group_rank_emit_small(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t row0, int64_t local_rows, int64_t *d_block);
/* Emit, in group order: keys, first row ids, and one result column per aggregate (8-byte elements typed as the
 * reference types them: sum keeps the input type, avg -> f64, count -> i64, min/max keep the input type).
 * Any of d_keys / d_first_ids / d_results[a] may be NULL to skip it. */
int rfx_hip_group_emit(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t *d_keys,
                       int64_t *d_first_ids, void *const *d_results);
/* Row-range sharding: the tables hold GLOBAL first-row ids after the merge, the aggregate columns are this GPU's rows
 * [row0, row0 + local_rows).  FIRST results are the value where this GPU owns the group's first row and 0 elsewhere: summing the
 * result column over the GPUs (as integers) yields the value -- exactly one GPU contributes.  Everything else as rfx_hip_group_emit. */
int rfx_hip_group_emit_sharded(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t row0, int64_t local_rows,
                               int64_t *d_keys, int64_t *d_first_ids, void *const *d_results);
/* Rank -> emit as ONE sequence of launches with no host round trip between them (tables of at most RFX_RANK_EMIT_MAX slots): rfx_hip_group_rank's
 * steps, the last of which also writes the result cells (rfx_hip_group_emit_sharded's), and the group count back at the end (syncs).  The outputs are
 * sized BEFORE the count is known: `out_cap` cells per column (>= the window's groups: min(slots, selected rows) / nsl + 1; RFX_ELIMIT if it did not
 * hold).  nsl > 1: only the groups of slice si (RFX_SLICE_G0 / _GN) are written, from cell 0 on (the emit window, computed on the device). */
#define RFX_RANK_EMIT_MAX (1 << 22)
/* slice i of nsl over g groups: every slice g / nsl groups, the first g % nsl one more -- so that slice 0 is never empty while there are groups (a result's
 * columns are named by slice 0's pointers).  The device computes the same window from the group count it finds. */
#define RFX_SLICE_G0(g, i, nsl) ((int64_t)(i) * ((g) / (nsl)) + ((int64_t)(i) < (g) % (nsl) ? (int64_t)(i) : (g) % (nsl)))
#define RFX_SLICE_GN(g, i, nsl) ((g) / (nsl) + ((int64_t)(i) < (g) % (nsl) ? 1 : 0))
int rfx_hip_group_rank_emit(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t total_rows, int64_t row0, int64_t local_rows,
                            int nsl, int si, int64_t out_cap, int64_t *d_keys, int64_t *d_first_ids, void *const *d_results, int64_t *ngroups);
/* Emit WINDOW: until it is reset (n = 0), rfx_hip_group_emit(_sharded) / rfx_hip_hash_emit(_sharded) on this context write only the groups
 * [g0, g0 + n) -- group g at cell g - g0 of every output, which then need n cells.  The sharded tail of a group-by: after the merge every
 * device holds the whole tables, ranks them (the same order everywhere) and emits ITS slice of the groups, which its own host thread
 * copies into the host result at the slice's offset -- N PCIe links instead of one.  What the reference's pool does for the same step:
 * AGGR_COLLECT fans the per-group finalisation out over its workers by group range (core/aggr.c:163-181, core/pool.c:369-424). */
int rfx_hip_ctx_emit_window(rfx_ctx_t *ctx, int64_t g0, int64_t n);

/* ---- K9: sparse keys (range > rows): open-addressed table, same table/merge contract keyed by slot ---- */
typedef struct rfx_hash_tables {
    int64_t capacity; /* power of two */
    int32_t nagg;
    int32_t _pad;
    int64_t *d_keys;  /* [capacity], empty = RFX_NULL_I64 (as the reference, core/hash.c:35-56) */
    int64_t *d_first;
    void *d_acc[RFX_MAX_AGGS];
    int64_t *d_cnt[RFX_MAX_AGGS];
} rfx_hash_tables_t;
int rfx_hip_hash_tables_init(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *t);
int rfx_hip_group_hash_accumulate(rfx_ctx_t *ctx, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic,
                                  const rfx_agg_t *aggs, int64_t nrows, int64_t row0, const rfx_hash_tables_t *t);
/* ... the same, and -- when the rows went straight into the device-wide table (about as many groups as rows: too large for the partitioned forms) -- every
 * row's slot into d_row_slots[nrows] (-1: not selected) with *recorded = 1; the partitioned forms aggregate in LDS tables first, no row learns its slot there
 * (*recorded = 0: rfx_hip_join_probe_hash_slots answers).  rfx_hip_hash_slot_first: d_ids[i] = the first row of row i's group, from its slot. */
int rfx_hip_group_hash_accumulate_slots(rfx_ctx_t *ctx, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int64_t nrows,
                                        int64_t row0, const rfx_hash_tables_t *t, int64_t *d_row_slots, int *recorded);
int rfx_hip_hash_slot_first(rfx_ctx_t *ctx, const rfx_hash_tables_t *t, const int64_t *d_row_slots, int64_t nrows, int64_t *d_ids);
/* The PACKED form of a hashed table set (round 6; the row-hash route's device-wide table): ONE block of capacity + 1 entries of `stride` cells -- key, first
 * row, accumulators, counts of a slot side by side, so that an insert touches one line -- described by the same struct: d_keys = the block, every other array
 * pointer = another cell of the FIRST entry, and every index into the arrays is slot * stride.  rfx_hip_group_hash_accumulate_packed inserts the rows
 * directly (no partitioned form) and leaves every row's SCALED slot in d_row_slots (-1: not selected); rfx_hip_hash_slot_first and rfx_hip_hash_rows_emit take
 * such a table and such slots as they are.  RFX_ESTATE: a query this form does not carry (expression trees, too many columns for one launch). */
int rfx_hip_hash_tables_init_packed(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int stride);
int rfx_hip_group_hash_accumulate_packed(rfx_ctx_t *ctx, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int64_t nrows,
                                         int64_t row0, const rfx_hash_tables_t *t, int stride, int64_t *d_row_slots);
/* Merge another GPU's table (same capacity) into ours: re-inserts its occupied slots. */
int rfx_hip_hash_tables_merge(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *into,
                              const rfx_hash_tables_t *from);
int rfx_hip_hash_rank(rfx_ctx_t *ctx, const rfx_hash_tables_t *t, int64_t total_rows, int64_t *ngroups);
int rfx_hip_hash_emit(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int64_t *d_keys,
                      int64_t *d_first_ids, void *const *d_results);
int rfx_hip_hash_emit_sharded(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int64_t row0, int64_t local_rows,
                              int64_t *d_keys, int64_t *d_first_ids, void *const *d_results);
/* Emit by ROWS (many groups: the table is large against the rows): the groups are the rows that head their own group (d_probe_first[r] == r, from
 * rfx_hip_join_probe_hash_slots over the group-by's own table), ascending = first-occurrence order.  _begin counts them (syncs); _emit writes the first rows
 * (d_first_ids, ngroups cells), the keys and every aggregate's column from the slots of those rows -- no ranking of the slots, no permutation. */
int rfx_hip_hash_rows_begin(rfx_ctx_t *ctx, const int64_t *d_probe_first, int64_t nrows, int64_t *ngroups);
/* (d_row_keys: the grouped-on column itself, or NULL -- a group's key is then read from it at the group's first row instead of from the table) */
int rfx_hip_hash_rows_emit(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, const int64_t *d_row_slots, const int64_t *d_row_keys, int64_t row0,
                           int64_t local_rows, int64_t ngroups, int64_t *d_keys, int64_t *d_first_ids, void *const *d_results);
int rfx_hip_hash_rank_emit(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int64_t total_rows, int64_t row0, int64_t local_rows,
                           int nsl, int si, int64_t out_cap, int64_t *d_keys, int64_t *d_first_ids, void *const *d_results, int64_t *ngroups);

/* ---- per-row group ids (INDEX_TYPE_IDS payload, core/index.c:2069-2089) -- only when a caller wants it ---- */
int rfx_hip_group_ids_dense(rfx_ctx_t *ctx, const int64_t *d_key, int64_t nrows, const rfx_group_tables_t *t,
                            int64_t *d_gids);
/* ... through a slot -> group id table of the caller's (range cells, on this context's device): a shard that holds rows but did not rank the merged tables */
int rfx_hip_group_ids_table(rfx_ctx_t *ctx, const int64_t *d_key, int64_t nrows, int64_t kmin, int64_t range, const int64_t *d_table, int64_t *d_gids);

/* ... for SPARSE keys (hashed tables: no slot -> id table over the key range): from every row's group-first row (RFX_Q_PROBE_FIRST of the planner: the join probe
 * against the group-by's own table) and the groups' first rows in first-occurrence order: d_gids[d_first[g]] = g, every other row copies its first row's id --
 * the IDS payload of index_group_i64_unscoped (core/index.c:1959-1977) in the one-executor order. */
int rfx_hip_group_ids_first(rfx_ctx_t *ctx, const int64_t *d_probe_first, int64_t nrows, const int64_t *d_first, int64_t groups, int64_t *d_gids);
/* slot -> group id of dense tables after rfx_hip_group_rank, NULL_I64 for an unoccupied slot: the key table of the reference's
 * INDEX_TYPE_SHIFT group index (core/index.c:2037-2062). */
int rfx_hip_group_slot_ids(rfx_ctx_t *ctx, const rfx_group_tables_t *t, int64_t *d_out);

/* ---- several `by:` columns: composite ("perfect hash") key, index_group_list_perfect, core/index.c:2308-2424 ----
 * rfx_composite_plan (host, no device work): from the per-column scopes [mins[i], maxs[i]] compute the multipliers
 * mult_0 = 1, mult_i = mult_{i-1} * range_{i-1} and the composite maximum, with the reference's overflow tests
 * (:2364-2383).  RFX_ELIMIT when the product of ranges does not fit a signed 64-bit integer (the reference then takes its
 * row-hash path, which this library does not cover) -- a null key (INT64_MIN) always ends up there.
 * rfx_hip_composite_key: d_out[r] = sum_i (d_cols[i][r] - mins[i]) * mults[i]  for every local row (:2238-2305).  Rows a
 * later predicate rejects get a meaningless value; the group kernels never look at it.  Group the result with the forced
 * scope {kmin = 0, range = total_max + 1} (dense when range <= selected rows, else hashed), exactly as :2421 does.
 * rfx_hip_composite_decode: d_out[g] = min + (d_comp[g] / mult) % range  -- key column i of the result from the emitted
 * composite keys (equals the reference's key_i[first_row[g]], core/query.c:110-135). */
int rfx_composite_plan(const int64_t *mins, const int64_t *maxs, int nkeys, int64_t *mults, int64_t *total_max);
int rfx_hip_composite_key(rfx_ctx_t *ctx, const void *const *d_cols, const int64_t *mins, const int64_t *mults, int nkeys,
                          int64_t nrows, int64_t *d_out);
int rfx_hip_composite_decode(rfx_ctx_t *ctx, const int64_t *d_comp, int64_t n, int64_t min, int64_t mult, int64_t range,
                             int64_t *d_out);
/* Dense scatter-aggregate keyed by several columns: the same contract as rfx_hip_group_dense_accumulate with
 * slot = sum_i (d_keys[i][row] - mins[i]) * mults[i]  and tables over {kmin = 0, range = total_max + 1}.  Where the tables
 * fit LDS the keys are folded on the fly (no composite column is written); otherwise the composite column is materialised
 * in context-owned scratch (as the reference does) and the single-key paths run on it. */
int rfx_hip_group_dense_accumulate_keys(rfx_ctx_t *ctx, const void *const *d_keys, const int64_t *mins, const int64_t *mults,
                                        int nkeys, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                                        int64_t nrows, int64_t row0, const rfx_group_tables_t *t);

/* ---- several key columns whose ranges overflow the composite key: the row-hash path (index_group_list, core/index.c:2731-2790) ----
 * d_out[r] = the reference's row hash of the key tuple: h = U64_HASH_SEED, then for every column in order
 * h = hash_index_u64(h, col_c[r]) (value_first = 0: unfiltered i64-like columns, core/hash.h:130-143) or
 * h = hash_index_u64(col_c[r], h) (value_first = 1: filtered rows / f64 columns, core/index.c:155-175).
 * Group on d_out with the sparse-key path; rfx_hip_replace_null_i64
 * (d_out[r] = d_col[r] is null ? repl : d_col[r]) prepares key columns with nulls for the min == max collision proof. */
int rfx_hip_row_hash(rfx_ctx_t *ctx, const void *const *d_cols, int nkeys, int64_t nrows, int value_first, int64_t *d_out);
int rfx_hip_replace_null_i64(rfx_ctx_t *ctx, const int64_t *d_col, int64_t nrows, int64_t repl, int64_t *d_out);
/* d_out[r] = d_col[r] == from ? to : d_col[r]; any alignment, d_out may be d_col (a null key through the MIN / MAX proof of a sharded row-hash
 * result and back: rfx_exec_group_by) */
int rfx_hip_replace_i64(rfx_ctx_t *ctx, const int64_t *d_col, int64_t nrows, int64_t from, int64_t to, int64_t *d_out);

/* ---- element-wise arithmetic as a column: ray_add / ray_sub / ray_mul / ray_div over vectors and atoms (binop_map, core/math.c:2280-2345) ----
 * `expr` describes (op x y) or an expression tree exactly as an aggregate's argument does (xop / xnodes fields of rfx_agg_t; kind
 * and d_col are ignored); d_out receives one 8-byte value per row, *out_type its element type.  One pass, no temporaries per
 * operation.  Also what `where:` needs for predicates over expressions: evaluate, then compare the column. */
int rfx_hip_eval_expr(rfx_ctx_t *ctx, const rfx_agg_t *expr, int64_t nrows, void *d_out, int32_t *out_type);

/* ---- equi-joins: lj / ij (ray_left_join / ray_inner_join, core/join.c:158-298; index_left_join_obj, core/index.c:2886-2928) ----
 * The reference's join index is, per LEFT row, the FIRST right row with an equal key, or null.  Build side = the group-by's
 * first-occurrence table over the RIGHT key column with zero aggregates (rfx_hip_group_dense_accumulate into d_first, or
 * rfx_hip_group_hash_accumulate into a hashed table set); these probe it with the LEFT keys:
 *   d_ids[i] = first right row whose key equals d_left_keys[i], else null (INT64_MIN).
 * Several key columns: probe on rfx_hip_row_hash of both sides, then compare the gathered key columns (Engine.join_index).
 * rfx_hip_gather_or assembles a result column (select_column, core/join.c:38-66):
 *   d_out[i] = d_ids[i] is null ? (d_left ? d_left[i] : fill_bits) : d_right[d_ids[i]]. */
int rfx_hip_join_probe_dense(rfx_ctx_t *ctx, const int64_t *d_left_keys, int64_t nleft, int64_t kmin, int64_t range, const int64_t *d_first,
                             int64_t *d_ids);
int rfx_hip_join_probe_hash(rfx_ctx_t *ctx, const int64_t *d_left_keys, int64_t nleft, const rfx_hash_tables_t *t, int64_t *d_ids);
/* ... also leaving, per left row, the table SLOT its key sits in (capacity = the null key's cell, -1 = absent) */
int rfx_hip_join_probe_hash_slots(rfx_ctx_t *ctx, const int64_t *d_left_keys, int64_t nleft, const rfx_hash_tables_t *t, int64_t *d_ids, int64_t *d_slots);
/* one hash = one tuple (__index_list_cmp_row, core/index.c:2465-2790, once per row): *differ = cells of the nk key columns that differ between a row and its
 * group's first row d_first_of_row[r] (0: every row hash stands for exactly one key tuple).  One pass, one counter back.  (syncs) */
int rfx_hip_tuple_check(rfx_ctx_t *ctx, const void *const *d_keys, int nk, const int64_t *d_first_of_row, int64_t nrows, int64_t *differ);
int rfx_hip_gather_or(rfx_ctx_t *ctx, const void *d_right, const void *d_left, const int64_t *d_ids, int64_t n, uint64_t fill_bits, void *d_out);

/* ---- multi-GPU: row-range sharding, one process per GPU, ONE exchange per query over RCCL / xGMI (rfx_dist.hip) ----
 * GPU g holds rows [row0_g, row0_g + n_g) of every column; the per-GPU partial states merge exactly as the reference merges its
 * per-thread partials (AGGR_COLLECT core/aggr.c:163-181, unop_fold core/math.c:2206-2228, core/index.c:1866-1906).
 * Communicator: rank 0 draws 128 bytes with rfx_dist_unique_id, hands them to every rank through the host's own side channel,
 * every rank calls rfx_dist_init (collective).  RCCL is loaded at that moment (dlopen), never by single-GPU users.  Without a
 * communicator every call below is the identity (world 1). */
#define RFX_DIST_ID_BYTES 128
int rfx_dist_unique_id(void *id128);
int rfx_dist_init(rfx_ctx_t *ctx, int world, int rank, const void *id128); /* collective */
int rfx_dist_finalize(rfx_ctx_t *ctx);
int rfx_dist_world(rfx_ctx_t *ctx, int *world, int *rank);
/* key scope agreed over the ranks: in = this rank's (min, max, rows seen), out = the table's.  One all-gather of 32 B.  (syncs) */
int rfx_dist_scope(rfx_ctx_t *ctx, int64_t *kmin, int64_t *kmax, int64_t *seen);
/* dense group tables, in place: first MIN, sums / counts SUM, min / max MIN / MAX on the ordered image -- one fused exchange
 * (ncclGroupStart .. End; neighbouring arrays of one class become one call), asynchronous on the context's stream */
int rfx_dist_group_tables_allreduce(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *t);
/* scalar partials: d_all[r * n + i] = rank r's d_local[i]; asynchronous */
int rfx_dist_partials_allgather(rfx_ctx_t *ctx, const rfx_partial_t *d_local, int n, rfx_partial_t *d_all);
/* rfx_hip_filter_aggr_host over the sharded table: local fused pass + one all-gather + the rank-ordered fold.  (syncs) */
int rfx_dist_filter_aggr_host(rfx_ctx_t *ctx, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg,
                              int64_t nrows, int64_t row0, rfx_value_t *values, int64_t *selected);
/* in-place all-reduce of 8-byte integers, op 0 SUM / 1 MIN / 2 MAX; all-gather of `bytes` bytes per rank; both asynchronous */
int rfx_dist_allreduce_i64(rfx_ctx_t *ctx, int64_t *d_buf, int64_t n, int op);
int rfx_dist_allgather(rfx_ctx_t *ctx, const void *d_in, size_t bytes, void *d_out);
int64_t rfx_dist_calls(rfx_ctx_t *ctx); /* collectives issued so far */

/* ---- ONE process driving several devices / several shards per device (rfx_exec.c: the operator layer's planner) ----
 * rfx_dist_init_all: one communicator per context, created by THIS process (ncclCommInitAll) -- the contexts sit on distinct devices;
 * such communicators are process-local (rfx_dist_is_local): what the host can fold itself never goes through RCCL.  The *_all calls
 * issue one fused exchange over the contexts from the calling thread (ncclGroupStart .. End). */
int rfx_dist_init_all(rfx_ctx_t *const *ctxs, int n);
int rfx_dist_is_local(rfx_ctx_t *ctx);
int rfx_dist_has_comm(rfx_ctx_t *ctx); /* 1: the context carries a communicator (of either kind) */
int rfx_dist_group_tables_allreduce_all(rfx_ctx_t *const *ctxs, int n, const rfx_agg_t *aggs, const rfx_group_tables_t *const *tables);
int rfx_dist_allreduce_i64_all(rfx_ctx_t *const *ctxs, int n, int64_t *const *d_bufs, int64_t cells, int op);
int rfx_dist_allgather_all(rfx_ctx_t *const *ctxs, int n, const void *const *d_ins, size_t bytes, void *const *d_outs);
/* `bytes` of HOST memory from every rank in rank order (scopes, scalar partials, flags); identity without a communicator.  (syncs) */
int rfx_dist_allgather_host(rfx_ctx_t *ctx, const void *in, size_t bytes, void *out);
/* shards that share a device merge by a kernel: into[slot] (op)= from[slot] for every array of two dense table sets over one scope
 * (first MIN, sums / counts SUM, min / max on the ordered image) -- AGGR_COLLECT's element-wise merge, core/aggr.c:163-181 */
int rfx_hip_group_tables_merge(rfx_ctx_t *ctx, const rfx_agg_t *aggs, const rfx_group_tables_t *into, const rfx_group_tables_t *from);
int rfx_hip_add_i64(rfx_ctx_t *ctx, int64_t *d_into, const int64_t *d_from, int64_t n); /* FIRST values: one shard contributed each */
int rfx_hip_d2d(rfx_ctx_t *ctx, void *d_dst, const void *d_src, size_t bytes);           /* asynchronous; the source may sit on a peer device */
int rfx_hip_ctx_bind_thread(rfx_ctx_t *ctx); /* hipSetDevice(ctx's device) for the calling thread */
int rfx_hip_ctx_device(rfx_ctx_t *ctx);

/* ---- `update ... where / by` (ray_update, core/update.c:936-1106): the writes, on a device copy of the column ----
 * rfx_hip_update_set:   d_col[d_ids[i]] = d_vals ? d_vals[d_ids[i]] : atom_bits   for i < m  (d_ids == NULL: rows 0 .. m-1).  d_vals is the
 *                       element-wise mapping evaluated over ALL rows (rfx_hip_eval_expr), so its value at a selected row is what
 *                       the reference's evaluation over the filtered table yields (set_ids, update.c filter arm).
 * rfx_hip_update_group: d_col[row] = the final aggregate of row's group, for the selected rows -- tables of ONE aggregate filled by
 *                       rfx_hip_group_dense_accumulate over the same selection (aggr_row + set_ids per group, update.c:781-850). */
int rfx_hip_update_set(rfx_ctx_t *ctx, void *d_col, const int64_t *d_ids, int64_t m, const void *d_vals, uint64_t atom_bits);
/* over shards: d_out[i] = (d_mask01 ? d_mask01[i] != 0 : 1) ? (d_vals ? d_vals[i] : atom_bits) : (d_old ? d_old[i] : null_bits) -- one row-local write, no ids */
int rfx_hip_update_select(rfx_ctx_t *ctx, void *d_out, const void *d_old, uint64_t null_bits, const int64_t *d_mask01, const void *d_vals, uint64_t atom_bits, int64_t n);
int rfx_hip_update_group(rfx_ctx_t *ctx, void *d_col, const int64_t *d_key, const int64_t *d_ids, int64_t m, const rfx_agg_t *agg,
                         const rfx_group_tables_t *t);

/* ---- reproducible grouped f64 sums (opt-in; rfx_ops_set_deterministic): a column scaled by 2^k and rounded to i64 once per cell sums to the same bits in
 * any order.  rfx_hip_absmax_f64: max |x| over the column and whether a NaN / infinity occurs (syncs); rfx_hip_fix_f64: d_out[i] = llrint(d_in[i] * 2^k),
 * d_out may be d_in. */
int rfx_hip_absmax_f64(rfx_ctx_t *ctx, const double *d_in, int64_t n, double *absmax, int *nonfinite);
int rfx_hip_fix_f64(rfx_ctx_t *ctx, const double *d_in, int64_t n, int k, int64_t *d_out);
/* ... and the part of every cell that rfx_hip_fix_f64 rounded away, as a second limb: llrint((x * 2^k - rint(x * 2^k)) * 2^m), 0 <= m <= 62 (not in place) */
int rfx_hip_fix_f64_low(rfx_ctx_t *ctx, const double *d_in, int64_t n, int k, int m, int64_t *d_out);
/* ... and a result column of such integer sums back as f64, in place: (double)hi * sc + (double)lo * sc2 (d_lo may be NULL), divided by d_cnt[i] (may be NULL;
 * a zero count gives NaN) */
int rfx_hip_unfix_f64(rfx_ctx_t *ctx, int64_t *d_hi_io, const int64_t *d_lo, const int64_t *d_cnt, int64_t n, double sc, double sc2);

/* ---- bucketed group keys: (xbar col width), XBARI64 core/ops.h:192-193 ----
 * d_out[r] = null for a null input, else the largest multiple of `width` (> 0) that is <= d_col[r].  Group on the result
 * (`by: {t: (xbar ts 60000)}`). */
int rfx_hip_xbar_i64(rfx_ctx_t *ctx, const int64_t *d_col, int64_t nrows, int64_t width, int64_t *d_out);

/* ---- hash primitives pinned against the reference (core/hash.c:530-542, core/hash.h:86-97) ---- */
int rfx_hip_hash_fnv1a_i64(rfx_ctx_t *ctx, const int64_t *d_in, int64_t n, uint64_t *d_out);
int rfx_hip_hash_mix_u64(rfx_ctx_t *ctx, const uint64_t *d_in, int64_t n, uint64_t seed_or_prev, uint64_t *d_out);

#ifndef __HIPCC_RTC__
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RFX_HIP_H */
