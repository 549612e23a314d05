// rfx_common.hpp -- internal (not part of the C ABI): context layout, launch plan, device-side scalar rules.
// gfx950 / wave64 only.  The scalar rules restate core/ops.h:63-197 of the reference (cited per function).
#pragma once
// Kernels compiled at run time for one plan (hiprtc, rfx_rtc.hip) include this header as well: there the HIP device builtins are
// pre-included, the C library headers do not exist, and everything host-side is left out (__HIPCC_RTC__).
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#endif
#include <stdint.h>
#include "../../include/rfx_hip.h"

#define RFX_WAVE 64
#define RFX_BLOCK 256 /* 4 waves: one per SIMD of a CU */
#define RFX_NULL_I64_D ((int64_t)0x8000000000000000LL)
#define RFX_INF_I64_D ((int64_t)0x7FFFFFFFFFFFFFFFLL)
#define RFX_NAN_BITS 0x7FF8000000000000ULL
#define RFX_PINF_BITS 0x7FF0000000000000ULL

typedef unsigned long long u64;
typedef long long i64;

#ifndef __HIPCC_RTC__
void rfx_set_error(const char *fmt, ...);

#define RFX_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            rfx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
            return (_e == hipErrorOutOfMemory) ? RFX_ENOMEM : RFX_EHIP;                           \
        }                                                                                         \
    } while (0)

#define RFX_REQUIRE(cond, code, msg)                      \
    do {                                                  \
        if (!(cond)) {                                    \
            rfx_set_error("%s: %s", __func__, msg);       \
            return (code);                                \
        }                                                 \
    } while (0)

struct rfx_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int num_cus;
    int blocks_per_cu; // streaming-kernel grid = num_cus * blocks_per_cu
    int flags;
    hipEvent_t ev0, ev1;
    hipEvent_t evk[8][2]; // kernel brackets (profile mode): pair (evk_n & 7) is the next one -- the last eight bracketed kernels, in launch order
    int evk_n;
    int profile;
    int evk_valid;
    // scratch: per-block partials of the fused reductions
    void *d_ws;
    size_t ws_bytes;
    // pinned host staging for small read-backs
    void *h_pin;
    size_t pin_bytes;
    // ordered-compaction state (where_begin -> where_emit, group rank -> emit)
    u64 *d_bitmap;      // 1 bit per row
    size_t bitmap_cap;  // in 64-bit words
    i64 *d_blksum;      // per 2048-row block: popcount, then exclusive prefix
    size_t blksum_cap;  // entries
    i64 where_n;        // rows covered by the bitmap
    i64 where_count;    // selected rows
    i64 rank_rows;      // group rank: total rows of the bitmap
    i64 rank_groups;
    i64 *d_gid;         // group rank: slot -> group id (or -1)
    size_t gid_cap;
    void *d_part;       // partitioned group-by: offsets + record planes (grow-only)
    size_t part_bytes;
    void *d_comp;       // materialised composite key column of a multi-key group-by (grow-only)
    size_t comp_bytes;
    void *io_stage[4];  // pinned staging buffers of the pipelined host-to-device path (rfx_io.hip), created on first use
    hipEvent_t io_done[4];
    void *d_expr;       // materialised expression columns (grow-only)
    size_t expr_bytes;
    void *d_sel;        // compacted (key, values, row ids) of a selectively filtered partitioned group-by (grow-only)
    size_t sel_bytes;
    // scope + low-bit histogram computed together by rfx_hip_scope_i64 and consumed by the next group_dense_accumulate
    u64 *d_pc_counts;   // [pc_nwg][256]
    int pc_valid, pc_npred, pc_logic, pc_nwg;
    int pc_bitmap;      // that scope pass also left the selection in d_bitmap
    const void *pc_key;
    i64 pc_nrows;
    i64 pc_seen;        // rows that passed the predicates in that scope pass
    u64 pc_sig[RFX_MAX_PREDS][6];
    // one-pass chunk partitioning left by rfx_hip_group_scope (rfx_group_chunk.hip) for the next group_dense_accumulate
    void *d_chunk;      // ctl + scope partials + chunk counts + metas + chunk lists + record pool (grow-only)
    size_t chunk_bytes;
    int ck_valid, ck_npred, ck_logic, ck_nwg, ck_chs;
    const void *ck_key, *ck_val;
    i64 ck_nrows, ck_tpw;
    size_t ck_max_chunks;
    u64 ck_sig[RFX_MAX_PREDS][6];
    // multi-GPU exchange (rfx_dist.hip): RCCL communicator of this context's rank, scratch for the small gathers
    void *comm;
    int world, rank;
    void *d_dist;
    size_t dist_bytes;
    i64 dist_calls;     // collectives issued (tests count them)
    void *ext_p[8];     // reserved: new state goes here without touching the layout every translation unit was compiled against
    i64 ext_i[8];
};

// more context state without touching the layout above: ext_p[7] points at this block (made by rfx_hip_ctx_create, zeroed)
struct CtxExt {
    i64 emit_g0, emit_gn; // emit window (rfx_hip_ctx_emit_window): gn == 0 = none
    i64 spare[30];
};
static inline CtxExt *rfx_ext(rfx_ctx *c) { return (CtxExt *)c->ext_p[7]; }

int rfx_ws_reserve(rfx_ctx *ctx, size_t bytes);
int rfx_bitmap_reserve(rfx_ctx *ctx, i64 nrows);
int rfx_gid_reserve(rfx_ctx *ctx, i64 slots);
int rfx_part_reserve(rfx_ctx *ctx, size_t bytes);
int rfx_comp_reserve(rfx_ctx *ctx, size_t bytes);
int rfx_expr_reserve(rfx_ctx *ctx, size_t bytes);
void rfx_io_release(rfx_ctx *ctx);
int rfx_sel_reserve(rfx_ctx *ctx, size_t bytes);
int rfx_chunk_reserve(rfx_ctx *ctx, size_t bytes);
int rfx_plan_add_col(struct Plan *P, const void *col); // index of `col` in P->cols (added if new), -1 when full
#define RFX_KERNEL_BEGIN(c) do { if ((c)->profile) { (void)hipEventRecord((c)->evk[(c)->evk_n & 7][0], (c)->stream); } } while (0)
#define RFX_KERNEL_END(c) do { if ((c)->profile) { (void)hipEventRecord((c)->evk[(c)->evk_n & 7][1], (c)->stream); (c)->evk_n++; (c)->evk_valid = 1; } } while (0)
static inline int rfx_grid(const rfx_ctx *ctx) { return ctx->num_cus * ctx->blocks_per_cu; }
int rfx_rtc_filter_aggr(rfx_ctx *c, const struct Plan &P, int grid, void *ws, int *na_stride); // rfx_rtc.hip; RFX_ESTATE: the prebuilt kernels run
#else
struct rfx_ctx;
#endif

// ------------------------------------------------------------------------------------------------
// Launch plan: what one fused pass reads and computes.  Passed to kernels by value (kernarg segment).
// ------------------------------------------------------------------------------------------------
struct PlanPred {
    int col;      // index into Plan::cols
    int rhs_col;  // -1: atom in rhs_bits ; else index into Plan::cols
    int op;       // RFX_EQ..RFX_GE
    int dom_f64;  // 1: compare as f64 (core/cmp.c: mt = f64), 0: as i64
    int lhs_cvt;  // 1: column is i64 but domain is f64 -> i64_to_f64 (null -> NaN), core/ops.h:250
    int rhs_cvt;  // same for a vector rhs
    u64 rhs_bits; // atom, already promoted to the comparison domain on the host
    int more;     // 1: same parenthesis as the next predicate (rfx_pred_t::more); the parenthesis combines with the opposite of Plan::logic
    int tree;     // 0: flat list / two-level form (`more`).  Else RFX_PRED_TREE | depth | close << 4: a leaf of an arbitrarily nested and / or
                  // tree (rfx_pred_t::more, RFX_PRED_TREE form) -- level 0 combines with Plan::logic, every deeper level with the opposite
                  // of the level above; `close` parentheses end after this leaf
};
#define RFX_XCOL 64 /* PlanAgg::col >= RFX_XCOL: the aggregate folds expression Plan::xs[col - RFX_XCOL] */
struct PlanAgg {
    int col;      // index into Plan::cols (-1 for COUNT without a column), or RFX_XCOL + expression index
    int f64;      // folded element type is f64 (column type, or the expression's result type)
    int kind;     // RFX_AGG_* ; -1 = unused slot
    int skipnull; // grouped: fold with the SCALAR rules (nulls skipped, all-null group -> null) -- what the reference does
                  // when the aggregate's argument is an expression (per-group vectors folded one by one)
};
// One element-wise expression: up to RFX_MAX_XNODES operations in evaluation order.  An operand is a plan
// column, an atom, or the result of an earlier operation; the expression's value is the last operation's result.
struct PlanXNode {
    int op;             // RFX_X_ADD .. RFX_X_MOD
    int o_f64;          // result type of this operation
    int l_kind, r_kind; // RFX_XK_COL / RFX_XK_ATOM / RFX_XK_NODE
    int l_idx, r_idx;   // plan column index (COL) or earlier node index (NODE)
    int l_f64, r_f64;   // operand element types
    u64 l_atom, r_atom; // atom bits in the operand's own type
};
struct PlanExpr {
    int nops;
    int out_f64; // == ops[nops - 1].o_f64
    PlanXNode ops[RFX_MAX_XNODES];
};
struct Plan {
    int ncols, npred, nagg, logic;
    const u64 *cols[RFX_MAX_COLS];
    PlanPred preds[RFX_MAX_PREDS];
    PlanAgg aggs[RFX_MAX_AGGS];
    i64 nrows;
    i64 row0;
    int nx, _pad;
    PlanExpr xs[RFX_MAX_EXPRS];
};
// Evaluate every expression into context scratch and rewrite the plan to read the results as plain columns (nx = 0):
// for the kernels that do not fold expressions on the fly (partitioned and hashed group-by).
int rfx_plan_materialise_exprs(rfx_ctx *c, Plan *P);
// Any expression tree (more than one operation)?  The group-by kernels evaluate single operations in place and read trees as
// materialised scratch columns.
static inline bool rfx_plan_has_deep_expr(const Plan &P) {
    for (int i = 0; i < P.nx; i++)
        if (P.xs[i].nops > 1) return true;
    return false;
}

// Build a Plan from the public descriptors (dedupes columns, promotes atoms).  Returns RFX_OK or an error.
int rfx_plan_build(Plan *P, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg,
                   const void *extra_col, int *extra_idx, i64 nrows, i64 row0);

// ------------------------------------------------------------------------------------------------
// Device-side scalar rules
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rfx_as_f64(u64 b) { return __longlong_as_double((i64)b); }
__device__ __forceinline__ u64 rfx_as_u64(double d) { return (u64)__double_as_longlong(d); }

// ISNANF64 -- bit-pattern test, core/ops.h:63-70
__device__ __forceinline__ bool rfx_isnan_bits(u64 u) {
    return (u & 0x7FF0000000000000ULL) == 0x7FF0000000000000ULL && (u & 0x000FFFFFFFFFFFFFULL) != 0;
}
// i64_to_f64 -- core/ops.h:250 : null -> NaN
__device__ __forceinline__ u64 rfx_i64_to_f64_bits(u64 x) {
    return ((i64)x == RFX_NULL_I64_D) ? RFX_NAN_BITS : rfx_as_u64((double)(i64)x);
}

// ADD/SUB/MUL{I64,F64}, FDIV{I64,F64} -- core/ops.h:153-174, with the binop type promotion of core/math.c (i64 (x) f64 ->
// f64 through i64_to_f64).  lx / rx are the operands' raw bits in their own types.
// DIVI64 / MODI64 (core/ops.h:165-176): floor division (EUCL_DIV rounds the truncated quotient down when the signs differ and the
// division is inexact), the remainder takes the divisor's sign; a zero divisor or a null -> null.  DIVF64 / MODF64 (:167-177):
// floor(x / y), x - floor(x / y) * y.  ray_div keeps the LEFT operand's type: i64 / f64 -> f64_to_i64(floor((double)x / y)).
__device__ __forceinline__ u64 rfx_expr_eval(int op, int out_f64, int l_f64, int r_f64, u64 lx, u64 rx) {
    if (op >= RFX_X_DIV) {
        if (!l_f64 && !r_f64) {
            const i64 x = (i64)lx, y = (i64)rx;
            if (y == 0 || x == RFX_NULL_I64_D || y == RFX_NULL_I64_D) return (u64)RFX_NULL_I64_D;
            i64 q = x / y;
            if (((x < 0) != (y < 0)) && q * y != x) q -= 1;
            return op == RFX_X_DIV ? (u64)q : (u64)(x - q * y);
        }
        const u64 lb = l_f64 ? lx : rfx_i64_to_f64_bits(lx), rb = r_f64 ? rx : rfx_i64_to_f64_bits(rx);
        const double a = rfx_as_f64(lb), b = rfx_as_f64(rb);
        double r = 0.0;
        bool null = b == 0.0 || rfx_isnan_bits(lb) || rfx_isnan_bits(rb);
        if (!null) {
            const double q = floor(a / b);
            r = op == RFX_X_DIV ? q : fma(-q, b, a); // (fused, as the reference's build contracts it: tests/golden/divmod_golden.npz)
            null = rfx_isnan_bits(rfx_as_u64(r)); // (inf / inf and the like: a NaN result is the null)
        }
        if (out_f64) return null ? RFX_NAN_BITS : rfx_as_u64(r);
        // i64 result of i64 / f64: f64_to_i64 (core/ops.h:255), out of range -> the null (x86's cvttsd2si "indefinite" = INT64_MIN)
        if (null || !(r > -9223372036854775808.0 && r < 9223372036854775808.0)) return (u64)RFX_NULL_I64_D;
        return (u64)(i64)r;
    }
    if (!out_f64) { // i64 (x) i64, op in {ADD, SUB, MUL}: null in -> null out, two's-complement wrap
        if ((i64)lx == RFX_NULL_I64_D || (i64)rx == RFX_NULL_I64_D) return (u64)RFX_NULL_I64_D;
        return op == RFX_X_ADD ? lx + rx : (op == RFX_X_SUB ? lx - rx : lx * rx);
    }
    const u64 lb = l_f64 ? lx : rfx_i64_to_f64_bits(lx), rb = r_f64 ? rx : rfx_i64_to_f64_bits(rx);
    if (rfx_isnan_bits(lb) || rfx_isnan_bits(rb)) return RFX_NAN_BITS;
    const double a = rfx_as_f64(lb), b = rfx_as_f64(rb);
    double r;
    switch (op) {
        case RFX_X_ADD: r = a + b; break;
        case RFX_X_SUB: r = a - b; break;
        case RFX_X_MUL: r = a * b; break;
        default:
            if (b == 0.0) return RFX_NAN_BITS; // FDIV*: zero divisor -> null
            r = a / b;
            break;
    }
    return rfx_as_u64(r);
}

// {EQ,NE,LT,GT,LE,GE}I64 -- core/ops.h:80,88,96,104,112,120 : plain signed compares, no null test
__device__ __forceinline__ bool rfx_cmp_i64(int op, i64 x, i64 y) {
    switch (op) {
        case RFX_EQ: return x == y;
        case RFX_NE: return x != y;
        case RFX_LT: return x < y;
        case RFX_GT: return x > y;
        case RFX_LE: return x <= y;
        default: return x >= y;
    }
}
// {EQ,NE,LT,GT,LE,GE}F64 -- core/ops.h:81,89,97,105,113,121 : NaN is the smallest value, NaN == NaN
__device__ __forceinline__ bool rfx_ltf64(u64 xb, u64 yb) {
    bool xn = rfx_isnan_bits(xb), yn = rfx_isnan_bits(yb);
    return xn ? !yn : (yn ? false : rfx_as_f64(xb) < rfx_as_f64(yb));
}
__device__ __forceinline__ bool rfx_gtf64(u64 xb, u64 yb) {
    bool xn = rfx_isnan_bits(xb), yn = rfx_isnan_bits(yb);
    return yn ? !xn : (xn ? false : rfx_as_f64(xb) > rfx_as_f64(yb));
}
__device__ __forceinline__ bool rfx_eqf64(u64 xb, u64 yb) {
    bool xn = rfx_isnan_bits(xb), yn = rfx_isnan_bits(yb);
    return xn ? yn : (yn ? false : rfx_as_f64(xb) == rfx_as_f64(yb));
}
__device__ __forceinline__ bool rfx_cmp_f64(int op, u64 x, u64 y) {
    switch (op) {
        case RFX_EQ: return rfx_eqf64(x, y);
        case RFX_NE: return !rfx_eqf64(x, y);
        case RFX_LT: return rfx_ltf64(x, y);
        case RFX_GT: return rfx_gtf64(x, y);
        case RFX_LE: return !rfx_gtf64(x, y);
        default: return !rfx_ltf64(x, y);
    }
}

// Order-preserving image of a non-NaN f64 in signed-i64 order (so integer min/max == IEEE min/max, with
// -0.0 < +0.0 as a deterministic tie-break the reference leaves to summation order).
__device__ __host__ __forceinline__ i64 rfx_f64_to_ord(u64 b) { return (i64)(b ^ (((i64)b >> 63) & 0x7FFFFFFFFFFFFFFFULL)); }
__device__ __host__ __forceinline__ u64 rfx_ord_to_f64(i64 o) { return (u64)(o ^ ((o >> 63) & 0x7FFFFFFFFFFFFFFFLL)); }

// splitmix64 output for counter value `state`
__device__ __host__ __forceinline__ u64 rfx_splitmix_mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// 64-bit wave shuffles
__device__ __forceinline__ u64 rfx_shfl_xor_u64(u64 v, int m) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((u64)hi << 32) | lo;
}

// 16-byte streaming load of two consecutive 8-byte elements
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
__device__ __forceinline__ u64x2 rfx_ld2(const u64 *p) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    v2 t = __builtin_nontemporal_load((const v2 *)p);
    u64x2 r; r.x = t.x; r.y = t.y;
    return r;
}
