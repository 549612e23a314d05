// rfx_scalar_kernel.hpp -- internal: the fused predicate -> aggregate register-tile machinery (K1/K5),
// shared by rfx_scalar.hip (final fold, masks), rfx_scalar_nc.hip (one translation unit per column count so the
// template instantiations compile in parallel), rfx_where.hip and the group-by files.
//
// Shape of the hot loop (what the ISA should look like on gfx950):
//   * every lane owns E = 2*U rows of a tile: U x global_load_dwordx4 ... nt per column, all issued before the first is used.
//     There is NO register double buffering across tiles (the loop loads a tile, then folds it): what keeps HBM requests in
//     flight across the ALU section is occupancy -- four 256-lane workgroups per CU (C2: 4.3 -> 6.7 TB/s from 2 -> 4 per CU);
//   * a predicate is ONE v_cmp per row whose result lives in an SGPR pair (a wave-wide lane mask); AND/OR of several
//     predicates are s_and_b64 / s_or_b64 on those masks -- no per-lane integer bit fiddling;
//   * sums are v_cndmask + 64-bit add per selected row; every COUNT-like quantity is s_bcnt1 of the mask, kept
//     wave-uniform (lane 0 carries it into the reduction), so counting costs no VALU at all.
#pragma once
#include "rfx_common.hpp"

struct Acc {
    u64 v;
    i64 c;
};

__device__ __forceinline__ void acc_init(Acc &a, int kind) {
    a.v = 0;
    a.c = 0;
    if (kind == RFX_AGG_MIN) a.v = (u64)RFX_INF_I64_D;
    else if (kind == RFX_AGG_MAX) a.v = (u64)RFX_NULL_I64_D;
    else if (kind == RFX_AGG_FIRST) a.c = RFX_INF_I64_D;
}

__device__ __forceinline__ void acc_combine(Acc &a, const Acc &b, int kind, int f64) {
    switch (kind) {
        case RFX_AGG_SUM:
        case RFX_AGG_AVG:
            if (f64) a.v = rfx_as_u64(rfx_as_f64(a.v) + rfx_as_f64(b.v));
            else a.v += b.v;
            a.c += b.c;
            break;
        case RFX_AGG_MIN:
            a.v = ((i64)b.v < (i64)a.v) ? b.v : a.v;
            a.c += b.c;
            break;
        case RFX_AGG_MAX:
            a.v = ((i64)b.v > (i64)a.v) ? b.v : a.v;
            a.c += b.c;
            break;
        case RFX_AGG_COUNT:
            a.c += b.c;
            break;
        case RFX_AGG_FIRST:
            if (b.c < a.c) a = b;
            break;
        default:
            break;
    }
}

__device__ __forceinline__ Acc acc_shfl_xor(const Acc &a, int m) {
    Acc r;
    r.v = rfx_shfl_xor_u64(a.v, m);
    r.c = (i64)rfx_shfl_xor_u64((u64)a.c, m);
    return r;
}

// pick column `col` (wave-uniform) out of the register tile without dynamic register indexing
template <int NC, int E>
__device__ __forceinline__ void sel_col(u64 (&x)[E], const u64 (&v)[NC][E], int col) {
    if (NC == 1) {
#pragma unroll
        for (int e = 0; e < E; e++) x[e] = v[0][e];
        return;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (col == c) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = v[c][e];
        }
    }
}

// The input of an expression aggregate, computed from the register tile: no temporary column exists.
// Operations run in order; operand selection is wave-uniform (column index / atom / earlier result), values are per lane.
template <int NC, int E>
__device__ __forceinline__ void expr_operand(u64 (&o)[E], const u64 (&v)[NC][E], int kind, int idx, u64 atom, const u64 (&r0)[E], const u64 (&r1)[E],
                                             const u64 (&r2)[E]) {
    if (kind == RFX_XK_COL) sel_col<NC, E>(o, v, idx);
    else if (kind == RFX_XK_ATOM) {
#pragma unroll
        for (int e = 0; e < E; e++) o[e] = atom;
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) o[e] = (idx == 0) ? r0[e] : ((idx == 1) ? r1[e] : r2[e]);
    }
}
// One operation over columns / atoms: the common case, and the only one the group-by kernels evaluate in place (keeping the
// general evaluator out of them keeps their register count where it was).
template <int NC, int E>
__device__ __forceinline__ void expr_input(u64 (&x)[E], const u64 (&v)[NC][E], const PlanExpr &X) {
    const PlanXNode n = X.ops[0];
    u64 l[E], r[E];
    if (n.l_kind == RFX_XK_COL) sel_col<NC, E>(l, v, n.l_idx);
    else {
#pragma unroll
        for (int e = 0; e < E; e++) l[e] = n.l_atom;
    }
    if (n.r_kind == RFX_XK_COL) sel_col<NC, E>(r, v, n.r_idx);
    else {
#pragma unroll
        for (int e = 0; e < E; e++) r[e] = n.r_atom;
    }
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, l[e], r[e]);
}
// Expression trees (up to RFX_MAX_XNODES operations, operands may be earlier results): K1's DEEP instantiations only.
template <int NC, int E>
__device__ __forceinline__ void expr_input_deep(u64 (&x)[E], const u64 (&v)[NC][E], const PlanExpr &X) {
    u64 res[RFX_MAX_XNODES - 1][E]; // results of the operations before the last one (never read beyond nops - 2)
#pragma unroll
    for (int i = 0; i < RFX_MAX_XNODES - 1; i++) {
#pragma unroll
        for (int e = 0; e < E; e++) res[i][e] = 0;
    }
#pragma unroll
    for (int i = 0; i < RFX_MAX_XNODES; i++) {
        if (i >= X.nops) break; // wave-uniform
        const PlanXNode n = X.ops[i];
        u64 l[E], r[E];
        expr_operand<NC, E>(l, v, n.l_kind, n.l_idx, n.l_atom, res[0], res[1], res[2]);
        expr_operand<NC, E>(r, v, n.r_kind, n.r_idx, n.r_atom, res[0], res[1], res[2]);
        if (i == X.nops - 1) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, l[e], r[e]);
        } else if (i < RFX_MAX_XNODES - 1) {
#pragma unroll
            for (int e = 0; e < E; e++) res[i][e] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, l[e], r[e]);
        }
    }
}


// The same selections through a wave-uniform SWITCH (one scalar branch, then plain moves from the statically indexed column)
// instead of a compare-and-select chain over all NC columns (NC x E x 2 v_cndmask per selection): for wide plans -- 7 columns,
// a dozen selections per tile -- the chains were a quarter of the group kernel's instructions.  Used by k_group_dense's TINY form.
template <int NC, int E>
__device__ __forceinline__ void sel_col_sw(u64 (&x)[E], const u64 (&v)[NC][E], int col) {
#define RFX_SEL_CASE(c)                                  \
    case c:                                              \
        if (c < NC) {                                    \
            _Pragma("unroll") for (int e = 0; e < E; e++) x[e] = v[c < NC ? c : 0][e]; \
        }                                                \
        break;
    switch (col) {
        RFX_SEL_CASE(0) RFX_SEL_CASE(1) RFX_SEL_CASE(2) RFX_SEL_CASE(3) RFX_SEL_CASE(4) RFX_SEL_CASE(5) RFX_SEL_CASE(6) RFX_SEL_CASE(7)
        default: break;
    }
#undef RFX_SEL_CASE
}
template <int NC, int E>
__device__ __forceinline__ void expr_input_deep_sw(u64 (&x)[E], const u64 (&v)[NC][E], const PlanExpr &X) {
    u64 res[RFX_MAX_XNODES - 1][E];
#pragma unroll
    for (int i = 0; i < RFX_MAX_XNODES - 1; i++) {
#pragma unroll
        for (int e = 0; e < E; e++) res[i][e] = 0;
    }
#pragma unroll
    for (int i = 0; i < RFX_MAX_XNODES; i++) {
        if (i >= X.nops) break; // wave-uniform
        const PlanXNode n = X.ops[i];
        u64 l[E], r[E];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            u64(&o)[E] = side ? r : l;
            const int kind = side ? n.r_kind : n.l_kind, idx = side ? n.r_idx : n.l_idx;
            const u64 atom = side ? n.r_atom : n.l_atom;
            if (kind == RFX_XK_COL) sel_col_sw<NC, E>(o, v, idx);
            else if (kind == RFX_XK_ATOM) {
#pragma unroll
                for (int e = 0; e < E; e++) o[e] = atom;
            } else { // an earlier result: i static, idx < i
#pragma unroll
                for (int e = 0; e < E; e++) o[e] = (idx == 0) ? res[0][e] : ((idx == 1) ? res[1][e] : res[2][e]);
            }
        }
        if (i == X.nops - 1) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, l[e], r[e]);
        } else if (i < RFX_MAX_XNODES - 1) {
#pragma unroll
            for (int e = 0; e < E; e++) res[i][e] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, l[e], r[e]);
        }
    }
}

// Predicate / aggregate descriptors copied out of the kernarg segment ONCE per kernel into SGPRs (indexing the by-value
// Plan with a runtime loop counter inside the tile loop makes every field a dependent s_load + s_waitcnt per tile).
//
// A comparison is evaluated as TWO lane masks per row,  lt = "x sorts before y"  and  eq = "x equals y"  under the
// reference's total order (core/ops.h:76-123: NaN lowest, NaN == NaN, -0.0 == 0.0, i64 plain signed), and the operator
// only decides which of {lt, eq, gt = !(lt|eq)} it keeps -- three wave-uniform booleans, combined with s_and/s_or on
// the masks.  So the op costs no per-row branch and no per-op code copy.
enum { PF_F64DOM = 1, PF_LCVT = 2, PF_RCVT = 4, PF_CNAN = 8, PF_RCOL = 16, PF_KEEP_LT = 256, PF_KEEP_EQ = 512, PF_KEEP_GT = 1024, PF_MORE = 2048, PF_DEPTH_SHIFT = 12 /* 2 bits */, PF_CLOSE_SHIFT = 14 /* 2 bits */ };
struct PredR {
    int col, rhs_col, flags;
    u64 rhs;
};
template <int NP>
struct PredSet {
    int npred;
    bool is_and;
    bool grouped; // some predicate has PF_MORE: a two-level tree (parentheses of the opposite operator)
    bool deep;    // the predicates are the leaves of a deeper tree (PlanPred::tree: depth and closing parentheses per leaf)
    PredR p[NP > 0 ? NP : 1];
};
__device__ __forceinline__ int pred_keep_bits(int op) {
    switch (op) {
        case RFX_EQ: return PF_KEEP_EQ;
        case RFX_NE: return PF_KEEP_LT | PF_KEEP_GT;
        case RFX_LT: return PF_KEEP_LT;
        case RFX_GT: return PF_KEEP_GT;
        case RFX_LE: return PF_KEEP_LT | PF_KEEP_EQ;
        default: return PF_KEEP_GT | PF_KEEP_EQ;
    }
}
template <int NP>
__device__ __forceinline__ void predset_load(const Plan &P, PredSet<NP> &S) {
    S.npred = P.npred;
    S.is_and = (P.logic == RFX_AND);
    S.grouped = false;
    S.deep = false;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const PlanPred q = P.preds[i];
        S.p[i].col = q.col;
        S.p[i].rhs_col = q.rhs_col;
        S.p[i].rhs = q.rhs_bits;
        int f = pred_keep_bits(q.op);
        if (q.dom_f64) f |= PF_F64DOM;
        if (q.lhs_cvt) f |= PF_LCVT;
        if (q.rhs_cvt) f |= PF_RCVT;
        if (q.rhs_col >= 0) f |= PF_RCOL;
        else if (q.dom_f64 && rfx_isnan_bits(q.rhs_bits)) f |= PF_CNAN;
        if (i < P.npred && q.more) {
            f |= PF_MORE;
            S.grouped = true;
        }
        if (NP >= 3 && i < P.npred && q.tree) {
            f |= ((q.tree & 3) << PF_DEPTH_SHIFT) | (((q.tree >> 4) & 3) << PF_CLOSE_SHIFT);
            S.deep = true;
        }
        S.p[i].flags = f;
    }
}

// lt / eq masks of one predicate over the rows of one column tile `x`
template <int NC, int E>
__device__ __forceinline__ void pred_lt_eq(const PredR &pr, const u64 (&x)[E], const u64 (&v)[NC][E], bool (&lt)[E], bool (&eq)[E]) {
    const int f = pr.flags;
    if (!(f & (PF_LCVT | PF_RCOL | PF_CNAN))) {
        if (f & PF_F64DOM) {
            // atom is not NaN: "x sorts before c" == !(x >= c) (true for NaN x), one v_cmp_nge_f64; eq is IEEE ==
            const double c = rfx_as_f64(pr.rhs);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const double d = rfx_as_f64(x[e]);
                lt[e] = !(d >= c);
                eq[e] = (d == c);
            }
        } else {
            const i64 c = (i64)pr.rhs;
#pragma unroll
            for (int e = 0; e < E; e++) {
                lt[e] = (i64)x[e] < c;
                eq[e] = (i64)x[e] == c;
            }
        }
        return;
    }
    // general form: optional i64 -> f64 promotion (null -> NaN), rhs column, NaN atom
    u64 y[E];
    if (f & PF_RCOL) {
        sel_col<NC, E>(y, v, pr.rhs_col);
        if (f & PF_RCVT) {
#pragma unroll
            for (int e = 0; e < E; e++) y[e] = rfx_i64_to_f64_bits(y[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) y[e] = pr.rhs;
    }
    if (f & PF_F64DOM) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const u64 xe = (f & PF_LCVT) ? rfx_i64_to_f64_bits(x[e]) : x[e];
            lt[e] = rfx_ltf64(xe, y[e]);
            eq[e] = rfx_eqf64(xe, y[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) {
            lt[e] = (i64)x[e] < (i64)y[e];
            eq[e] = (i64)x[e] == (i64)y[e];
        }
    }
}

// The general form: the comparisons are the leaves of an and / or tree, in order; leaf p sits `depth` parentheses deep (level 0 = the
// query's own operator, every level below the opposite of the one above -- the same operator nested in itself is flattened by the
// host) and `close` parentheses end after it.  One accumulator mask per level; depth and close are wave-uniform plan constants, so
// every branch below is a scalar branch (and folds away in a kernel compiled for the plan).  Four levels: what core/logic.c's
// recursion reaches with eight comparisons in practice; deeper trees take the mask path.
template <int NC, int E, int NP>
__device__ __forceinline__ void eval_sel_tree(const PredSet<NP> &S, const u64 (&v)[NC][E], const bool (&valid)[E], bool (&sel)[E]) {
    const bool and0 = S.is_and; // level l folds with AND iff (l even) == and0; the identity of AND is true, of OR false
    bool a1[E], a2[E], a3[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        sel[e] = and0;
        a1[e] = a2[e] = a3[e] = false;
    }
    int cur = 0; // deepest open level (wave-uniform)
#pragma unroll
    for (int p = 0; p < NP; p++) {
        if (p < S.npred) {
            const int f = S.p[p].flags, d = (f >> PF_DEPTH_SHIFT) & 3, k = (f >> PF_CLOSE_SHIFT) & 3;
            bool pm[E];
#pragma unroll
            for (int e = 0; e < E; e++) pm[e] = false;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (S.p[p].col == c) {
                    bool lt[E], eq[E];
                    pred_lt_eq<NC, E>(S.p[p], v[c], v, lt, eq);
                    const bool klt = (f & PF_KEEP_LT) != 0, keq = (f & PF_KEEP_EQ) != 0, kgt = (f & PF_KEEP_GT) != 0;
#pragma unroll
                    for (int e = 0; e < E; e++) pm[e] = (lt[e] && klt) || (eq[e] && keq) || (!(lt[e] || eq[e]) && kgt);
                }
            }
            // parentheses that open before this leaf start at their operator's identity
            if (d >= 1 && cur < 1) {
#pragma unroll
                for (int e = 0; e < E; e++) a1[e] = !and0;
            }
            if (d >= 2 && cur < 2) {
#pragma unroll
                for (int e = 0; e < E; e++) a2[e] = and0;
            }
            if (d >= 3 && cur < 3) {
#pragma unroll
                for (int e = 0; e < E; e++) a3[e] = !and0;
            }
            cur = d;
            if (d == 0) {
#pragma unroll
                for (int e = 0; e < E; e++) sel[e] = and0 ? (sel[e] && pm[e]) : (sel[e] || pm[e]);
            } else if (d == 1) {
#pragma unroll
                for (int e = 0; e < E; e++) a1[e] = !and0 ? (a1[e] && pm[e]) : (a1[e] || pm[e]);
            } else if (d == 2) {
#pragma unroll
                for (int e = 0; e < E; e++) a2[e] = and0 ? (a2[e] && pm[e]) : (a2[e] || pm[e]);
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) a3[e] = !and0 ? (a3[e] && pm[e]) : (a3[e] || pm[e]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j < k) {
                    if (cur == 3) {
#pragma unroll
                        for (int e = 0; e < E; e++) a2[e] = and0 ? (a2[e] && a3[e]) : (a2[e] || a3[e]);
                    } else if (cur == 2) {
#pragma unroll
                        for (int e = 0; e < E; e++) a1[e] = !and0 ? (a1[e] && a2[e]) : (a1[e] || a2[e]);
                    } else if (cur == 1) {
#pragma unroll
                        for (int e = 0; e < E; e++) sel[e] = and0 ? (sel[e] && a1[e]) : (sel[e] || a1[e]);
                    }
                    cur--;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; e++) sel[e] = sel[e] && valid[e];
}

// Evaluate all predicates on a register tile: sel[e] = valid[e] && combine(pred_p(row e)).  Each predicate is evaluated straight on
// the tile of the column it reads (the static `col == c` chain: no register copies).  Flat lists fold into sel directly; a two-level
// tree (PF_MORE: "the next predicate is in the same parenthesis") folds each parenthesis with the OPPOSITE operator into `par`, and
// `par` into sel when the parenthesis closes -- lane masks in SGPRs, wave-uniform selects, no per-row branch.
template <int NC, int E, int NP>
__device__ __forceinline__ void eval_sel(const PredSet<NP> &S, const u64 (&v)[NC][E], const bool (&valid)[E], bool (&sel)[E]) {
    if (NP == 0 || S.npred == 0) {
#pragma unroll
        for (int e = 0; e < E; e++) sel[e] = valid[e];
        return;
    }
    const bool is_and = S.is_and, grouped = S.grouped;
    if constexpr (NP >= 3) {
        if (S.deep) { // an arbitrarily nested tree (logic_map, core/logic.c:89-260): still ONE pass, still lane masks only
            eval_sel_tree<NC, E, NP>(S, v, valid, sel);
            return;
        }
    }
    bool par[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        sel[e] = is_and;
        par[e] = !is_and;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        if (p < S.npred) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (S.p[p].col == c) {
                    bool lt[E], eq[E];
                    pred_lt_eq<NC, E>(S.p[p], v[c], v, lt, eq);
                    const bool klt = (S.p[p].flags & PF_KEEP_LT) != 0, keq = (S.p[p].flags & PF_KEEP_EQ) != 0, kgt = (S.p[p].flags & PF_KEEP_GT) != 0;
#pragma unroll
                    for (int e = 0; e < E; e++) {
                        const bool pm = (lt[e] && klt) || (eq[e] && keq) || (!(lt[e] || eq[e]) && kgt);
                        if (!grouped) sel[e] = is_and ? (sel[e] && pm) : (sel[e] || pm);
                        else par[e] = is_and ? (par[e] || pm) : (par[e] && pm);
                    }
                }
            }
            if (grouped && !(S.p[p].flags & PF_MORE)) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    sel[e] = is_and ? (sel[e] && par[e]) : (sel[e] || par[e]);
                    par[e] = !is_and;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; e++) sel[e] = sel[e] && valid[e];
}

// Compatibility form for the scatter kernels: bit e of the result = row e of this lane is selected.
template <int NC, int E, int NP>
__device__ __forceinline__ unsigned eval_preds(const PredSet<NP> &S, const u64 (&v)[NC][E], unsigned valid_bits) {
    bool valid[E], sel[E];
#pragma unroll
    for (int e = 0; e < E; e++) valid[e] = (valid_bits >> e) & 1u;
    eval_sel<NC, E, NP>(S, v, valid, sel);
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) m |= (unsigned)sel[e] << e;
    return m;
}

// wave-uniform population count of a lane predicate (s_bcnt1_i32_b64 of the compare mask)
__device__ __forceinline__ int wave_count(bool p) { return __popcll(__ballot(p)); }

// In-loop accumulator state.  f64 MIN/MAX run on raw doubles (init +inf / -inf) and are converted to the
// order-preserving integer image only once, before the cross-lane reduction (acc_finish_lane).
__device__ __forceinline__ void acc_init_loop(Acc &a, int kind, int f64) {
    acc_init(a, kind);
    if (f64 && kind == RFX_AGG_MIN) a.v = RFX_PINF_BITS;
    if (f64 && kind == RFX_AGG_MAX) a.v = 0xFFF0000000000000ULL;
}
__device__ __forceinline__ void acc_finish_lane(Acc &a, int kind, int f64) {
    if (f64 && (kind == RFX_AGG_MIN || kind == RFX_AGG_MAX)) a.v = (u64)rfx_f64_to_ord(a.v);
}

// Fold the selected rows of one column tile into one accumulator.  Counts (a.c for every kind but FIRST) are kept
// WAVE-UNIFORM: they hold the wave's total, not the lane's -- the reduction takes them from lane 0 only.
// Scalar rules: FOLD_ADD* skip nulls (core/ops.h:156-158), MIN*/MAX* skip nulls (:179-187), CNT* (:148-152).
template <int E>
__device__ __forceinline__ void acc_update(Acc &a, int kind, int f64, const u64 (&x)[E], const bool (&sel)[E], i64 row_of_e0, int jstride) {
    if (kind == RFX_AGG_SUM || kind == RFX_AGG_AVG) {
        int c = 0;
        if (f64) {
            double s = rfx_as_f64(a.v);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const double d = rfx_as_f64(x[e]);
                const bool ok = sel[e] && (d == d);
                s += ok ? d : 0.0;
                c += wave_count(ok);
            }
            a.v = rfx_as_u64(s);
        } else {
            u64 s = a.v;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const bool ok = sel[e] && (i64)x[e] != RFX_NULL_I64_D;
                s += ok ? x[e] : 0ULL;
                c += wave_count(ok);
            }
            a.v = s;
        }
        a.c += c;
    } else if (kind == RFX_AGG_MIN || kind == RFX_AGG_MAX) {
        const bool is_min = (kind == RFX_AGG_MIN);
        int c = 0;
        if (f64) {
            double m = rfx_as_f64(a.v);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const double d = rfx_as_f64(x[e]);
                const bool ok = sel[e] && (d == d);
                const bool better = ok && (is_min ? (d < m) : (d > m));
                m = better ? d : m;
                c += wave_count(ok);
            }
            a.v = rfx_as_u64(m);
        } else {
            i64 m = (i64)a.v;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const i64 d = (i64)x[e];
                const bool ok = sel[e] && d != RFX_NULL_I64_D;
                const bool better = ok && (is_min ? (d < m) : (d > m));
                m = better ? d : m;
                c += wave_count(ok);
            }
            a.v = (u64)m;
        }
        a.c += c;
    } else if (kind == RFX_AGG_FIRST) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const i64 row = row_of_e0 + (i64)(e >> 1) * jstride + (e & 1);
            if (sel[e] && row < a.c) {
                a.c = row;
                a.v = x[e];
            }
        }
    }
}

struct AggR {
    int col, f64, kind;
};
template <int NC, int NA, int E, int NP, int NX = 0, bool DEEP = false>
__device__ __forceinline__ void fold_tile(const PredSet<NP> &S, const AggR (&ag)[NA], const u64 (&v)[NC][E], const bool (&valid)[E], Acc (&acc)[NA],
                                          i64 &nsel, i64 row_of_e0, int jstride, const PlanExpr *xs = nullptr) {
    bool sel[E];
    eval_sel<NC, E, NP>(S, v, valid, sel);
    int c = 0;
#pragma unroll
    for (int e = 0; e < E; e++) c += wave_count(sel[e]);
    nsel += c;
#pragma unroll
    for (int a = 0; a < NA; a++) {
        if (ag[a].kind == RFX_AGG_COUNT) acc[a].c += c;
    }
#pragma unroll
    for (int col = 0; col < NC; col++) {
#pragma unroll
        for (int a = 0; a < NA; a++) {
            if (ag[a].kind >= 0 && ag[a].kind != RFX_AGG_COUNT && ag[a].col == col) acc_update<E>(acc[a], ag[a].kind, ag[a].f64, v[col], sel, row_of_e0, jstride);
        }
    }
    if (NX > 0) {
#pragma unroll
        for (int i = 0; i < NX; i++) {
            bool used = false;
#pragma unroll
            for (int a = 0; a < NA; a++) used |= (ag[a].kind >= 0 && ag[a].col == RFX_XCOL + i);
            if (!used) continue; // wave-uniform
            u64 x[E];
            if (DEEP) expr_input_deep<NC, E>(x, v, xs[i]);
            else expr_input<NC, E>(x, v, xs[i]);
#pragma unroll
            for (int a = 0; a < NA; a++) {
                if (ag[a].kind >= 0 && ag[a].kind != RFX_AGG_COUNT && ag[a].col == RFX_XCOL + i)
                    acc_update<E>(acc[a], ag[a].kind, ag[a].f64, x, sel, row_of_e0, jstride);
            }
        }
    }
}

// Workgroup partial layout in the workspace: ws[(block * (NA + 1) + a)] ; slot NA = selected-row count.
// D: the plan's DESCRIPTORS (which columns, predicates, aggregates, expressions), R: its run-time values (column pointers, row
// counts, the predicates' atoms).  The prebuilt kernels pass one plan for both; a kernel compiled at run time for one plan
// (rfx_rtc.hip) passes a constexpr D, which turns every descriptor test below into a constant.
template <int NC, int NA, int U, int NP, int NX, bool DEEP>
__device__ __forceinline__ void filter_aggr_body(const Plan &D, const Plan &R, Acc *__restrict__ ws) {
    const Plan &P = D;
    constexpr int E = 2 * U;
    constexpr int TILE = RFX_BLOCK * E;     // rows per workgroup per iteration
    constexpr int JSTRIDE = RFX_BLOCK * 2;  // row distance between the U loads of one lane
    const int tid = threadIdx.x;
    PredSet<NP> S;
    predset_load<NP>(P, S);
#pragma unroll
    for (int i = 0; i < NP; i++) S.p[i].rhs = R.preds[i].rhs_bits;
    AggR ag[NA];
    Acc acc[NA];
    i64 nsel = 0; // wave-uniform
#pragma unroll
    for (int a = 0; a < NA; a++) {
        ag[a].col = P.aggs[a].col;
        ag[a].f64 = P.aggs[a].f64;
        ag[a].kind = P.aggs[a].kind;
        acc_init_loop(acc[a], ag[a].kind, ag[a].f64);
    }
    const u64 *cols[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) cols[c] = R.cols[c];
    const i64 nrows = R.nrows, row0 = R.row0;
    PlanExpr xs[NX > 0 ? NX : 1]; // expression descriptors, hoisted like the others (static indices only)
#pragma unroll
    for (int i = 0; i < (NX > 0 ? NX : 1); i++) xs[i] = P.xs[i];

    bool all[E];
#pragma unroll
    for (int e = 0; e < E; e++) all[e] = true;

    const i64 nfull = nrows / TILE;
    for (i64 t = blockIdx.x; t < nfull; t += gridDim.x) {
        const i64 base = t * TILE + tid * 2;
        u64 v[NC][E];
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int j = 0; j < U; j++) {
                u64x2 q = rfx_ld2(cols[c] + base + (i64)j * JSTRIDE);
                v[c][2 * j] = q.x;
                v[c][2 * j + 1] = q.y;
            }
        }
        fold_tile<NC, NA, E, NP, NX, DEEP>(S, ag, v, all, acc, nsel, row0 + base, JSTRIDE, xs);
    }
    // ragged tail: one workgroup, guarded element loads
    const i64 tail0 = nfull * TILE;
    if (tail0 < nrows && blockIdx.x == (unsigned)(nfull % gridDim.x)) {
        const i64 base = tail0 + tid * 2;
        u64 v[NC][E];
        bool valid[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const i64 row = base + (i64)(e >> 1) * JSTRIDE + (e & 1);
            valid[e] = row < nrows;
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][e] = valid[e] ? cols[c][row] : 0ULL;
        }
        fold_tile<NC, NA, E, NP, NX, DEEP>(S, ag, v, valid, acc, nsel, row0 + base, JSTRIDE, xs);
    }

    // wave reduction (64 lanes), then across the 4 waves through LDS.  Counts are wave-uniform already: keep lane 0's.
    __shared__ Acc lds[RFX_BLOCK / RFX_WAVE][NA + 1];
    const int wave = tid / RFX_WAVE, lane = tid % RFX_WAVE;
#pragma unroll
    for (int a = 0; a < NA; a++) {
        acc_finish_lane(acc[a], ag[a].kind, ag[a].f64);
        if (ag[a].kind != RFX_AGG_FIRST && lane != 0) acc[a].c = 0;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
        for (int a = 0; a < NA; a++) {
            Acc o = acc_shfl_xor(acc[a], s);
            acc_combine(acc[a], o, ag[a].kind, ag[a].f64);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NA; a++) lds[wave][a] = acc[a];
        Acc n;
        n.v = 0;
        n.c = nsel;
        lds[wave][NA] = n;
    }
    __syncthreads();
    if (tid <= NA) {
        const int kind = (tid == NA) ? RFX_AGG_COUNT : P.aggs[tid].kind;
        const int f64 = (tid == NA) ? 0 : P.aggs[tid].f64;
        Acc r = lds[0][tid];
        for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) acc_combine(r, lds[w][tid], kind, f64);
        ws[(size_t)blockIdx.x * (NA + 1) + tid] = r;
    }
}

template <int NC, int NA, int U, int NP, int NX = 0, bool DEEP = false>
__global__ __launch_bounds__(RFX_BLOCK) void k_filter_aggr(const Plan P, Acc *__restrict__ ws) {
    filter_aggr_body<NC, NA, U, NP, NX, DEEP>(P, P, ws);
}

#ifndef __HIPCC_RTC__
// one launcher per distinct-column count, defined in rfx_scalar_nc.hip compiled with -DRFX_NC=<n>
#define RFX_DECL_LAUNCH(n) int rfx_launch_filter_aggr_nc##n(rfx_ctx *c, const Plan &P, int grid, Acc *ws, int *na_stride);
RFX_DECL_LAUNCH(1) RFX_DECL_LAUNCH(2) RFX_DECL_LAUNCH(3) RFX_DECL_LAUNCH(4)
RFX_DECL_LAUNCH(5) RFX_DECL_LAUNCH(6) RFX_DECL_LAUNCH(7) RFX_DECL_LAUNCH(8)
#endif
