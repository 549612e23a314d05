// rfx_scalar_kernel.hpp -- internal: the fused predicate -> aggregate register-tile machinery (K1/K5),
// shared by rfx_scalar.hip (final fold, masks) and rfx_scalar_nc.hip (one translation unit per column count so
// the 8 x 4 template instantiations compile in parallel).
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
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (col == c) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = v[c][e];
        }
    }
}

// Evaluate all predicates on a register tile.  Bit e of the result = row e of this lane is selected.
template <int NC, int E>
__device__ __forceinline__ unsigned eval_preds(const Plan &P, const u64 (&v)[NC][E], unsigned valid) {
    if (P.npred == 0) return valid;
    unsigned m = (P.logic == RFX_AND) ? valid : 0u;
    for (int p = 0; p < P.npred; p++) {
        const PlanPred &pr = P.preds[p];
        u64 x[E];
        sel_col<NC, E>(x, v, pr.col);
        if (pr.lhs_cvt) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = rfx_i64_to_f64_bits(x[e]);
        }
        unsigned pm = 0;
        if (pr.rhs_col < 0) {
            const u64 r = pr.rhs_bits;
            if (pr.dom_f64) {
#pragma unroll
                for (int e = 0; e < E; e++) pm |= (unsigned)rfx_cmp_f64(pr.op, x[e], r) << e;
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) pm |= (unsigned)rfx_cmp_i64(pr.op, (i64)x[e], (i64)r) << e;
            }
        } else {
            u64 y[E];
            sel_col<NC, E>(y, v, pr.rhs_col);
            if (pr.rhs_cvt) {
#pragma unroll
                for (int e = 0; e < E; e++) y[e] = rfx_i64_to_f64_bits(y[e]);
            }
            if (pr.dom_f64) {
#pragma unroll
                for (int e = 0; e < E; e++) pm |= (unsigned)rfx_cmp_f64(pr.op, x[e], y[e]) << e;
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) pm |= (unsigned)rfx_cmp_i64(pr.op, (i64)x[e], (i64)y[e]) << e;
            }
        }
        m = (P.logic == RFX_AND) ? (m & pm) : (m | pm);
    }
    return m & valid;
}

// Fold the selected rows of a register tile into one accumulator.
// Scalar rules: FOLD_ADD* skip nulls (core/ops.h:156-158), MIN*/MAX* skip nulls (:179-187), CNT* (:148-152).
template <int E>
__device__ __forceinline__ void acc_update(Acc &a, int kind, int f64, const u64 (&x)[E], unsigned m, i64 row_of_e0, int U_stride) {
    switch (kind) {
        case RFX_AGG_SUM:
        case RFX_AGG_AVG:
            if (f64) {
                double s = rfx_as_f64(a.v);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    bool ok = ((m >> e) & 1u) && !rfx_isnan_bits(x[e]);
                    s += ok ? rfx_as_f64(x[e]) : 0.0;
                    a.c += ok;
                }
                a.v = rfx_as_u64(s);
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    bool ok = ((m >> e) & 1u) && (i64)x[e] != RFX_NULL_I64_D;
                    a.v += ok ? x[e] : 0ULL;
                    a.c += ok;
                }
            }
            break;
        case RFX_AGG_MIN:
#pragma unroll
            for (int e = 0; e < E; e++) {
                bool ok = ((m >> e) & 1u) && (f64 ? !rfx_isnan_bits(x[e]) : (i64)x[e] != RFX_NULL_I64_D);
                i64 o = f64 ? rfx_f64_to_ord(x[e]) : (i64)x[e];
                a.v = (ok && o < (i64)a.v) ? (u64)o : a.v;
                a.c += ok;
            }
            break;
        case RFX_AGG_MAX:
#pragma unroll
            for (int e = 0; e < E; e++) {
                bool ok = ((m >> e) & 1u) && (f64 ? !rfx_isnan_bits(x[e]) : (i64)x[e] != RFX_NULL_I64_D);
                i64 o = f64 ? rfx_f64_to_ord(x[e]) : (i64)x[e];
                a.v = (ok && o > (i64)a.v) ? (u64)o : a.v;
                a.c += ok;
            }
            break;
        case RFX_AGG_COUNT:
            a.c += __popc(m);
            break;
        case RFX_AGG_FIRST:
#pragma unroll
            for (int e = 0; e < E; e++) {
                i64 row = row_of_e0 + (i64)(e >> 1) * U_stride + (e & 1);
                if (((m >> e) & 1u) && row < a.c) {
                    a.c = row;
                    a.v = x[e];
                }
            }
            break;
        default:
            break;
    }
}

// Workgroup partial layout in the workspace: ws[(block * (NA + 1) + a)] ; slot NA = selected-row count.
template <int NC, int NA, int U>
__global__ __launch_bounds__(RFX_BLOCK) void k_filter_aggr(const Plan P, Acc *__restrict__ ws) {
    constexpr int E = 2 * U;
    constexpr int TILE = RFX_BLOCK * E;     // rows per workgroup per iteration
    constexpr int JSTRIDE = RFX_BLOCK * 2;  // row distance between the U loads of one lane
    const int tid = threadIdx.x;
    Acc acc[NA];
    Acc nsel;
    nsel.v = 0;
    nsel.c = 0;
#pragma unroll
    for (int a = 0; a < NA; a++) acc_init(acc[a], P.aggs[a].kind);

    const i64 nfull = P.nrows / TILE;
    for (i64 t = blockIdx.x; t < nfull; t += gridDim.x) {
        const i64 base = t * TILE + tid * 2;
        u64 v[NC][E];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const u64 *p = P.cols[c] + base;
#pragma unroll
            for (int j = 0; j < U; j++) {
                u64x2 q = rfx_ld2(p + (i64)j * JSTRIDE);
                v[c][2 * j] = q.x;
                v[c][2 * j + 1] = q.y;
            }
        }
        const unsigned m = eval_preds<NC, E>(P, v, (1u << E) - 1u);
        nsel.c += __popc(m);
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const PlanAgg ag = P.aggs[a];
            if (ag.kind < 0) continue;
            u64 x[E];
            if (ag.col >= 0) sel_col<NC, E>(x, v, ag.col);
            acc_update<E>(acc[a], ag.kind, ag.f64, x, m, P.row0 + base, JSTRIDE);
        }
    }
    // ragged tail: one workgroup, guarded element loads
    const i64 tail0 = nfull * TILE;
    if (tail0 < P.nrows && blockIdx.x == (unsigned)(nfull % gridDim.x)) {
        const i64 base = tail0 + tid * 2;
        u64 v[NC][E];
        unsigned valid = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            i64 row = base + (i64)(e >> 1) * JSTRIDE + (e & 1);
            bool in = row < P.nrows;
            valid |= (unsigned)in << e;
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
        }
        const unsigned m = eval_preds<NC, E>(P, v, valid);
        nsel.c += __popc(m);
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const PlanAgg ag = P.aggs[a];
            if (ag.kind < 0) continue;
            u64 x[E];
            if (ag.col >= 0) sel_col<NC, E>(x, v, ag.col);
            acc_update<E>(acc[a], ag.kind, ag.f64, x, m, P.row0 + base, JSTRIDE);
        }
    }

    // wave reduction (64 lanes), then across the 4 waves through LDS
    __shared__ Acc lds[RFX_BLOCK / RFX_WAVE][NA + 1];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
        for (int a = 0; a < NA; a++) {
            Acc o = acc_shfl_xor(acc[a], s);
            acc_combine(acc[a], o, P.aggs[a].kind, P.aggs[a].f64);
        }
        nsel.c += (i64)rfx_shfl_xor_u64((u64)nsel.c, s);
    }
    const int wave = tid / RFX_WAVE, lane = tid % RFX_WAVE;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NA; a++) lds[wave][a] = acc[a];
        lds[wave][NA] = nsel;
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


// one launcher per distinct-column count, defined in rfx_scalar_nc.hip compiled with -DRFX_NC=<n>
#define RFX_DECL_LAUNCH(n) int rfx_launch_filter_aggr_nc##n(rfx_ctx *c, const Plan &P, int grid, Acc *ws, int *na_stride);
RFX_DECL_LAUNCH(1) RFX_DECL_LAUNCH(2) RFX_DECL_LAUNCH(3) RFX_DECL_LAUNCH(4)
RFX_DECL_LAUNCH(5) RFX_DECL_LAUNCH(6) RFX_DECL_LAUNCH(7) RFX_DECL_LAUNCH(8)
