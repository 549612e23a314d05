// rfx_group_few_rtc.hpp -- K7 + K10 for a HANDFUL of groups, compiled at run time for ONE plan (hiprtc; rfx_rtc.hip).
//
// Why: with six groups (the TPC-H Q1 shape: 7 columns, 8 aggregates, 1e9 rows) the LDS-table kernel issues twelve LDS atomics per
// row that all land on six addresses -- 1.2e10 same-address atomics, 30 of the query's 34 ms, however the descriptors are decoded
// (fixing every descriptor field with __builtin_assume bought 6 %).  Here every lane keeps its own accumulator per (aggregate, group)
// in REGISTERS and the row is folded into all groups with selects -- no atomics in the row loop; lanes meet once, at the end
// (LDS, then one merge per workgroup into the global tables, as the LDS-table kernel ends).  Registers for NA x NG accumulators
// only exist when NA and NG are compile-time constants, and the per-(aggregate, group) code only stays small when the kinds are:
// hence one kernel per plan, generated text = the defines below + this file.
//
// Generated before the #include:  FEW_NC (plan columns)  FEW_NA (aggregates)  FEW_NG (slots of the dense table = key range)
//   FEW_NPT (predicate descriptor sets: 0 / 2 / RFX_MAX_PREDS)  FEW_U (16-byte loads per lane, column and tile)
//   RTC_PLAN  the plan's DESCRIPTOR part as a braced initialiser of `Plan` (column pointers null, row counts 0: those and the
//             predicates' right-hand atoms come from the kernel argument at run time, so one kernel serves every constant of a
//             filter; a NaN atom is part of the signature because it selects the comparison's code path)
//   FEW_KEY_IDX, FEW_NKEYS, FEW_KIDX {..}  the key column(s) of the dense slot
// A local constexpr Plan makes every descriptor a constant after unrolling.  (__builtin_assume on the kernel argument's fields, and
// assignments into a local copy of it, were tried first: neither reaches the loads, the code stays as generic as the prebuilt kernel.)
// Semantics: group_apply / group_merge_cell (rfx_group_common.hpp), i.e. core/aggr.c's grouped rules, cell by cell.
#include "rfx_group_common.hpp"


// one row folded into one group's register cells; `hit` = the row is selected and belongs to the group
// (count_nulls: wave-uniform -- some lane of the wave holds a selected null in this tile; without one the null counter of an i64 sum is left alone.
//  COUNT is not folded here: every group's selected rows are counted once per row for all aggregates, k_group_few's rsel[])
__device__ __forceinline__ void few_apply(u64 &acc, unsigned &cnt, int kind, int f64, int skip, u64 x, bool hit, bool count_nulls) {
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) {
                const bool on = hit && (!skip || !rfx_isnan_bits(x));
                acc = rfx_as_u64(rfx_as_f64(acc) + (on ? rfx_as_f64(x) : 0.0));
            } else {
                const bool null = (i64)x == RFX_NULL_I64_D;
                if (count_nulls) cnt += (hit && null) ? 1u : 0u;
                acc += (hit && !null) ? x : 0ULL;
            }
            break;
        case RFX_AGG_AVG: {
            const bool on = hit && (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D));
            acc = rfx_as_u64(rfx_as_f64(acc) + (on ? (f64 ? rfx_as_f64(x) : (double)(i64)x) : 0.0));
            cnt += on ? 1u : 0u;
            break;
        }
        case RFX_AGG_MIN: {
            const bool on = hit && (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D));
            const i64 y = f64 ? rfx_f64_to_ord(x) : (i64)x;
            acc = (on && y < (i64)acc) ? (u64)y : acc;
            break;
        }
        case RFX_AGG_MAX: {
            const bool on = hit && (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D));
            const i64 y = f64 ? rfx_f64_to_ord(x) : (i64)x;
            acc = (on && y > (i64)acc) ? (u64)y : acc;
            break;
        }
        default: // COUNT: rsel[] (see above); FIRST is resolved at emit time from d_first
            break;
    }
}
// a lane's cells into the workgroup's LDS cells
__device__ __forceinline__ void few_merge_lds(u64 *lacc, unsigned *lcnt, int kind, int f64, u64 a, unsigned c) {
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) unsafeAtomicAdd((double *)lacc, rfx_as_f64(a));
            else {
                atomicAdd((unsigned long long *)lacc, (unsigned long long)a);
                if (c) atomicAdd(lcnt, c);
            }
            break;
        case RFX_AGG_AVG:
            if (c) {
                unsafeAtomicAdd((double *)lacc, rfx_as_f64(a));
                atomicAdd(lcnt, c);
            }
            break;
        case RFX_AGG_MIN: atomicMin((long long *)lacc, (i64)a); break;
        case RFX_AGG_MAX: atomicMax((long long *)lacc, (i64)a); break;
        case RFX_AGG_COUNT: atomicAdd((unsigned long long *)lacc, (unsigned long long)a); break;
        default: break;
    }
}

extern "C" __global__ __launch_bounds__(RFX_BLOCK) void k_group_few(const Plan P0, const GroupArgs G) {
    constexpr Plan P = RTC_PLAN; // descriptors only
    constexpr int KIDX[RFX_MAX_KEYS] = FEW_KIDX;
    constexpr int NC = FEW_NC, NA = FEW_NA, NG = FEW_NG, U = FEW_U, E = 2 * U;
    constexpr int TILE = RFX_BLOCK * E, JSTRIDE = RFX_BLOCK * 2;
    __shared__ u64 lacc[NA][NG];
    __shared__ unsigned lcnt[NA][NG];
    __shared__ unsigned lfirst[NG];
    const int tid = threadIdx.x;
    PredSet<FEW_NPT> S;
    predset_load<FEW_NPT>(P, S);
#pragma unroll
    for (int i = 0; i < FEW_NPT; i++) S.p[i].rhs = P0.preds[i].rhs_bits;
    u64 racc[NA][NG];
    unsigned rcnt[NA][NG], rfirst[NG];
#pragma unroll
    for (int a = 0; a < NA; a++) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
            racc[a][g] = acc_identity(P.aggs[a].kind, P.aggs[a].f64);
            rcnt[a][g] = 0;
        }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) rfirst[g] = 0xffffffffu;
    // Round 6: every group's selected rows, counted ONCE per row for all aggregates -- COUNT is that number, an average's divisor is that number less the
    // nulls of its own column, and those (rcnt[a][g] under the masked fma) are only counted in tiles where some lane of the wave holds a selected null:
    // three averages and a count beside five sums (the Q1 shape) spent 96 of ~420 vector instructions per tile on four identical counters
    unsigned rsel[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) rsel[g] = 0;
    if (tid < NG) {
        lfirst[tid] = 0xffffffffu;
#pragma unroll
        for (int a = 0; a < NA; a++) {
            lacc[a][tid] = acc_identity(P.aggs[a].kind, P.aggs[a].f64);
            lcnt[a][tid] = 0;
        }
    }
    const i64 nrows = P0.nrows;
    const i64 nfull = nrows / TILE;
    const i64 ntiles = nfull + ((nfull * TILE < nrows) ? 1 : 0);
    // Round 6: the NEXT tile's loads are issued before this tile is folded (FEW_PREFETCH, default on): with 48 register accumulators the kernel runs at two
    // waves per SIMD (210 VGPRs), i.e. 8 waves x 64 lanes x NC x 16 bytes = 57 KB in flight per CU -- 14.7 MB on the part, at the edge of what 6+ TB/s needs at
    // ~2 us of loaded latency (the Q1 shape read 4.9 TB/s); the second tile in flight costs NC x E x 2 registers and doubles that.
    auto load_tile = [&](const i64 t, u64 (&v)[NC][E], unsigned &valid) __attribute__((always_inline)) {
        const i64 base = t * TILE + tid * 2;
        if (t < nfull) {
            valid = (1u << E) - 1u;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const u64 *p = P0.cols[c] + base;
#pragma unroll
                for (int j = 0; j < U; j++) {
                    u64x2 q = rfx_ld2(p + (i64)j * JSTRIDE);
                    v[c][2 * j] = q.x;
                    v[c][2 * j + 1] = q.y;
                }
            }
        } else {
            valid = 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
                i64 row = base + (i64)(e >> 1) * JSTRIDE + (e & 1);
                bool in = row < nrows;
                valid |= (unsigned)in << e;
#pragma unroll
                for (int c = 0; c < NC; c++) v[c][e] = in ? P0.cols[c][row] : 0ULL;
            }
        }
    };
#ifndef FEW_PREFETCH
#define FEW_PREFETCH 1
#endif
    u64 vn[NC][E];
    unsigned validn = 0;
    if (FEW_PREFETCH && (i64)blockIdx.x < ntiles) load_tile(blockIdx.x, vn, validn);
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 base = t * TILE + tid * 2;
        u64 v[NC][E];
        unsigned valid;
        if (FEW_PREFETCH) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
#pragma unroll
                for (int e = 0; e < E; e++) v[c][e] = vn[c][e];
            }
            valid = validn;
            if (t + (i64)gridDim.x < ntiles) load_tile(t + gridDim.x, vn, validn);
        } else load_tile(t, v, valid);
        const unsigned m = eval_preds<NC, E, FEW_NPT>(S, v, valid);
        u64 key[E]; // slot in the dense table
        if (FEW_NKEYS <= 1) {
            sel_col_sw<NC, E>(key, v, FEW_KEY_IDX);
#pragma unroll
            for (int e = 0; e < E; e++) key[e] -= (u64)G.kmin;
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) key[e] = 0;
            bool out[E];
#pragma unroll
            for (int e = 0; e < E; e++) out[e] = false;
#pragma unroll
            for (int i = 0; i < FEW_NKEYS; i++) {
                u64 x[E];
                sel_col_sw<NC, E>(x, v, KIDX[i]);
                const u64 mn = G.kmn[i], mu = G.kmul[i], rg = G.krng[i];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    out[e] |= (x[e] - mn) >= rg;
                    key[e] += (x[e] - mn) * mu;
                }
            }
#pragma unroll
            for (int e = 0; e < E; e++) key[e] = out[e] ? ~0ULL : key[e];
        }
        {   // a selected row outside the agreed scope (a sampled scope missed its key): reported, the host runs the exact scope
            bool bad = false;
#pragma unroll
            for (int e = 0; e < E; e++) bad |= ((m >> e) & 1u) && key[e] >= (u64)NG;
            if (bad && G.oob) *(volatile unsigned *)G.oob = 1u;
        }
        double hitd[E][NG]; // 1.0: row e is selected and belongs to group g
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool sel = (m >> e) & 1u;
            const unsigned lrow = (unsigned)(base + (i64)(e >> 1) * JSTRIDE + (e & 1));
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const bool hit = sel && key[e] == (u64)g;
                hitd[e][g] = hit ? 1.0 : 0.0;
                rsel[g] += hit ? 1u : 0u;
                rfirst[g] = (hit && lrow < rfirst[g]) ? lrow : rfirst[g];
            }
        }
        bool special = false; // a selected row of this tile holds a value the masked fma had to zero out but the aggregate must see
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const int kind = P.aggs[a].kind, f64 = P.aggs[a].f64, skip = P.aggs[a].skipnull, col = P.aggs[a].col;
            u64 x[E];
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = 0;
            if (col >= RFX_XCOL) expr_input_deep_sw<NC, E>(x, v, P.xs[col - RFX_XCOL]);
            else if (col >= 0) sel_col_sw<NC, E>(x, v, col);
            if (FEW_FMA && (kind == RFX_AGG_AVG || (kind == RFX_AGG_SUM && f64))) {
                // f64 accumulation: acc += x * (1.0 if the row is this group's else 0.0) -- one v_fma_f64 per (aggregate, group) where a
                // select + add is three instructions.  x * 0.0 is only harmless for finite x, so nulls (NaN / null i64, skipped or
                // poisoning by the aggregate's rule) and infinities are zeroed here and -- rare -- added for real on a side path.
                bool selnul[E], anynul = false; // (an average's rcnt[a][g]: the group's selected NULLS of this column)
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool sel = (m >> e) & 1u;
                    const bool nul = f64 ? rfx_isnan_bits(x[e]) : ((i64)x[e] == RFX_NULL_I64_D);
                    const bool inf = f64 && (x[e] & 0x7FFFFFFFFFFFFFFFULL) == RFX_PINF_BITS;
                    const double xd = f64 ? rfx_as_f64(x[e]) : (double)(i64)x[e];
                    const bool real_add = sel && (inf || (nul && kind == RFX_AGG_SUM && !skip)); // must reach the accumulator as it is
                    const double xz = (nul || inf) ? 0.0 : xd;
                    selnul[e] = sel && nul;
                    anynul |= selnul[e];
#pragma unroll
                    for (int g = 0; g < NG; g++) racc[a][g] = rfx_as_u64(__builtin_fma(xz, hitd[e][g], rfx_as_f64(racc[a][g])));
                    special |= real_add;
                }
                if (kind == RFX_AGG_AVG && __builtin_amdgcn_ballot_w64(anynul) != 0) { // wave-uniform: skipped by the tiles without a selected null
#pragma unroll
                    for (int e = 0; e < E; e++) {
#pragma unroll
                        for (int g = 0; g < NG; g++) rcnt[a][g] += (selnul[e] && key[e] == (u64)g) ? 1u : 0u;
                    }
                }
            } else if (kind != RFX_AGG_COUNT) {
                bool anynul = false;
                if (kind == RFX_AGG_SUM && !f64) {
#pragma unroll
                    for (int e = 0; e < E; e++) anynul |= ((m >> e) & 1u) && (i64)x[e] == RFX_NULL_I64_D;
                }
                const bool count_nulls = __builtin_amdgcn_ballot_w64(anynul) != 0;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool sel = (m >> e) & 1u;
#pragma unroll
                    for (int g = 0; g < NG; g++) few_apply(racc[a][g], rcnt[a][g], kind, f64, skip, x[e], sel && key[e] == (u64)g, count_nulls);
                }
            }
        }
        if (FEW_FMA && special) { // divergent and rare (the wave skips it unless a lane holds an infinity or a poisoning NaN): add those for real
#pragma unroll
            for (int a = 0; a < NA; a++) {
                const int kind = P.aggs[a].kind, f64 = P.aggs[a].f64, skip = P.aggs[a].skipnull, col = P.aggs[a].col;
                if (!(f64 && (kind == RFX_AGG_AVG || kind == RFX_AGG_SUM))) continue; // i64 inputs have neither
                u64 x[E];
#pragma unroll
                for (int e = 0; e < E; e++) x[e] = 0;
                if (col >= RFX_XCOL) expr_input_deep_sw<NC, E>(x, v, P.xs[col - RFX_XCOL]);
                else if (col >= 0) sel_col_sw<NC, E>(x, v, col);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool sel = (m >> e) & 1u;
                    const bool nul = rfx_isnan_bits(x[e]);
                    const bool inf = (x[e] & 0x7FFFFFFFFFFFFFFFULL) == RFX_PINF_BITS;
                    if (sel && (inf || (nul && kind == RFX_AGG_SUM && !skip))) {
#pragma unroll
                        for (int g = 0; g < NG; g++)
                            if (key[e] == (u64)g) racc[a][g] = rfx_as_u64(rfx_as_f64(racc[a][g]) + rfx_as_f64(x[e]));
                    }
                }
            }
        }
    }
    // lanes -> workgroup (LDS) -> global tables
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (rfirst[g] == 0xffffffffu) continue; // this lane saw no row of the group: its cells are identities
        atomicMin(&lfirst[g], rfirst[g]);
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const int kind = P.aggs[a].kind;
            // (COUNT: the shared counter; an average under the masked fma: the selected rows less its column's nulls)
            few_merge_lds(&lacc[a][g], &lcnt[a][g], kind, P.aggs[a].f64, kind == RFX_AGG_COUNT ? (u64)rsel[g] : racc[a][g],
                          (FEW_FMA && kind == RFX_AGG_AVG) ? rsel[g] - rcnt[a][g] : rcnt[a][g]);
        }
    }
    __syncthreads();
    if (tid < NG) {
        const unsigned lf = lfirst[tid];
        if (lf != 0xffffffffu) {
            const u64 f = (u64)(P0.row0 + (i64)lf);
            if (f < G.first[tid]) atomicMin((unsigned long long *)&G.first[tid], (unsigned long long)f);
#pragma unroll
            for (int a = 0; a < NA; a++) {
                const bool hc = agg_has_cnt(P.aggs[a].kind, P.aggs[a].f64);
                group_merge_cell(&G.acc[a][tid], hc ? &G.cnt[a][tid] : (u64 *)0, P.aggs[a].kind, P.aggs[a].f64, lacc[a][tid], hc ? (u64)lcnt[a][tid] : 0ULL);
            }
        }
    }
}
