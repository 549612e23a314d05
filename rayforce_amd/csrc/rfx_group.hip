// rfx_group.hip -- dense group-by: K7 first-occurrence table, K8 first-occurrence ranking, K10 fused scatter-aggregate.
//
// Reference: index_group_i64_scoped (core/index.c:2002-2092) builds group ids with a SEQUENTIAL first-occurrence loop
// over all rows and AGGR_ITER/AGGR_COLLECT (core/aggr.c:73-181) scatter into per-thread arrays of `groups` entries.
// Here one fused pass reads key + predicate + value columns once and scatters into tables indexed by  key - kmin :
//   d_first[slot]  = min global row id           (atomic min)      -> group ORDER = ascending first row
//   d_acc[a][slot] = running aggregate           (atomic add/min/max on 8-byte cells)
// Tables that fit the LDS budget are privatised per workgroup in LDS (ds_* atomics run at HBM streaming rate:
// tools/probe_hw measured 400+ G rows/s against 23 G rows/s for device-scope atomics) and merged into the global
// tables once per workgroup.  Larger ranges use the radix-partitioned path (rfx_group_part.hip) or, as the
// always-correct fallback in this file, direct device-scope atomics.
// The rank step reuses `where`'s bitmap machinery: mark bit first[slot] in a 1-bit-per-row bitmap, prefix-popcount
// it, and the rank of a slot's first row IS its group id (first-occurrence order, bit-exact with the reference).
#include "rfx_group_common.hpp"

int rfx_group_part_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t); // rfx_group_part.hip


// ---- K7 + K10, one pass.  LDS = true: tables privatised in dynamic LDS, merged at the end. ----
// TINY = true (256-lane LDS form only), two things for small table sets:
//  * expression TREES are evaluated on the fly too (expr_input_deep) instead of being materialised by k_derive, so a wide plan
//    (TPC-H Q1: 7 columns, 8 outputs) streams ONCE;
//  * the tables are replicated 2^rep_shift times, replica = low bits of the lane id: with two to four groups all 64 lanes of
//    a ds_add hit the same addresses and the LDS serialises them; with lane-private replicas every lane of an instruction has
//    its own cell and bank.  Replicas are folded before the merge.
template <int NC, bool LDS, int BLOCK, bool TINY = false, int NPT = RFX_MAX_PREDS>
__device__ __forceinline__ void group_dense_body(const Plan &P, const GroupArgs &G) {
    constexpr bool DEEP = TINY;
    constexpr bool SW = TINY || BLOCK == 1024; // column operands through a wave-uniform switch instead of select chains
    constexpr int U = TINY ? (NC <= 5 ? 4 : 3) : ((NC <= 2) ? 4 : (NC <= 4 ? 2 : 1)); // rows per lane = 2 U; TINY: measured per NC (tools/q1_variants.py)
    constexpr int E = 2 * U;
    constexpr int TILE = BLOCK * E;
    constexpr int JSTRIDE = BLOCK * 2;
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int tid = threadIdx.x;
    const i64 range = G.range;
    PredSet<NPT> S; // (the TINY form is instantiated for 0 / <= 2 / <= 8 predicates: 8 descriptor sets in SGPRs cost it 10 ms per 1e9 rows)
    predset_load<NPT>(P, S);

    // LDS layout (compact: more groups fit): [acc arrays: u64 x range each] [first: u32 x range, local row, 0xffffffff = none]
    // [count arrays: u32 x range each].  Local rows and per-workgroup counts fit 32 bits because nrows < 2^32 on this path.
    int nacc = 0, ncnt = 0;
    for (int a = 0; a < G.nagg; a++) {
        nacc++;
        if (agg_has_cnt(P.aggs[a].kind, P.aggs[a].f64)) ncnt++;
    }
    const int rs = TINY ? G.rep_shift : 0;
    const i64 lrange = range << rs; // cells per LDS array
    const unsigned rl = TINY ? ((unsigned)tid & ((1u << rs) - 1u)) : 0u; // this lane's replica
    unsigned *lfirst = (unsigned *)(smem + (i64)nacc * lrange);
    unsigned *lcnt = lfirst + lrange;
    if (LDS) {
        for (i64 i = tid; i < lrange; i += BLOCK) lfirst[i] = 0xffffffffu;
        for (i64 i = tid; i < (i64)ncnt * lrange; i += BLOCK) lcnt[i] = 0;
        for (int a = 0; a < G.nagg; a++) {
            const u64 id = acc_identity(P.aggs[a].kind, P.aggs[a].f64);
            for (i64 i = tid; i < lrange; i += BLOCK) smem[(i64)a * lrange + i] = id;
        }
        __syncthreads();
    }

    const i64 nfull = P.nrows / TILE;
    const i64 ntiles = nfull + ((nfull * TILE < P.nrows) ? 1 : 0);
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 base = t * TILE + tid * 2;
        u64 v[NC][E];
        unsigned valid;
        if (t < nfull) {
            valid = (1u << E) - 1u;
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
        } else {
            valid = 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
                i64 row = base + (i64)(e >> 1) * JSTRIDE + (e & 1);
                bool in = row < P.nrows;
                valid |= (unsigned)in << e;
#pragma unroll
                for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
            }
        }
        const unsigned m = eval_preds<NC, E, NPT>(S, v, valid);
        if (m == 0) continue;
        u64 key[E]; // slot in the dense table
        if (G.nkeys <= 1) {
            if (SW) sel_col_sw<NC, E>(key, v, G.key_idx);
            else sel_col<NC, E>(key, v, G.key_idx);
#pragma unroll
            for (int e = 0; e < E; e++) key[e] -= (u64)G.kmin;
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) key[e] = 0;
            bool out[E];
#pragma unroll
            for (int e = 0; e < E; e++) out[e] = false;
            for (int i = 0; i < G.nkeys; i++) {
                u64 x[E];
                if (SW) sel_col_sw<NC, E>(x, v, G.kidx[i]);
                else sel_col<NC, E>(x, v, G.kidx[i]);
                const u64 mn = G.kmn[i], mu = G.kmul[i], rg = G.krng[i];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    out[e] |= (x[e] - mn) >= rg; // (a key outside its column's range could wrap the composite back into the table)
                    key[e] += (x[e] - mn) * mu;
                }
            }
#pragma unroll
            for (int e = 0; e < E; e++) key[e] = out[e] ? ~0ULL : key[e];
        }
        // first-occurrence table
#pragma unroll
        for (int e = 0; e < E; e++) {
            if (!((m >> e) & 1u)) continue;
            const u64 slot = key[e];
            if (slot >= (u64)range) { // outside the agreed scope: impossible when the scope came from these rows, reported when it was sampled
                if (G.oob) *(volatile unsigned *)G.oob = 1u;
                continue;
            }
            const u64 row = (u64)(P.row0 + base + (i64)(e >> 1) * JSTRIDE + (e & 1));
            if (LDS) {
                const unsigned lrow = (unsigned)(base + (i64)(e >> 1) * JSTRIDE + (e & 1));
                const u64 ls = TINY ? ((slot << rs) + rl) : slot;
                if (lrow < lfirst[ls]) atomicMin(&lfirst[ls], lrow); // (round 6 A/B: one no-return ds_min_u32 instead changes nothing -- q2 3.78 ms either way)
            } else {
                // plain pre-check: a stale (larger) value only costs a redundant atomic, never a wrong minimum
                if (row < G.first[slot]) atomicMin((unsigned long long *)&G.first[slot], (unsigned long long)row);
            }
        }
        int ci = 0; // index of this aggregate's count array in LDS
        for (int a = 0; a < G.nagg; a++) {
            const PlanAgg ag = P.aggs[a];
            const bool hc = agg_has_cnt(ag.kind, ag.f64);
            u64 x[E];
            if (ag.col >= RFX_XCOL) { // expression folded on the fly
                if (DEEP) expr_input_deep_sw<NC, E>(x, v, P.xs[ag.col - RFX_XCOL]);
                else expr_input<NC, E>(x, v, P.xs[ag.col - RFX_XCOL]);
            } else if (ag.col >= 0) {
                if (SW) sel_col_sw<NC, E>(x, v, ag.col);
                else sel_col<NC, E>(x, v, ag.col);
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (!((m >> e) & 1u)) continue;
                const u64 slot = key[e];
                if (slot >= (u64)range) continue;
                if (LDS) {
                    const u64 ls = TINY ? ((slot << rs) + rl) : slot;
                    group_apply(&smem[(i64)a * lrange + ls], &lcnt[(i64)ci * lrange + ls], ag.kind, ag.f64, x[e], ag.skipnull);
                }
                else group_apply(&G.acc[a][slot], G.cnt[a] ? &G.cnt[a][slot] : (u64 *)0, ag.kind, ag.f64, x[e], ag.skipnull);
            }
            ci += hc ? 1 : 0;
        }
    }

    if (LDS) {
        __syncthreads();
        const int nrep = 1 << rs;
        for (i64 i = tid; i < range; i += BLOCK) {
            unsigned lf = lfirst[i << rs];
            for (int r = 1; r < nrep; r++) lf = min(lf, lfirst[(i << rs) + r]);
            if (lf == 0xffffffffu) continue; // slot untouched by this workgroup
            const u64 f = (u64)(P.row0 + (i64)lf);
            if (f < G.first[i]) atomicMin((unsigned long long *)&G.first[i], (unsigned long long)f);
            int ci = 0;
            for (int a = 0; a < G.nagg; a++) {
                const PlanAgg ag = P.aggs[a];
                const bool hc = agg_has_cnt(ag.kind, ag.f64);
                u64 av = smem[(i64)a * lrange + (i << rs)];
                u64 cv = hc ? (u64)lcnt[(i64)ci * lrange + (i << rs)] : 0ULL;
                for (int r = 1; r < nrep; r++) { // fold the replicas (replica order: fixed, so a workgroup's partial is reproducible)
                    const u64 x = smem[(i64)a * lrange + (i << rs) + r];
                    if (hc) cv += (u64)lcnt[(i64)ci * lrange + (i << rs) + r];
                    switch (ag.kind) {
                        case RFX_AGG_MIN: av = ((i64)x < (i64)av) ? x : av; break;
                        case RFX_AGG_MAX: av = ((i64)x > (i64)av) ? x : av; break;
                        case RFX_AGG_COUNT: av += x; break;
                        case RFX_AGG_SUM:
                            if (!ag.f64) { av += x; break; }
                            [[fallthrough]];
                        default: av = (u64)__double_as_longlong(rfx_as_f64(av) + rfx_as_f64(x)); break; // f64 SUM, AVG
                    }
                }
                group_merge_cell(&G.acc[a][i], hc ? &G.cnt[a][i] : (u64 *)0, ag.kind, ag.f64, av, cv);
                ci += hc ? 1 : 0;
            }
        }
    }
}
template <int NC, bool LDS, int BLOCK, bool TINY = false, int NPT = RFX_MAX_PREDS>
__global__ __launch_bounds__(BLOCK) void k_group_dense(const Plan P, const GroupArgs G) {
    group_dense_body<NC, LDS, BLOCK, TINY, NPT>(P, G);
}

// ---- table init ----
__global__ __launch_bounds__(RFX_BLOCK) void k_fill_u64(u64 *p, i64 n, u64 val) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) p[i] = val;
}
int rfx_fill_u64(rfx_ctx *c, void *p, i64 n, u64 val) {
    if (n <= 0 || !p) return RFX_OK;
    i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c) * 4;
    if (blocks < grid) grid = (int)blocks;
    hipLaunchKernelGGL(k_fill_u64, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (u64 *)p, n, val);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_group_table_arrays(const rfx_agg_t *aggs, int nagg, int *n_arrays) {
    if (!n_arrays || nagg < 0 || nagg > RFX_MAX_AGGS || (nagg && !aggs)) return RFX_EINVAL;
    int n = 1;
    for (int a = 0; a < nagg; a++) n += 1 + (agg_has_cnt(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64) ? 1 : 0);
    *n_arrays = n;
    return RFX_OK;
}

static int check_tables(const rfx_agg_t *aggs, const rfx_group_tables_t *t) {
    RFX_REQUIRE(t && t->d_first, RFX_EINVAL, "tables / d_first is NULL");
    RFX_REQUIRE(t->range > 0, RFX_EINVAL, "range must be > 0");
    RFX_REQUIRE(t->nagg >= 0 && t->nagg <= RFX_MAX_AGGS, RFX_ELIMIT, "too many aggregates");
    for (int a = 0; a < t->nagg; a++) {
        RFX_REQUIRE(t->d_acc[a] != NULL, RFX_EINVAL, "d_acc[a] is NULL");
        if (agg_has_cnt(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64)) RFX_REQUIRE(t->d_cnt[a] != NULL, RFX_EINVAL, "d_cnt[a] is NULL for SUM(i64)/AVG");
    }
    return RFX_OK;
}

// small tables: every array of the set in ONE launch (a launch per array is most of a small query's cost)
struct FillSet {
    u64 *p[1 + 2 * RFX_MAX_AGGS];
    u64 v[1 + 2 * RFX_MAX_AGGS];
    int n;
    i64 cells;
};
__global__ __launch_bounds__(RFX_BLOCK) void k_fill_tables(const FillSet F) {
    const i64 total = F.cells * F.n;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < total; i += (i64)gridDim.x * RFX_BLOCK) {
        const int a = (int)(i / F.cells);
        F.p[a][i - (i64)a * F.cells] = F.v[a];
    }
}
extern "C" int rfx_hip_group_tables_init(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    int rc = check_tables(aggs, t);
    if (rc != RFX_OK) return rc;
    if (t->range <= (1 << 16)) {
        FillSet F;
        F.n = 0;
        F.cells = t->range;
        F.p[F.n] = (u64 *)t->d_first;
        F.v[F.n++] = (u64)RFX_INF_I64_D;
        for (int a = 0; a < t->nagg; a++) {
            F.p[F.n] = (u64 *)t->d_acc[a];
            F.v[F.n++] = acc_identity(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64);
            if (t->d_cnt[a]) {
                F.p[F.n] = (u64 *)t->d_cnt[a];
                F.v[F.n++] = 0;
            }
        }
        const i64 blocks = (F.cells * F.n + RFX_BLOCK - 1) / RFX_BLOCK;
        const int grid = blocks < rfx_grid((rfx_ctx *)c) * 4 ? (int)blocks : rfx_grid((rfx_ctx *)c) * 4;
        hipLaunchKernelGGL(k_fill_tables, dim3(grid), dim3(RFX_BLOCK), 0, ((rfx_ctx *)c)->stream, F);
        RFX_HIP_CHECK(hipGetLastError());
        return RFX_OK;
    }
    rc = rfx_fill_u64(c, t->d_first, t->range, (u64)RFX_INF_I64_D);
    if (rc != RFX_OK) return rc;
    for (int a = 0; a < t->nagg; a++) {
        rc = rfx_fill_u64(c, t->d_acc[a], t->range, acc_identity(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64));
        if (rc != RFX_OK) return rc;
        if (t->d_cnt[a]) {
            rc = rfx_fill_u64(c, t->d_cnt[a], t->range, 0);
            if (rc != RFX_OK) return rc;
        }
    }
    return RFX_OK;
}

#define RFX_LDS_GROUP_BYTES (64 * 1024)      /* per 256-thread workgroup: two workgroups per CU keep streaming at full rate */
#define RFX_LDS_GROUP_BIG_BYTES (160 * 1024) /* one 1024-thread workgroup per CU owning the whole LDS (mid-range key counts) */

// RFX_DUMP_PLAN=1: the descriptor fields of a launch as C++ conditions (development: what a plan-specialised kernel would fix)
static void dump_plan(const Plan &P, const GroupArgs &G, int nc, int grid, size_t lds, bool deep) {
    fprintf(stderr, "// NC %d deep %d grid %d lds %zu nrows %lld\n", nc, (int)deep, grid, lds, (long long)P.nrows);
    fprintf(stderr, "A(P.ncols == %d); A(P.npred == %d); A(P.nagg == %d); A(P.logic == %d); A(P.nx == %d);\n", P.ncols, P.npred, P.nagg, P.logic, P.nx);
    for (int i = 0; i < P.npred; i++)
        fprintf(stderr, "A(P.preds[%d].col == %d); A(P.preds[%d].rhs_col == %d); A(P.preds[%d].op == %d); A(P.preds[%d].dom_f64 == %d); A(P.preds[%d].lhs_cvt == %d); A(P.preds[%d].rhs_cvt == %d);\n",
                i, P.preds[i].col, i, P.preds[i].rhs_col, i, P.preds[i].op, i, P.preds[i].dom_f64, i, P.preds[i].lhs_cvt, i, P.preds[i].rhs_cvt);
    for (int a = 0; a < P.nagg; a++)
        fprintf(stderr, "A(P.aggs[%d].col == %d); A(P.aggs[%d].f64 == %d); A(P.aggs[%d].kind == %d); A(P.aggs[%d].skipnull == %d);\n", a, P.aggs[a].col, a, P.aggs[a].f64, a,
                P.aggs[a].kind, a, P.aggs[a].skipnull);
    for (int x = 0; x < P.nx; x++) {
        fprintf(stderr, "A(P.xs[%d].nops == %d); A(P.xs[%d].out_f64 == %d);\n", x, P.xs[x].nops, x, P.xs[x].out_f64);
        const int *w = (const int *)P.xs[x].ops;
        for (int j = 0; j < P.xs[x].nops * (int)(sizeof(PlanXNode) / 4); j++) fprintf(stderr, "A(((const int *)P.xs[%d].ops)[%d] == %d); ", x, j, w[j]);
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "A(G.key_idx == %d); A(G.nagg == %d); A(G.nkeys == %d); A(G.rep_shift == %d); A(G.range == %lld); A(G.kmin == %lld);\n", G.key_idx, G.nagg, G.nkeys, G.rep_shift,
            (long long)G.range, (long long)G.kmin);
    for (int i = 0; i < G.nkeys; i++) fprintf(stderr, "A(G.kidx[%d] == %d); A(G.kmn[%d] == %lluULL); A(G.kmul[%d] == %lluULL);\n", i, G.kidx[i], i, (unsigned long long)G.kmn[i], i, (unsigned long long)G.kmul[i]);
}
template <int NC>
static int launch_group(rfx_ctx *c, const Plan &P, const GroupArgs &G, int grid, size_t lds_bytes, bool deep) {
    if (getenv("RFX_DUMP_PLAN")) dump_plan(P, G, NC, grid, lds_bytes, deep);
    // a handful of groups: register accumulators, one kernel per plan compiled at run time (rfx_rtc.hip); RFX_ESTATE = not available
    if (deep && G.range <= RFX_FEW_MAX_GROUPS && !(c->flags & RFX_TUNE_NO_RTC)) {
        const int rrc = rfx_rtc_group_few(c, P, G, grid);
        if (rrc != RFX_ESTATE) return rrc;
    }
    if (deep) { // TINY form (group_dense_run decides): 256 lanes, table replicas within the 64 KB LDS budget
        if (P.npred == 0) hipLaunchKernelGGL((k_group_dense<NC, true, RFX_BLOCK, true, 0>), dim3(grid), dim3(RFX_BLOCK), lds_bytes, c->stream, P, G);
        else if (P.npred <= 2) hipLaunchKernelGGL((k_group_dense<NC, true, RFX_BLOCK, true, 2>), dim3(grid), dim3(RFX_BLOCK), lds_bytes, c->stream, P, G);
        else hipLaunchKernelGGL((k_group_dense<NC, true, RFX_BLOCK, true, RFX_MAX_PREDS>), dim3(grid), dim3(RFX_BLOCK), lds_bytes, c->stream, P, G);
        return RFX_OK;
    }
    if (lds_bytes > RFX_LDS_GROUP_BYTES) {
        // (dynamic LDS above 64 KB must be opted into once per template instance)
#define RFX_BIG_LAUNCH(...)                                                                                                                           \
    do {                                                                                                                                              \
        static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */                                                                                                                 \
        if (!((attr_set >> (c->device & 63)) & 1ull)) {                                                                                                                              \
            RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_group_dense<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, RFX_LDS_GROUP_BIG_BYTES)); \
            __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);                                                                                                                          \
        }                                                                                                                                             \
        hipLaunchKernelGGL((k_group_dense<__VA_ARGS__>), dim3(c->num_cus), dim3(1024), lds_bytes, c->stream, P, G);                                    \
    } while (0)
        if (P.npred == 0) RFX_BIG_LAUNCH(NC, true, 1024, false, 0); // predicate descriptor sets sized to the query, as in the TINY form
        else if (P.npred <= 2) RFX_BIG_LAUNCH(NC, true, 1024, false, 2);
        else RFX_BIG_LAUNCH(NC, true, 1024, false, RFX_MAX_PREDS);
#undef RFX_BIG_LAUNCH
    } else if (lds_bytes) hipLaunchKernelGGL((k_group_dense<NC, true, RFX_BLOCK>), dim3(grid), dim3(RFX_BLOCK), lds_bytes, c->stream, P, G);
    else hipLaunchKernelGGL((k_group_dense<NC, false, RFX_BLOCK>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, G);
    return RFX_OK;
}

// two parts of a table set: aggregates [0, h) and [h, nagg) over the same `first` array (h <= 0: halves)
// bytes of the compact LDS table set: 8 per aggregate + 4 (first) + 4 per count array, per slot
static size_t lds_table_bytes(i64 range, const rfx_agg_t *aggs, int nagg) {
    size_t per = 4;
    for (int a = 0; a < nagg; a++) per += 8 + (agg_has_cnt(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64) ? 4 : 0);
    return ((size_t)range * per + 15) & ~(size_t)15;
}

static int split_tables(const rfx_group_tables_t *t, rfx_group_tables_t *t1, rfx_group_tables_t *t2, int h = 0) {
    if (h <= 0) h = t->nagg / 2;
    *t1 = *t;
    *t2 = *t;
    t1->nagg = h;
    t2->nagg = t->nagg - h;
    for (int a = 0; a < t2->nagg; a++) {
        t2->d_acc[a] = t->d_acc[h + a];
        t2->d_cnt[a] = t->d_cnt[h + a];
    }
    return h;
}

// Mid-range key counts with several aggregates: the whole table set does not fit LDS (so the query would take the
// partitioned path: 29 ms per 1e9 rows with three value planes) but every aggregate alone does.  Then one streaming LDS
// pass per group of aggregates -- each re-reads the key column, 2.5-3.7 ms -- is several times cheaper.  Returns how many
// leading aggregates the first pass takes, 0 when no split is called for.
static int lds_pass_split(const rfx_ctx *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t) {
    if (t->nagg < 2 || (c->flags & (RFX_TUNE_NO_LDS_TABLES | RFX_TUNE_NO_LDS_SPLIT))) return 0;
    if (lds_table_bytes(t->range, aggs, t->nagg) <= ((c->flags & RFX_TUNE_NO_BIG_LDS) ? RFX_LDS_GROUP_BYTES : RFX_LDS_GROUP_BIG_BYTES)) return 0;
    const size_t cap = (c->flags & RFX_TUNE_NO_BIG_LDS) ? RFX_LDS_GROUP_BYTES : RFX_LDS_GROUP_BIG_BYTES;
    const size_t cell = (size_t)t->range * 4; // the first-row array
    size_t total = cell, run = cell;
    int h = 0;
    bool open = true;
    for (int a = 0; a < t->nagg; a++) {
        const size_t mine = (size_t)t->range * (8 + (agg_has_cnt(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64) ? 4 : 0)) + 16;
        if (cell + mine > cap) return 0; // this aggregate alone does not fit
        total += mine;
        if (open && run + mine <= cap) {
            run += mine;
            h = a + 1;
        } else open = false;
    }
    return (total > cap && h >= 1 && h < t->nagg) ? h : 0;
}

// Expression trees stay inside the scatter pass when the table set fits the 64 KB LDS form: nothing is materialised and the
// operand columns are read once.  Everywhere else (big LDS, partitioned, atomics) k_derive writes them out first.
#define RFX_LDS_TINY_BYTES ((size_t)32 << 10)
static bool group_deep_inline(const rfx_ctx *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t, i64 nrows) {
    return !(c->flags & (RFX_TUNE_NO_LDS_TABLES | RFX_TUNE_NO_DEEP_GROUP)) && nrows < (1LL << 32) && lds_table_bytes(t->range, aggs, t->nagg) <= RFX_LDS_GROUP_BYTES;
}

static int group_dense_run(rfx_ctx_t *c, Plan &P, GroupArgs &G, const rfx_agg_t *aggs, const rfx_group_tables_t *t, bool allow_part, bool *need_materialise) {
    const bool deep = rfx_plan_has_deep_expr(P) && group_deep_inline(c, aggs, t, P.nrows);
    if (rfx_plan_has_deep_expr(P) && !deep) { // expression trees: evaluated into scratch columns first (k_derive), then plain columns
        if (P.ncols + P.nx > RFX_MAX_COLS) {
            rfx_set_error("group_dense_accumulate: too many distinct columns once the expression trees are materialised");
            return RFX_ELIMIT;
        }
        const int rc0 = rfx_plan_materialise_exprs(c, &P);
        if (rc0 != RFX_OK) return rc0;
    }
    G.kmin = t->kmin;
    G.range = t->range;
    G.nagg = t->nagg;
    G.first = (u64 *)t->d_first;
    G.oob = (unsigned *)c->ext_p[1]; // NULL until rfx_hip_ctx_speculative was called once
    for (int i = 0; i < G.nkeys; i++) // column i's range from the multipliers (mult_i = product of the ranges before it)
        G.krng[i] = (i + 1 < G.nkeys) ? (G.kmul[i] ? G.kmul[i + 1] / G.kmul[i] : 0) : (G.kmul[i] ? (u64)t->range / G.kmul[i] : 0);
    int narr = 1;
    for (int a = 0; a < t->nagg; a++) {
        G.acc[a] = (u64 *)t->d_acc[a];
        G.cnt[a] = (u64 *)t->d_cnt[a];
        narr += 1 + (agg_has_cnt(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64) ? 1 : 0);
    }
    (void)narr;
    // TINY form: expression trees inline, or two to four groups under several aggregates (lane-private replicas, <= 32 KB).
    // Measured (1e9 rows, avg v1,v2,v3): 2 groups 16.7 -> 9.9 ms with replicas; from 6 groups on the same-address serialisation
    // hides behind the stream and replicas change nothing (keys 6 / 32 / 100: 9.1 vs 8.8-9.5 ms), so they stay off there.
    int rs = 0;
    if (t->range <= 4 && group_deep_inline(c, aggs, t, P.nrows) && !(c->flags & RFX_TUNE_NO_LDS_REPLICAS))
        while (rs < 6 && lds_table_bytes(t->range << (rs + 1), aggs, t->nagg) <= RFX_LDS_TINY_BYTES) rs++;
    // ... and, since the TINY form also carries predicate descriptors sized to the query and switch-selected column operands, every
    // plan whose tables fit the 64 KB form: 20-30 % on wide plans (8 plain aggregates over 7 columns 33.7 -> 28.1 ms per 1e9 rows), 8 %
    // on filtered narrow ones, 1-3 % elsewhere.  RFX_TUNE_NO_DEEP_GROUP brings the common form back (A/B, tests).
    const bool tiny = deep || group_deep_inline(c, aggs, t, P.nrows);
    G.rep_shift = tiny ? rs : 0;
    size_t lds_bytes = lds_table_bytes(tiny ? (t->range << rs) : t->range, aggs, t->nagg);
    const size_t lds_cap = (c->flags & RFX_TUNE_NO_BIG_LDS) ? RFX_LDS_GROUP_BYTES : RFX_LDS_GROUP_BIG_BYTES;
    const bool use_lds = lds_bytes <= lds_cap && !(c->flags & RFX_TUNE_NO_LDS_TABLES) && P.nrows < (1LL << 32);
    if (c->ext_i[1] && !use_lds) return RFX_ESTATE; // the scope was sampled: only the LDS-table kernels report keys outside it -- take the exact scope
    if (need_materialise) {
        // several keys: fold them on the fly only where the LDS tables make the pass stream; the big-range paths
        // (partitioned, device atomics) want the one materialised key column the reference builds too
        *need_materialise = !use_lds && !(c->flags & RFX_TUNE_FUSED_KEYS);
        if (*need_materialise) return RFX_OK;
    }
    int rc;
    if (!use_lds && allow_part && !(c->flags & RFX_TUNE_NO_PARTITION)) {
        rc = rfx_group_part_accumulate(c, P, G.key_idx, t);
        if (rc != RFX_ESTATE) return rc; // RFX_ESTATE = "partitioned path not applicable", fall through to atomics
    }
    if (!use_lds) lds_bytes = 0;
    int grid = rfx_grid(c);
    rc = RFX_OK;
    RFX_KERNEL_BEGIN(c);
    switch (P.ncols) {
        case 1: rc = launch_group<1>(c, P, G, grid, lds_bytes, tiny); break;
        case 2: rc = launch_group<2>(c, P, G, grid, lds_bytes, tiny); break;
        case 3: rc = launch_group<3>(c, P, G, grid, lds_bytes, tiny); break;
        case 4: rc = launch_group<4>(c, P, G, grid, lds_bytes, tiny); break;
        case 5: rc = launch_group<5>(c, P, G, grid, lds_bytes, tiny); break;
        case 6: rc = launch_group<6>(c, P, G, grid, lds_bytes, tiny); break;
        case 7: rc = launch_group<7>(c, P, G, grid, lds_bytes, tiny); break;
        default: rc = launch_group<8>(c, P, G, grid, lds_bytes, tiny); break;
    }
    RFX_KERNEL_END(c);
    if (rc != RFX_OK) return rc;
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_group_dense_accumulate(rfx_ctx_t *c, const int64_t *d_key, const rfx_pred_t *preds, int npred,
                                              int logic, const rfx_agg_t *aggs, int64_t nrows, int64_t row0,
                                              const rfx_group_tables_t *t) {
    RFX_REQUIRE(c && d_key, RFX_EINVAL, "NULL argument");
    int rc = check_tables(aggs, t);
    if (rc != RFX_OK) return rc;
    if (nrows == 0) return RFX_OK;
    Plan P;
    int key_idx = 0;
    const int hs = lds_pass_split(c, aggs, t);
    rc = hs ? RFX_ELIMIT : rfx_plan_build(&P, preds, npred, logic, aggs, t->nagg, d_key, &key_idx, nrows, row0);
    if (rc == RFX_OK && rfx_plan_has_deep_expr(P) && !group_deep_inline(c, aggs, t, nrows) && P.ncols + P.nx > RFX_MAX_COLS) rc = RFX_ELIMIT; // no room to materialise the trees
    if (rc == RFX_ELIMIT && t->nagg > 1) { // too many columns / expressions for one launch, or tables that fit LDS only in parts
        rfx_group_tables_t t1, t2;
        const int h = split_tables(t, &t1, &t2, hs);
        rc = rfx_hip_group_dense_accumulate(c, d_key, preds, npred, logic, aggs, nrows, row0, &t1);
        if (rc != RFX_OK) return rc;
        return rfx_hip_group_dense_accumulate(c, d_key, preds, npred, logic, aggs + h, nrows, row0, &t2);
    }
    if (rc != RFX_OK) return rc;
    GroupArgs G;
    memset(&G, 0, sizeof(G));
    G.key_idx = key_idx;
    G.nkeys = 1;
    return group_dense_run(c, P, G, aggs, t, true, NULL);
}

// Several key columns.  t->kmin must be 0 and t->range the composite range (rfx_composite_plan).
extern "C" int rfx_hip_group_dense_accumulate_keys(rfx_ctx_t *c, const void *const *d_keys, const int64_t *mins, const int64_t *mults,
                                                   int nkeys, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                                                   int64_t nrows, int64_t row0, const rfx_group_tables_t *t) {
    RFX_REQUIRE(c && d_keys && mins && mults, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nkeys >= 1 && nkeys <= RFX_MAX_KEYS, RFX_ELIMIT, "1..RFX_MAX_KEYS key columns");
    int rc = check_tables(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(t->kmin == 0, RFX_EINVAL, "composite tables start at key 0");
    if (nrows == 0) return RFX_OK;
    for (int i = 0; i < nkeys; i++) RFX_REQUIRE(d_keys[i] != NULL, RFX_EINVAL, "key column is NULL");
    Plan P;
    int k0 = 0;
    const int hs = lds_pass_split(c, aggs, t);
    rc = hs ? RFX_ELIMIT : rfx_plan_build(&P, preds, npred, logic, aggs, t->nagg, d_keys[0], &k0, nrows, row0);
    if (rc == RFX_OK && rfx_plan_has_deep_expr(P) && !group_deep_inline(c, aggs, t, nrows) && P.ncols + (nkeys - 1) + P.nx > RFX_MAX_COLS) rc = RFX_ELIMIT; // (upper bound on the key columns still to add)
    if (rc == RFX_ELIMIT && t->nagg > 1) {
        rfx_group_tables_t t1, t2;
        const int h = split_tables(t, &t1, &t2, hs);
        rc = rfx_hip_group_dense_accumulate_keys(c, d_keys, mins, mults, nkeys, preds, npred, logic, aggs, nrows, row0, &t1);
        if (rc != RFX_OK) return rc;
        return rfx_hip_group_dense_accumulate_keys(c, d_keys, mins, mults, nkeys, preds, npred, logic, aggs + h, nrows, row0, &t2);
    }
    if (rc != RFX_OK) return rc;
    GroupArgs G;
    memset(&G, 0, sizeof(G));
    G.key_idx = k0;
    G.nkeys = nkeys;
    bool fits = true;
    for (int i = 0; i < nkeys; i++) {
        G.kidx[i] = (i == 0) ? k0 : rfx_plan_add_col(&P, d_keys[i]);
        if (G.kidx[i] < 0) fits = false; // more than RFX_MAX_COLS distinct columns in one pass
        G.kmn[i] = (u64)mins[i];
        G.kmul[i] = (u64)mults[i];
    }
    bool materialise = !fits;
    if (c->ext_i[1] && !fits) return RFX_ESTATE; // sampled scope and a materialised composite key: not checked, take the exact scope
    if (fits) {
        rc = group_dense_run(c, P, G, aggs, t, false, &materialise);
        if (rc != RFX_OK || !materialise) return rc;
    }
    // one composite column (as the reference, core/index.c:2386-2418), then the single-key machinery
    rc = rfx_comp_reserve(c, (size_t)nrows * 8);
    if (rc != RFX_OK) return rc;
    rc = rfx_hip_composite_key(c, d_keys, mins, mults, nkeys, nrows, (int64_t *)c->d_comp);
    if (rc != RFX_OK) return rc;
    return rfx_hip_group_dense_accumulate(c, (const int64_t *)c->d_comp, preds, npred, logic, aggs, nrows, row0, t);
}

// ---------------- a SAMPLED scope, and the report that keeps it honest ----------------
// index_scope_i64 (core/index.c:376-435) reads the whole key column to learn [min, max] before it groups: for a key of a few
// hundred values over 1e9 rows that pass is a quarter to a third of the query.  A strided sample of 2^18 rows plus the first and
// last 2^11 almost always sees the whole range of such a key (callers only trust it for ranges of at most a tenth of the sample: a
// uniformly drawn extreme value is then missed with probability e^-10); "almost" is made exact by the kernels themselves: a selected row
// whose key lies outside the agreed scope is REPORTED (GroupArgs::oob) instead of dropped, the host asks rfx_hip_group_out_of_scope
// after the pass and, if anything was reported (an outlier, a null key, a range the sample missed), runs the exact scope and the
// pass again.  A sampled range can only be too SMALL (the sample is a subset), so an unreported pass is the exact pass.
__global__ __launch_bounds__(RFX_BLOCK) void k_scope_sample_i64(const i64 *__restrict__ key, i64 n, i64 stride, i64 nsamp, i64 edge, i64 *__restrict__ out) {
    __shared__ i64 red[2][RFX_BLOCK / RFX_WAVE];
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D;
    const i64 total = nsamp + 2 * edge;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < total; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 r = i < nsamp ? i * stride : (i < nsamp + edge ? i - nsamp : n - 1 - (i - nsamp - edge));
        if (r < 0 || r >= n) continue;
        const i64 k = key[r];
        mn = k < mn ? k : mn; // a null key is INT64_MIN: it shows as the minimum
        mx = k > mx ? k : mx;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s), omx = (i64)rfx_shfl_xor_u64((u64)mx, s);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = mn;
        red[1][threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) {
            mn = red[0][w] < mn ? red[0][w] : mn;
            mx = red[1][w] > mx ? red[1][w] : mx;
        }
        out[2 * blockIdx.x] = mn;
        out[2 * blockIdx.x + 1] = mx;
    }
}
extern "C" int rfx_hip_scope_sample_i64(rfx_ctx_t *c, const int64_t *d_key, int64_t nrows, int64_t *min, int64_t *max) {
    RFX_REQUIRE(c && d_key && min && max && nrows > 0, RFX_EINVAL, "bad argument");
    const int grid = 64;
    const i64 nsamp = nrows < RFX_SCOPE_SAMPLE_ROWS ? nrows : RFX_SCOPE_SAMPLE_ROWS, edge = nrows < (1 << 11) ? 0 : (1 << 11);
    int rc = rfx_ws_reserve(c, (size_t)grid * 16);
    if (rc != RFX_OK) return rc;
    hipLaunchKernelGGL(k_scope_sample_i64, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_key, (i64)nrows, (i64)(nrows / nsamp), nsamp, edge, (i64 *)c->d_ws);
    RFX_HIP_CHECK(hipGetLastError());
    i64 *h = (i64 *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, c->d_ws, (size_t)grid * 16, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D;
    for (int i = 0; i < grid; i++) {
        mn = h[2 * i] < mn ? h[2 * i] : mn;
        mx = h[2 * i + 1] > mx ? h[2 * i + 1] : mx;
    }
    *min = mn;
    *max = mx;
    return RFX_OK;
}
// on = 1: clears the report; the next accumulate calls run under a sampled scope (paths that cannot report out-of-scope keys answer
// RFX_ESTATE instead of running).  on = 0: back to normal, the report stays readable.
extern "C" int rfx_hip_ctx_speculative(rfx_ctx_t *c, int on) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!c->ext_p[1]) {
        void *p = NULL;
        RFX_HIP_CHECK(hipMalloc(&p, 256));
        c->ext_p[1] = p;
    }
    if (on) RFX_HIP_CHECK(hipMemsetAsync(c->ext_p[1], 0, 4, c->stream));
    c->ext_i[1] = on ? 1 : 0;
    return RFX_OK;
}
// did a pass since rfx_hip_ctx_speculative(ctx, 1) meet a selected row with a key outside the agreed scope?  (syncs)
extern "C" int rfx_hip_group_out_of_scope(rfx_ctx_t *c, int *violated) {
    RFX_REQUIRE(c && violated, RFX_EINVAL, "NULL argument");
    *violated = 0;
    if (!c->ext_p[1]) return RFX_OK;
    unsigned *h = (unsigned *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, c->ext_p[1], 4, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    *violated = h[0] != 0;
    return RFX_OK;
}

// ---------------- K8: rank occupied slots by first row ----------------
__device__ __forceinline__ void bitpos(u64 row, u64 *word, unsigned *bit) {
    const u64 g = row >> 7;
    const unsigned r = (unsigned)(row & 127);
    *word = g * 2 + (r & 1);
    *bit = r >> 1;
}

__global__ __launch_bounds__(RFX_BLOCK) void k_mark_first(const u64 *__restrict__ first, i64 slots, i64 row_base, u64 *__restrict__ bitmap) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < slots; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 f = first[i];
        if (f == (u64)RFX_INF_I64_D) continue;
        u64 w;
        unsigned b;
        bitpos(f - (u64)row_base, &w, &b);
        atomicOr((unsigned long long *)&bitmap[w], 1ULL << b);
    }
}

// The first rows of a table's groups lie where keys are NEW: with 1e6 uniformly random keys over 1e9 rows every key has shown up within
// the first ~1.4e7 rows (1e6 ln 1e6), so the ranking's bitmap, chunk counts and scan only need to reach the LAST first row -- found on the
// device (k_first_bound), handed to the kernels below as a device-side bound: nothing comes back to the host for it, the launches keep
// their worst-case grids and their surplus workgroups leave at once.  1e9 rows, 1e6 groups: 125 MB cleared + read and 1.95 M counts
// scanned before (135 us of kernels per query), 1.8 MB / 27 K after.
__global__ __launch_bounds__(RFX_BLOCK) void k_first_bound(const u64 *__restrict__ first, i64 slots, i64 row_base, i64 *__restrict__ nchunks_eff) {
    i64 mx = 0;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < slots; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 f = first[i];
        if (f == (u64)RFX_INF_I64_D) continue;
        const i64 q = (i64)((f - (u64)row_base) >> 9) + 1;
        mx = q > mx ? q : mx;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const i64 o = (i64)rfx_shfl_xor_u64((u64)mx, m);
        mx = o > mx ? o : mx;
    }
    __shared__ i64 red[RFX_BLOCK / RFX_WAVE];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { // ONE atomic per workgroup (a wave each: 16 K atomics on one address, 95 us)
        for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) mx = red[w] > mx ? red[w] : mx;
        if (mx > 0) atomicMax((long long *)nchunks_eff, (long long)mx);
    }
}
__global__ __launch_bounds__(RFX_BLOCK) void k_bitmap_clear(u64 *__restrict__ bitmap, const i64 *__restrict__ nchunks_eff) {
    const i64 nw = *nchunks_eff * 8; // 64-bit words
    for (i64 i = (blockIdx.x * (i64)RFX_BLOCK + threadIdx.x) * 2; i < nw; i += (i64)gridDim.x * RFX_BLOCK * 2) {
        bitmap[i] = 0;
        bitmap[i + 1] = 0;
    }
}

__global__ __launch_bounds__(RFX_BLOCK) void k_bitmap_counts(const u64 *__restrict__ bitmap, i64 nchunks, i64 *__restrict__ cnt, const i64 *__restrict__ nchunks_eff) {
    if (nchunks_eff && *nchunks_eff < nchunks) nchunks = *nchunks_eff;
    for (i64 q = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; q < nchunks; q += (i64)gridDim.x * RFX_BLOCK) {
        const u64 *w = bitmap + q * 8;
        int s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += __popcll(w[i]);
        cnt[q] = s;
    }
}

__global__ __launch_bounds__(RFX_BLOCK) void k_slot_gid(const u64 *__restrict__ first, i64 slots, i64 row_base, const u64 *__restrict__ bitmap,
                                                      const i64 *__restrict__ chunk_off, i64 *__restrict__ gid) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < slots; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 f = first[i];
        if (f == (u64)RFX_INF_I64_D) {
            gid[i] = -1;
            continue;
        }
        const u64 row = f - (u64)row_base;
        const u64 q = row >> 9;
        const unsigned within = (unsigned)(row & 511);
        const unsigned g = within >> 7, r = within & 127, lane = r >> 1, par = r & 1;
        const u64 *w = bitmap + q * 8;
        i64 rank = chunk_off[q];
        for (unsigned gg = 0; gg < g; gg++) rank += __popcll(w[2 * gg]) + __popcll(w[2 * gg + 1]);
        const u64 below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
        rank += __popcll(w[2 * g] & below) + __popcll(w[2 * g + 1] & below);
        if (par) rank += (i64)((w[2 * g] >> lane) & 1ULL);
        gid[i] = rank;
    }
}

// k_slot_gid + k_group_emit in one step: a slot's group id, and at once its result cells where the group falls into the window
// of slice si of nsl (RFX_SLICE_G0 / _GN) -- g read from the scan's total on the device.  *overflow: the window holds more groups than out_cap cells
__global__ __launch_bounds__(RFX_BLOCK) void k_slot_gid_emit(const EmitArgs A0, i64 row_base, const u64 *__restrict__ bitmap, const i64 *__restrict__ chunk_off,
                                                            i64 *__restrict__ gid, const i64 *__restrict__ d_total, int nsl, int si, i64 out_cap, int *__restrict__ overflow) {
    const i64 groups = *d_total;
    EmitArgs A = A0;
    A.g0 = 0;
    A.gn = groups;
    if (nsl > 1) {
        A.g0 = RFX_SLICE_G0(groups, si, nsl);
        A.gn = RFX_SLICE_GN(groups, si, nsl);
    }
    if (A.gn > out_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *overflow = 1;
        A.gn = 0;
    }
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < A.slots; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 f = A.first[i];
        if (f == (u64)RFX_INF_I64_D) {
            gid[i] = -1;
            continue;
        }
        const u64 row = f - (u64)row_base;
        const u64 q = row >> 9;
        const unsigned within = (unsigned)(row & 511);
        const unsigned gq = within >> 7, r = within & 127, lane = r >> 1, par = r & 1;
        const u64 *w = bitmap + q * 8;
        i64 rank = chunk_off[q];
        for (unsigned gg = 0; gg < gq; gg++) rank += __popcll(w[2 * gg]) + __popcll(w[2 * gg + 1]);
        const u64 below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
        rank += __popcll(w[2 * gq] & below) + __popcll(w[2 * gq + 1] & below);
        if (par) rank += (i64)((w[2 * gq] >> lane) & 1ULL);
        gid[i] = rank;
        const i64 g = rank - A.g0;
        if ((u64)g >= (u64)A.gn) continue;
        if (A.out_keys) A.out_keys[g] = A.keys ? (i64)A.keys[i] : A.kmin + i;
        if (A.out_first) A.out_first[g] = (i64)f;
        for (int a = 0; a < A.nagg; a++) {
            if (!A.out[a]) continue;
            if (A.kinds[a] == RFX_AGG_FIRST) {
                const i64 lr = (i64)f - A.row0;
                A.out[a][g] = (A.col[a] && lr >= 0 && (A.nloc == 0 || lr < A.nloc)) ? A.col[a][lr] : 0ULL;
            } else A.out[a][g] = group_final(A.kinds[a], A.f64s[a], A.acc[a][i], A.cnt[a] ? A.cnt[a][i] : 0ULL, A.skips[a]);
        }
    }
}

// shared by the dense and the hashed tables
static int rfx_rank_slots_emit(rfx_ctx *c, const u64 *d_first, i64 slots, i64 row_base, i64 total_rows, i64 *ngroups, const EmitArgs *emit, int nsl, int si, i64 out_cap) {
    c->pc_bitmap = 0; // ranking marks first rows in the context's bitmap
    c->where_n = -1;
    RFX_REQUIRE(total_rows >= 0, RFX_EINVAL, "total_rows < 0");
    const i64 nchunks = (total_rows + RFX_CHUNK - 1) / RFX_CHUNK;
    int rc = rfx_bitmap_reserve(c, nchunks * RFX_CHUNK);
    if (rc != RFX_OK) return rc;
    if (c->blksum_cap < (size_t)nchunks + 3) {
        RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_blksum) RFX_HIP_CHECK(hipFree(c->d_blksum));
        c->d_blksum = NULL;
        c->blksum_cap = 0;
        RFX_HIP_CHECK(hipMalloc((void **)&c->d_blksum, ((size_t)nchunks + 3) * 8));
        c->blksum_cap = (size_t)nchunks + 3;
    }
    rc = rfx_gid_reserve(c, slots);
    if (rc != RFX_OK) return rc;
    c->where_n = -1; // the bitmap no longer describes a `where`
    *ngroups = 0;
    if (nchunks == 0 || slots == 0) {
        c->rank_groups = 0;
        return RFX_OK;
    }
    i64 *d_total = c->d_blksum + nchunks, *d_bound = c->d_blksum + nchunks + 1; // chunks up to the last first row (device-side)
    RFX_HIP_CHECK(hipMemsetAsync(d_bound, 0, 16, c->stream)); // (the bound and, behind it, the emit's overflow flag)
    i64 sb = (slots + RFX_BLOCK - 1) / RFX_BLOCK;
    int sgrid = rfx_grid(c) * 4;
    if (sb < sgrid) sgrid = (int)sb;
    hipLaunchKernelGGL(k_first_bound, dim3(sgrid < 256 ? sgrid : 256), dim3(RFX_BLOCK), 0, c->stream, d_first, slots, row_base, d_bound);
    i64 cb = (nchunks + RFX_BLOCK - 1) / RFX_BLOCK;
    int cgrid = rfx_grid(c) * 4;
    if (cb < cgrid) cgrid = (int)cb;
    hipLaunchKernelGGL(k_bitmap_clear, dim3(cgrid), dim3(RFX_BLOCK), 0, c->stream, c->d_bitmap, (const i64 *)d_bound);
    hipLaunchKernelGGL(k_mark_first, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, d_first, slots, row_base, c->d_bitmap);
    hipLaunchKernelGGL(k_bitmap_counts, dim3(cgrid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)c->d_bitmap, nchunks, c->d_blksum, (const i64 *)d_bound);
    RFX_HIP_CHECK(hipGetLastError());
    rc = rfx_scan_counts(c, c->d_blksum, nchunks, d_total, d_bound);
    if (rc != RFX_OK) return rc;
    if (emit) { // rank -> emit WITHOUT the host in between: the last step also writes the groups' result cells (the window from the count on the device)
        EmitArgs A = *emit;
        hipLaunchKernelGGL(k_slot_gid_emit, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, A, row_base, (const u64 *)c->d_bitmap, (const i64 *)c->d_blksum, c->d_gid,
                           (const i64 *)d_total, nsl, si, out_cap, (int *)(d_bound + 1));
    } else
        hipLaunchKernelGGL(k_slot_gid, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, d_first, slots, row_base, (const u64 *)c->d_bitmap,
                           (const i64 *)c->d_blksum, c->d_gid);
    RFX_HIP_CHECK(hipGetLastError());
    i64 *h = (i64 *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, d_total, emit ? 24 : 8, hipMemcpyDeviceToHost, c->stream)); // (total, bound, overflow flag: neighbours)
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->rank_groups = h[0];
    *ngroups = h[0];
    RFX_REQUIRE(!emit || (int)h[2] == 0, RFX_ELIMIT, "rank_emit: more groups in the window than the output columns hold");
    return RFX_OK;
}
int rfx_rank_slots(rfx_ctx *c, const u64 *d_first, i64 slots, i64 row_base, i64 total_rows, i64 *ngroups) {
    return rfx_rank_slots_emit(c, d_first, slots, row_base, total_rows, ngroups, nullptr, 1, 0, 0);
}
// rank + emit, launch after launch with no host round trip between them (round 5): the host learns the group count when everything is enqueued,
// so the outputs are sized by an upper bound (`out_cap` cells a column); tables beyond 2^22 slots keep the two-step form (their emit walks the groups)
int rfx_rank_emit(rfx_ctx *c, const EmitArgs &A, i64 total_rows, int nsl, int si, i64 out_cap, i64 *ngroups) {
    RFX_REQUIRE(nsl >= 1 && si >= 0 && si < nsl && out_cap >= 0, RFX_EINVAL, "rank_emit: bad window");
    return rfx_rank_slots_emit(c, A.first, A.slots, 0, total_rows, ngroups, &A, nsl, si, out_cap);
}

extern "C" int rfx_hip_group_rank(rfx_ctx_t *c, const rfx_group_tables_t *t, int64_t total_rows, int64_t *ngroups) {
    RFX_REQUIRE(c && t && ngroups, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(t->d_first && t->range > 0, RFX_EINVAL, "bad tables");
    return rfx_rank_slots(c, (const u64 *)t->d_first, t->range, 0, total_rows, (i64 *)ngroups);
}

// ---------------- emit in group order ----------------
__global__ __launch_bounds__(RFX_BLOCK) void k_group_emit(const EmitArgs A, const i64 *__restrict__ gid) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < A.slots; i += (i64)gridDim.x * RFX_BLOCK) {
        i64 g = gid[i];
        if (g < 0) continue;
        g -= A.g0;
        if ((u64)g >= (u64)A.gn) continue; // outside this device's slice of the groups
        if (A.out_keys) A.out_keys[g] = A.keys ? (i64)A.keys[i] : A.kmin + i;
        const u64 f = A.first[i];
        if (A.out_first) A.out_first[g] = (i64)f;
        for (int a = 0; a < A.nagg; a++) {
            if (!A.out[a]) continue;
            if (A.kinds[a] == RFX_AGG_FIRST) {
                const i64 lr = (i64)f - A.row0; // row-range sharding: only the GPU that owns the group's first row has its value
                A.out[a][g] = (A.col[a] && lr >= 0 && (A.nloc == 0 || lr < A.nloc)) ? A.col[a][lr] : 0ULL;
            }
            else A.out[a][g] = group_final(A.kinds[a], A.f64s[a], A.acc[a][i], A.cnt[a] ? A.cnt[a][i] : 0ULL, A.skips[a]);
        }
    }
}

// Many groups (the row-hash path's 1e8 of 2.7e8 slots): walking the SLOTS scatters every output cell (ten arrays x 1e8 lone 8-byte
// stores: 37 ms); walking the GROUPS through the inverse permutation gathers instead (random 8-byte reads are twice as fast as
// random 8-byte writes here) and writes every output array in order.
__global__ __launch_bounds__(RFX_BLOCK) void k_emit_perm(const i64 *__restrict__ gid, i64 slots, i64 *__restrict__ perm) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < slots; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 g = gid[i];
        if (g >= 0) perm[g] = i;
    }
}
__global__ __launch_bounds__(RFX_BLOCK) void k_group_emit_by_group(const EmitArgs A, const i64 *__restrict__ perm, i64 groups) {
    const i64 gend = (A.gn < groups - A.g0 ? A.gn : groups - A.g0); // the window's groups are written from cell 0 on
    for (i64 g = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; g < gend; g += (i64)gridDim.x * RFX_BLOCK) {
        const i64 i = perm[g + A.g0];
        if (A.out_keys) A.out_keys[g] = A.keys ? (i64)A.keys[i] : A.kmin + i;
        const u64 f = A.first[i];
        if (A.out_first) A.out_first[g] = (i64)f;
        for (int a = 0; a < A.nagg; a++) {
            if (!A.out[a]) continue;
            if (A.kinds[a] == RFX_AGG_FIRST) {
                const i64 lr = (i64)f - A.row0;
                A.out[a][g] = (A.col[a] && lr >= 0 && (A.nloc == 0 || lr < A.nloc)) ? A.col[a][lr] : 0ULL;
            }
            else A.out[a][g] = group_final(A.kinds[a], A.f64s[a], A.acc[a][i], A.cnt[a] ? A.cnt[a][i] : 0ULL, A.skips[a]);
        }
    }
}

extern "C" int rfx_hip_ctx_emit_window(rfx_ctx_t *c, int64_t g0, int64_t n) {
    RFX_REQUIRE(c && g0 >= 0 && n >= 0, RFX_EINVAL, "bad emit window");
    rfx_ext(c)->emit_g0 = g0;
    rfx_ext(c)->emit_gn = n; // 0: no window
    return RFX_OK;
}
int rfx_emit_slots(rfx_ctx *c, const EmitArgs &A0) {
    if (A0.slots <= 0 || c->rank_groups == 0) return RFX_OK;
    EmitArgs A = A0;
    A.g0 = rfx_ext(c)->emit_gn > 0 ? rfx_ext(c)->emit_g0 : 0;
    A.gn = rfx_ext(c)->emit_gn > 0 ? rfx_ext(c)->emit_gn : (i64)0x7FFFFFFFFFFFFFFFLL;
    if (c->rank_groups >= (1LL << 22) && rfx_ws_reserve(c, (size_t)c->rank_groups * 8) == RFX_OK) {
        i64 *perm = (i64 *)c->d_ws;
        const int grid = rfx_grid(c) * 4;
        hipLaunchKernelGGL(k_emit_perm, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)c->d_gid, A.slots, perm);
        hipLaunchKernelGGL(k_group_emit_by_group, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, A, (const i64 *)perm, (i64)c->rank_groups);
        RFX_HIP_CHECK(hipGetLastError());
        return RFX_OK;
    }
    i64 sb = (A.slots + RFX_BLOCK - 1) / RFX_BLOCK;
    int sgrid = rfx_grid(c) * 4;
    if (sb < sgrid) sgrid = (int)sb;
    hipLaunchKernelGGL(k_group_emit, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, A, (const i64 *)c->d_gid);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// (Round 5 built rank + emit as ONE persistent launch -- bound, clear, mark, counts, scan, slot ids and emit as phases behind a grid barrier --
// and withdrew it: what one workgroup hands another between two phases must bypass the XCDs' non-coherent L2s (agent-scope atomic loads /
// stores) or pay a cache-wide write-back + invalidate per barrier; either way the launch took 0.22-0.28 ms for 1e6 slots where the ten
// launches above take 0.15 with their round trip.  profiles/r05_rank_ab.txt; git c54b80e has the kernel.)

// ---------------- few slots: rank and emit in ONE launch, no host round trip in between ----------------
// A dense table of <= RFX_RANK_SMALL slots: one 1024-lane workgroup holds every slot's first row in LDS, a group's id is the
// number of occupied slots whose first row is smaller (first rows are distinct: a row belongs to one group), and every lane
// writes its slots' results at that place.  d_block = [group count][keys: slots cells][first rows: slots][one array of `slots`
// cells per aggregate]: the caller copies it to the host in one piece and learns the group count from its first cell.
__global__ __launch_bounds__(1024) void k_rank_emit_small(const EmitArgs A, i64 *__restrict__ block) {
    __shared__ u64 f[RFX_RANK_SMALL];
    __shared__ unsigned n_occ;
    const int tid = threadIdx.x;
    const i64 S = A.slots;
    for (int i = tid; i < RFX_RANK_SMALL; i += 1024) f[i] = i < S ? A.first[i] : (u64)RFX_INF_I64_D;
    if (tid == 0) n_occ = 0;
    __syncthreads();
    for (int i = tid; i < (int)S; i += 1024) {
        const u64 mine = f[i];
        if (mine == (u64)RFX_INF_I64_D) continue;
        unsigned g = 0;
        for (int j = 0; j < (int)S; j++) g += (f[j] < mine) ? 1u : 0u;
        atomicAdd(&n_occ, 1u);
        A.out_keys[g] = A.kmin + i;
        A.out_first[g] = (i64)mine;
        for (int a = 0; a < A.nagg; a++) {
            if (A.kinds[a] == RFX_AGG_FIRST) {
                const i64 lr = (i64)mine - A.row0;
                A.out[a][g] = (A.col[a] && lr >= 0 && (A.nloc == 0 || lr < A.nloc)) ? A.col[a][lr] : 0ULL;
            } else A.out[a][g] = group_final(A.kinds[a], A.f64s[a], A.acc[a][i], A.cnt[a] ? A.cnt[a][i] : 0ULL, A.skips[a]);
        }
    }
    __syncthreads();
    if (tid == 0) block[0] = (i64)n_occ;
}
extern "C" int rfx_hip_group_rank_emit_small(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t row0, int64_t local_rows,
                                             int64_t *d_block) {
    RFX_REQUIRE(c && d_block, RFX_EINVAL, "NULL argument");
    int rc = check_tables(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(t->range >= 1 && t->range <= RFX_RANK_SMALL, RFX_EINVAL, "rank_emit_small: 1 .. RFX_RANK_SMALL slots");
    EmitArgs A;
    memset(&A, 0, sizeof(A));
    A.kmin = t->kmin;
    A.slots = t->range;
    A.nagg = t->nagg;
    A.first = (const u64 *)t->d_first;
    A.out_keys = (i64 *)d_block + 1;
    A.out_first = (i64 *)d_block + 1 + t->range;
    A.row0 = row0;
    A.nloc = local_rows;
    for (int a = 0; a < t->nagg; a++) {
        A.kinds[a] = aggs[a].kind;
        A.f64s[a] = rfx_agg_input_type(&aggs[a]) == RFX_F64;
        A.skips[a] = aggs[a].xop != RFX_X_NONE || aggs[a].nxnodes > 0;
        A.acc[a] = (const u64 *)t->d_acc[a];
        A.cnt[a] = (const u64 *)t->d_cnt[a];
        A.col[a] = (const u64 *)aggs[a].d_col;
        A.out[a] = (u64 *)d_block + 1 + (size_t)(2 + a) * t->range;
    }
    hipLaunchKernelGGL(k_rank_emit_small, dim3(1), dim3(1024), 0, ((rfx_ctx *)c)->stream, A, (i64 *)d_block);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_group_rank_emit(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t total_rows, int64_t row0, int64_t local_rows,
                                       int nsl, int si, int64_t out_cap, int64_t *d_keys, int64_t *d_first_ids, void *const *d_results, int64_t *ngroups) {
    RFX_REQUIRE(c && t && ngroups, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(local_rows >= 0, RFX_EINVAL, "local_rows < 0");
    int rc = check_tables(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(t->range >= 1 && t->range <= RFX_RANK_EMIT_MAX, RFX_EINVAL, "rank_emit: 1 .. RFX_RANK_EMIT_MAX slots");
    EmitArgs A;
    memset(&A, 0, sizeof(A));
    A.kmin = t->kmin;
    A.slots = t->range;
    A.nagg = t->nagg;
    A.first = (const u64 *)t->d_first;
    A.out_keys = (i64 *)d_keys;
    A.out_first = (i64 *)d_first_ids;
    A.row0 = row0;
    A.nloc = local_rows;
    for (int a = 0; a < t->nagg; a++) {
        A.kinds[a] = aggs[a].kind;
        A.f64s[a] = rfx_agg_input_type(&aggs[a]) == RFX_F64;
        A.skips[a] = aggs[a].xop != RFX_X_NONE || aggs[a].nxnodes > 0;
        A.acc[a] = (const u64 *)t->d_acc[a];
        A.cnt[a] = (const u64 *)t->d_cnt[a];
        A.col[a] = (const u64 *)aggs[a].d_col;
        A.out[a] = d_results ? (u64 *)d_results[a] : NULL;
    }
    return rfx_rank_emit(c, A, total_rows, nsl, si, out_cap, (i64 *)ngroups);
}
extern "C" int rfx_hip_group_emit(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t *d_keys,
                                  int64_t *d_first_ids, void *const *d_results) {
    return rfx_hip_group_emit_sharded(c, aggs, t, 0, 0, d_keys, d_first_ids, d_results);
}
extern "C" int rfx_hip_group_emit_sharded(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t, int64_t row0, int64_t local_rows,
                                          int64_t *d_keys, int64_t *d_first_ids, void *const *d_results) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(local_rows >= 0, RFX_EINVAL, "local_rows < 0");
    int rc = check_tables(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(c->gid_cap >= (size_t)t->range, RFX_ESTATE, "group_emit without group_rank");
    EmitArgs A;
    memset(&A, 0, sizeof(A));
    A.kmin = t->kmin;
    A.slots = t->range;
    A.nagg = t->nagg;
    A.first = (const u64 *)t->d_first;
    A.out_keys = (i64 *)d_keys;
    A.out_first = (i64 *)d_first_ids;
    A.row0 = row0;
    A.nloc = local_rows;
    for (int a = 0; a < t->nagg; a++) {
        A.kinds[a] = aggs[a].kind;
        A.f64s[a] = rfx_agg_input_type(&aggs[a]) == RFX_F64;
        A.skips[a] = aggs[a].xop != RFX_X_NONE || aggs[a].nxnodes > 0;
        A.acc[a] = (const u64 *)t->d_acc[a];
        A.cnt[a] = (const u64 *)t->d_cnt[a];
        A.col[a] = (const u64 *)aggs[a].d_col;
        A.out[a] = d_results ? (u64 *)d_results[a] : NULL;
    }
    return rfx_emit_slots(c, A);
}

// ---------------- per-row group ids (INDEX_TYPE_IDS payload) ----------------
__global__ __launch_bounds__(RFX_BLOCK) void k_group_ids(const i64 *__restrict__ key, i64 n, i64 kmin, i64 range, const i64 *__restrict__ gid,
                                                       i64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        u64 s = (u64)key[i] - (u64)kmin;
        out[i] = (s < (u64)range) ? gid[s] : -1;
    }
}

// ... the same through a caller's slot -> group id table (a shard that did not rank the merged tables itself)
extern "C" int rfx_hip_group_ids_table(rfx_ctx_t *c, const int64_t *d_key, int64_t nrows, int64_t kmin, int64_t range, const int64_t *d_table, int64_t *d_gids) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_key && d_gids && d_table && range > 0, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_group_ids, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_key, (i64)nrows, (i64)kmin, (i64)range, (const i64 *)d_table, (i64 *)d_gids);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
// ... and for SPARSE keys (the hashed tables: no slot -> id table over the key range): from every row's group-first row (the join probe against the
// group-by's own table, RFX_Q_PROBE_FIRST) and the groups' first rows in first-occurrence order -- strictly ascending, so group g's first row names g:
// gids[first[g]] = g, then every other row copies its first row's id (first rows are never rewritten: no cell is read while it changes).
__global__ __launch_bounds__(RFX_BLOCK) void k_gid_seed(const i64 *__restrict__ first, i64 groups, i64 *__restrict__ gids) {
    for (i64 g = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; g < groups; g += (i64)gridDim.x * RFX_BLOCK) gids[first[g]] = g;
}
__global__ __launch_bounds__(RFX_BLOCK) void k_gid_spread(const i64 *__restrict__ probe, i64 n, i64 *gids) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 f = probe[i];
        if (f != i) gids[i] = ((u64)f < (u64)n) ? gids[f] : -1;
    }
}
extern "C" int rfx_hip_group_ids_first(rfx_ctx_t *c, const int64_t *d_probe_first, int64_t nrows, const int64_t *d_first, int64_t groups, int64_t *d_gids) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_probe_first && d_first && d_gids && groups > 0, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_gid_seed, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_first, (i64)groups, (i64 *)d_gids);
    hipLaunchKernelGGL(k_gid_spread, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_probe_first, (i64)nrows, (i64 *)d_gids);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_group_ids_dense(rfx_ctx_t *c, const int64_t *d_key, int64_t nrows, const rfx_group_tables_t *t,
                                       int64_t *d_gids) {
    RFX_REQUIRE(c && t, RFX_EINVAL, "NULL argument");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_key && d_gids, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(c->gid_cap >= (size_t)t->range, RFX_ESTATE, "group_ids without group_rank");
    hipLaunchKernelGGL(k_group_ids, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_key, (i64)nrows, t->kmin, t->range,
                       (const i64 *)c->d_gid, (i64 *)d_gids);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
