// rfx_scalar.hip -- K1/K5 fused predicate -> scalar aggregates, K2 byte masks, K6 key scope.
//
// One pass over every DISTINCT column the query touches: 16-byte non-temporal loads (two 8-byte rows per lane,
// U of them in flight per lane per column), predicates evaluated in registers into a per-lane bit set, aggregates
// folded into per-lane accumulators, wave reduction by cross-lane shuffles, one partial per workgroup, and a
// single-workgroup second kernel that folds the partials in a fixed order (run-to-run deterministic f64 sums).
// Replaces, for i64/f64 columns, the reference's mask -> and -> where -> gather -> fold chain
// (core/cmp.c:35-68, core/logic.c:34-86, core/ops.c:254-273, core/rayforce.c:1036-1158, core/math.c:37-44).
#include "rfx_scalar_kernel.hpp"

// Fold nblocks workgroup partials per aggregate into rfx_partial_t (fixed order => deterministic).
__global__ __launch_bounds__(RFX_BLOCK) void k_filter_aggr_final(const Plan P, const Acc *__restrict__ ws, int nblocks, int na_stride,
                                                               rfx_partial_t *__restrict__ out) {
    __shared__ Acc lds[RFX_BLOCK / RFX_WAVE];
    const int tid = threadIdx.x;
    for (int a = 0; a <= P.nagg; a++) {
        const bool is_sel = (a == P.nagg);
        const int kind = is_sel ? RFX_AGG_COUNT : P.aggs[a].kind;
        const int f64 = is_sel ? 0 : P.aggs[a].f64;
        const int slot = is_sel ? (na_stride - 1) : a;
        Acc r;
        acc_init(r, kind);
        for (int b = tid; b < nblocks; b += RFX_BLOCK) acc_combine(r, ws[(size_t)b * na_stride + slot], kind, f64);
        for (int s = 32; s >= 1; s >>= 1) {
            Acc o = acc_shfl_xor(r, s);
            acc_combine(r, o, kind, f64);
        }
        if (tid % RFX_WAVE == 0) lds[tid / RFX_WAVE] = r;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) acc_combine(r, lds[w], kind, f64);
            rfx_partial_t o;
            o.isum = 0;
            o.fsum = 0.0;
            o.cnt = 0;
            o.ext = 0;
            o.pos = RFX_INF_I64_D;
            o._rsv[0] = o._rsv[1] = o._rsv[2] = 0;
            switch (kind) {
                case RFX_AGG_SUM:
                case RFX_AGG_AVG:
                    if (f64) o.fsum = rfx_as_f64(r.v);
                    else o.isum = (i64)r.v;
                    o.cnt = r.c;
                    break;
                case RFX_AGG_MIN:
                case RFX_AGG_MAX:
                    o.ext = f64 ? (i64)rfx_ord_to_f64((i64)r.v) : (i64)r.v;
                    o.cnt = r.c;
                    break;
                case RFX_AGG_COUNT:
                    o.cnt = r.c;
                    break;
                case RFX_AGG_FIRST:
                    o.ext = (i64)r.v;
                    o.pos = r.c;
                    o.cnt = (r.c != RFX_INF_I64_D) ? 1 : 0;
                    break;
                default:
                    break;
            }
            out[a] = o;
        }
        __syncthreads();
    }
}

// Workgroups per CU of the fused reduction, by distinct-column count (in-process sweep, bench.py --ab): the lighter the
// ALU work per byte, the fewer waves it takes to keep HBM busy -- 1 column: 4, 2 columns: 2, 3+ columns (more predicates /
// aggregates per row): 6.  blocks_per_cu (rfx_hip_ctx_tune) scales it: 2 = as measured.
// Expression aggregates add ALU work per row: 16 (x6, 3 columns: 4.91 ms at 6, 4.34 at 12, 4.17 at 24).
static inline int rfx_scalar_wg_per_cu(int ncols, int nx) { return nx > 0 ? 16 : (ncols <= 1 ? 4 : (ncols == 2 ? 2 : 6)); }
static inline int rfx_scalar_grid_for(const rfx_ctx *c, int ncols, int nx) {
    int per = rfx_scalar_wg_per_cu(ncols, nx) * (c->blocks_per_cu > 0 ? c->blocks_per_cu : 2) / 2;
    return c->num_cus * (per < 1 ? 1 : per);
}
static inline int rfx_scalar_grid(const rfx_ctx *c) { return c->num_cus * 16 * ((c->blocks_per_cu > 2 ? c->blocks_per_cu : 2) / 2); } // upper bound, for workspace sizing

int rfx_run_filter_aggr(rfx_ctx *c, Plan &P, rfx_partial_t *d_out) {
    // a plan with no columns at all (COUNT without predicate): give it a harmless column-free path
    int grid = rfx_scalar_grid_for(c, P.ncols, P.nx);
    const i64 tiles = P.nrows / (RFX_BLOCK * 4) + 1;
    if (tiles < grid) grid = (int)tiles;
    int rc = rfx_ws_reserve(c, (size_t)grid * 9 * sizeof(Acc));
    if (rc != RFX_OK) return rc;
    Acc *ws = (Acc *)c->d_ws;
    int na_stride = 0;
    RFX_KERNEL_BEGIN(c);
    if (rfx_rtc_filter_aggr(c, P, grid, ws, &na_stride) == RFX_OK) goto launched; // this plan has a kernel of its own (rfx_rtc.hip)
    switch (P.ncols) {
        case 0:
        case 1: rfx_launch_filter_aggr_nc1(c, P, grid, ws, &na_stride); break;
        case 2: rfx_launch_filter_aggr_nc2(c, P, grid, ws, &na_stride); break;
        case 3: rfx_launch_filter_aggr_nc3(c, P, grid, ws, &na_stride); break;
        case 4: rfx_launch_filter_aggr_nc4(c, P, grid, ws, &na_stride); break;
        case 5: rfx_launch_filter_aggr_nc5(c, P, grid, ws, &na_stride); break;
        case 6: rfx_launch_filter_aggr_nc6(c, P, grid, ws, &na_stride); break;
        case 7: rfx_launch_filter_aggr_nc7(c, P, grid, ws, &na_stride); break;
        default: rfx_launch_filter_aggr_nc8(c, P, grid, ws, &na_stride); break;
    }
launched:
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_filter_aggr_final, dim3(1), dim3(RFX_BLOCK), 0, c->stream, P, (const Acc *)ws, grid, na_stride, d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_filter_aggr(rfx_ctx_t *c, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                                   int nagg, int64_t nrows, int64_t row0, rfx_partial_t *d_out) {
    RFX_REQUIRE(c && d_out, RFX_EINVAL, "NULL argument");
    Plan P;
    int rc = rfx_plan_build(&P, preds, npred, logic, aggs, nagg, NULL, NULL, nrows, row0);
    if (rc == RFX_ELIMIT && nagg > 1) {
        // more distinct columns / expressions than one launch carries: two passes over the same selection.  The first
        // writes partials [0, h) plus its count into slot h, which the second overwrites with partial h (its own count
        // lands in slot nagg).
        const int h = nagg / 2;
        rc = rfx_hip_filter_aggr(c, preds, npred, logic, aggs, h, nrows, row0, d_out);
        if (rc != RFX_OK) return rc;
        return rfx_hip_filter_aggr(c, preds, npred, logic, aggs + h, nagg - h, nrows, row0, d_out + h);
    }
    if (rc != RFX_OK) return rc;
    if (P.ncols == 0) {
        // nothing to read: COUNT(s) over all rows.  Use a 1-element dummy so the kernel shape stays uniform.
        RFX_REQUIRE(npred == 0, RFX_EINVAL, "predicate without column");
        rc = rfx_ws_reserve(c, 64);
        if (rc != RFX_OK) return rc;
        P.cols[0] = (const u64 *)c->d_ws; // never dereferenced for rows: every aggregate is COUNT
        P.ncols = 1;
        // COUNT over nrows rows does not need the data; emit directly
        rfx_partial_t h[RFX_MAX_AGGS + 1];
        for (int a = 0; a <= nagg; a++) {
            rfx_partial_identity(&h[a]);
            h[a].cnt = nrows;
        }
        RFX_HIP_CHECK(hipMemcpyAsync(d_out, h, sizeof(rfx_partial_t) * (nagg + 1), hipMemcpyHostToDevice, c->stream));
        RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
        return RFX_OK;
    }
    return rfx_run_filter_aggr(c, P, d_out);
}

extern "C" int rfx_agg_input_type(const rfx_agg_t *a) {
    if (!a) return RFX_I64;
    if (a->nxnodes > 0 && a->xnodes) { // expression tree: a node is f64 when it divides or any operand is f64
        int f64[RFX_MAX_XNODES] = {0, 0, 0, 0};
        const int n = a->nxnodes < RFX_MAX_XNODES ? a->nxnodes : RFX_MAX_XNODES;
        for (int i = 0; i < n; i++) {
            const rfx_xoperand_t *o[2] = {&a->xnodes[i].l, &a->xnodes[i].r};
            int of[2];
            for (int j = 0; j < 2; j++) {
                if (o[j]->kind == RFX_XK_NODE) of[j] = (o[j]->node >= 0 && o[j]->node < i) ? f64[o[j]->node] : 0;
                else of[j] = o[j]->type == RFX_F64;
            }
            f64[i] = RFX_XOP_RESULT_F64(a->xnodes[i].op, of[0], of[1]);
        }
        return f64[n - 1] ? RFX_F64 : RFX_I64;
    }
    if (a->xop == RFX_X_NONE) return a->col_type;
    const int cf = a->col_type == RFX_F64, of = a->xrhs_type == RFX_F64, swap = (a->xflags & RFX_XF_SWAP) != 0;
    return RFX_XOP_RESULT_F64(a->xop, swap ? of : cf, swap ? cf : of) ? RFX_F64 : RFX_I64;
}

// ---------------- host-side partial algebra ----------------
extern "C" void rfx_partial_identity(rfx_partial_t *p) {
    memset(p, 0, sizeof(*p));
    p->pos = RFX_INF_I64_D;
}

static inline double bits_f64(int64_t b) { double d; memcpy(&d, &b, 8); return d; }

// Field-wise merge of two row ranges' partials.  f64 extrema travel as raw bits; they are ordered through the
// same order-preserving image the kernels use.
extern "C" void rfx_partial_merge(int kind, int col_type, rfx_partial_t *into, const rfx_partial_t *from) {
    switch (kind) {
        case RFX_AGG_SUM:
        case RFX_AGG_AVG:
            into->isum = (int64_t)((uint64_t)into->isum + (uint64_t)from->isum);
            into->fsum += from->fsum;
            into->cnt += from->cnt;
            break;
        case RFX_AGG_MIN:
        case RFX_AGG_MAX:
            if (from->cnt > 0) {
                if (into->cnt == 0) into->ext = from->ext;
                else {
                    i64 a = into->ext, b = from->ext;
                    if (col_type == RFX_F64) { a = rfx_f64_to_ord((u64)a); b = rfx_f64_to_ord((u64)b); }
                    if (kind == RFX_AGG_MIN ? (b < a) : (b > a)) into->ext = from->ext;
                }
            }
            into->cnt += from->cnt;
            break;
        case RFX_AGG_COUNT:
            into->cnt += from->cnt;
            break;
        case RFX_AGG_FIRST:
            if (from->pos < into->pos) {
                into->pos = from->pos;
                into->ext = from->ext;
                into->cnt = from->cnt;
            }
            break;
        default:
            break;
    }
}

// Scalar result rules: core/math.c:1837-2045 (folds), :2445-2526 (avg), core/ops.h:172-174 (FDIV*).
extern "C" int rfx_agg_finalize(int kind, int col_type, const rfx_partial_t *p, rfx_value_t *out) {
    if (!p || !out) return RFX_EINVAL;
    memset(out, 0, sizeof(*out));
    const bool f = (col_type == RFX_F64);
    switch (kind) {
        case RFX_AGG_SUM:
            if (f) { out->type = RFX_F64; out->f = p->fsum; }
            else { out->type = RFX_I64; out->i = p->isum; out->is_null = (p->isum == RFX_NULL_I64_D); }
            return RFX_OK;
        case RFX_AGG_MIN:
        case RFX_AGG_MAX:
            out->type = f ? RFX_F64 : RFX_I64;
            if (p->cnt == 0) {
                out->is_null = 1;
                if (f) out->i = (int64_t)RFX_NAN_BITS;
                else out->i = RFX_NULL_I64_D;
            } else out->i = p->ext;
            return RFX_OK;
        case RFX_AGG_COUNT:
            out->type = RFX_I64;
            out->i = p->cnt;
            return RFX_OK;
        case RFX_AGG_AVG:
            out->type = RFX_F64;
            if (f) {
                // FDIVF64(sum, (f64)cnt): cnt == 0 -> null
                if (p->cnt == 0 || p->fsum != p->fsum) { out->is_null = 1; out->i = (int64_t)RFX_NAN_BITS; }
                else out->f = p->fsum / (double)p->cnt;
            } else {
                // FDIVI64(sum, cnt): cnt == 0 or sum == NULL_I64 -> null
                if (p->cnt == 0 || p->isum == RFX_NULL_I64_D) { out->is_null = 1; out->i = (int64_t)RFX_NAN_BITS; }
                else out->f = (double)p->isum / (double)p->cnt;
            }
            return RFX_OK;
        case RFX_AGG_FIRST:
            out->type = f ? RFX_F64 : RFX_I64;
            if (p->pos == RFX_INF_I64_D) {
                out->is_null = 1;
                out->i = f ? (int64_t)RFX_NAN_BITS : RFX_NULL_I64_D;
            } else {
                out->i = p->ext;
                out->is_null = f ? (bits_f64(p->ext) != bits_f64(p->ext)) : (p->ext == RFX_NULL_I64_D);
            }
            return RFX_OK;
        default:
            return RFX_EINVAL;
    }
}

extern "C" int rfx_hip_filter_aggr_host(rfx_ctx_t *c, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                                        int nagg, int64_t nrows, rfx_value_t *values, int64_t *selected) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(nagg >= 0 && nagg <= RFX_MAX_AGGS, RFX_ELIMIT, "too many aggregates");
    size_t bytes = sizeof(rfx_partial_t) * (size_t)(nagg + 1);
    // device staging for the partials lives behind the block partials in the workspace
    int rc = rfx_ws_reserve(c, (size_t)rfx_scalar_grid(c) * 9 * sizeof(Acc) + bytes + 256);
    if (rc != RFX_OK) return rc;
    rfx_partial_t *d_out = (rfx_partial_t *)((char *)c->d_ws + (((size_t)rfx_scalar_grid(c) * 9 * sizeof(Acc) + 255) & ~(size_t)255));
    rc = rfx_hip_filter_aggr(c, preds, npred, logic, aggs, nagg, nrows, 0, d_out);
    if (rc != RFX_OK) return rc;
    rfx_partial_t *h = (rfx_partial_t *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int a = 0; a < nagg; a++) {
        rc = rfx_agg_finalize(aggs[a].kind, rfx_agg_input_type(&aggs[a]), &h[a], &values[a]);
        if (rc != RFX_OK) return rc;
    }
    if (selected) *selected = h[nagg].cnt;
    return RFX_OK;
}

// ---------------- K6: key scope ----------------
int rfx_part_scope_hist(rfx_ctx *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, i64 nrows, i64 *kmin, i64 *kmax,
                        i64 *seen); // rfx_group_part.hip
extern "C" int rfx_hip_scope_i64(rfx_ctx_t *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic,
                                 int64_t nrows, int64_t *min, int64_t *max, int64_t *count) {
    RFX_REQUIRE(c && d_key && min && max && count, RFX_EINVAL, "NULL argument");
    {
        // large inputs: one pass that also leaves the partition histogram for the group-by that usually follows
        int prc = rfx_part_scope_hist(c, d_key, preds, npred, logic, nrows, (i64 *)min, (i64 *)max, (i64 *)count);
        if (prc != RFX_ESTATE) return prc;
    }
    rfx_agg_t aggs[2] = {{d_key, RFX_I64, RFX_AGG_MIN}, {d_key, RFX_I64, RFX_AGG_MAX}};
    rfx_value_t v[2];
    int64_t sel = 0;
    int rc = rfx_hip_filter_aggr_host(c, preds, npred, logic, aggs, 2, nrows, v, &sel);
    if (rc != RFX_OK) return rc;
    // index_scope_i64 (core/index.c:376-435) treats a null key as the value INT64_MIN; our MIN skips nulls, so
    // detect "some selected key was null" through the non-null count and widen the scope exactly as a plain
    // signed min would.
    rfx_partial_t *h = (rfx_partial_t *)c->h_pin;
    *count = sel;
    *min = v[0].i;
    *max = v[1].i;
    if (sel > 0 && h[0].cnt < sel) {
        *min = RFX_NULL_I64_D;
        if (h[0].cnt == 0) *max = RFX_NULL_I64_D;
    }
    return RFX_OK;
}

int rfx_chunk_scope(rfx_ctx *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg, i64 nrows,
                    i64 *kmin, i64 *kmax, i64 *seen); // rfx_group_chunk.hip
extern "C" int rfx_hip_group_scope(rfx_ctx_t *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic,
                                   const rfx_agg_t *aggs, int nagg, int64_t nrows, int64_t *min, int64_t *max, int64_t *count) {
    RFX_REQUIRE(c && d_key && min && max && count, RFX_EINVAL, "NULL argument");
    if (aggs && nagg > 0) {
        int prc = rfx_chunk_scope(c, d_key, preds, npred, logic, aggs, nagg, nrows, (i64 *)min, (i64 *)max, (i64 *)count);
        if (prc != RFX_ESTATE) return prc;
    }
    return rfx_hip_scope_i64(c, d_key, preds, npred, logic, nrows, min, max, count);
}

// ---------------- K2: byte masks ----------------
// A wave owns 512 consecutive rows per step: lane l loads rows 2l, 2l+1 of each 128-row group (one 16-byte load per
// group and column, 1 KB contiguous per wave instruction).  The mask bytes leave TRANSPOSED: the step's eight ballots are wave-uniform
// words, lane l picks the two that hold rows 8l .. 8l + 7 (group l / 16, even and odd rows), spreads four bits of each into bytes and
// stores 8 bytes -- 512 contiguous bytes per wave instruction, non-temporal.  (Round 3 stored each lane's own two bytes as one 16-bit
// store: 2-byte stores reach 1.3-1.7 TB/s on this part, profiles/r03_write_probe.jsonl; m2 1.90 ms -> see DESIGN section 3.)
template <int NC>
__global__ __launch_bounds__(RFX_BLOCK) void k_cmp_mask(const Plan P, int8_t *__restrict__ out) {
    PredSet<1> S; // exactly one comparison: one descriptor set in SGPRs, not eight
    predset_load<1>(P, S);
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nfull = P.nrows / 512;
    for (i64 q = wave_id; q < nfull; q += nwaves) {
        const i64 base = q * 512 + lane * 2;
        u64 v[NC][8];
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u64x2 t = rfx_ld2(P.cols[c] + base + j * 128);
                v[c][2 * j] = t.x;
                v[c][2 * j + 1] = t.y;
            }
        }
        const unsigned m = eval_preds<NC, 8, 1>(S, v, 0xffu);
        u64 be = 0, bo = 0; // ballots of the even / odd rows of this lane's OUTPUT group (lane / 16)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u64 b0 = __ballot((m >> (2 * j)) & 1u), b1 = __ballot((m >> (2 * j + 1)) & 1u);
            const bool mine = (lane >> 4) == j;
            be = mine ? b0 : be;
            bo = mine ? b1 : bo;
        }
        const unsigned sh = 4u * ((unsigned)lane & 15u);
        const unsigned x0 = (unsigned)(be >> sh) & 15u, x1 = (unsigned)(bo >> sh) & 15u; // rows 8l, 8l+2, 8l+4, 8l+6 / 8l+1, ...
        const unsigned lo = (x0 & 1u) | ((x1 & 1u) << 8) | (((x0 >> 1) & 1u) << 16) | (((x1 >> 1) & 1u) << 24);
        const unsigned hi = ((x0 >> 2) & 1u) | (((x1 >> 2) & 1u) << 8) | (((x0 >> 3) & 1u) << 16) | (((x1 >> 3) & 1u) << 24);
        __builtin_nontemporal_store(((u64)hi << 32) | lo, (u64 *)(out + q * 512 + lane * 8));
    }
    // tail rows
    if (blockIdx.x == 0) {
        for (i64 r = nfull * 512 + threadIdx.x; r < P.nrows; r += RFX_BLOCK) {
            u64 v[NC][1];
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][0] = P.cols[c][r];
            out[r] = (int8_t)(eval_preds<NC, 1, 1>(S, v, 1u) & 1u);
        }
    }
}

extern "C" int rfx_hip_cmp_mask(rfx_ctx_t *c, const rfx_pred_t *pred, int64_t nrows, int8_t *d_mask) {
    RFX_REQUIRE(c && pred && (d_mask || nrows == 0), RFX_EINVAL, "NULL argument");
    if (nrows == 0) return RFX_OK;
    c->ext_i[0]++; // RFX_STAT_MASK_PASSES
    RFX_REQUIRE(((uintptr_t)d_mask & 7) == 0, RFX_EINVAL, "mask must be 8-byte aligned");
    Plan P;
    int rc = rfx_plan_build(&P, pred, 1, RFX_AND, NULL, 0, NULL, NULL, nrows, 0);
    if (rc != RFX_OK) return rc;
    int grid = c->num_cus * 8;
    if (P.ncols == 1) hipLaunchKernelGGL((k_cmp_mask<1>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, d_mask);
    else hipLaunchKernelGGL((k_cmp_mask<2>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, d_mask);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// acc[i] = acc[i] && next[i]  -- and_op_partial / or_op_partial, core/logic.c:34-86 (C `&&` / `||` -> 0/1 bytes)
__global__ __launch_bounds__(RFX_BLOCK) void k_mask_logic(int8_t *__restrict__ acc, const int8_t *__restrict__ next, int scalar, int logic, i64 n) {
    const i64 n8 = n / 8;
    const u64 ones = 0x0101010101010101ULL;
    for (i64 g = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; g < n8; g += (i64)gridDim.x * RFX_BLOCK) {
        u64 a = *(const u64 *)(acc + g * 8);
        u64 b = next ? *(const u64 *)(next + g * 8) : (scalar ? ones : 0ULL);
        // normalise any non-zero byte to 1
        u64 an = (((a & 0x7f7f7f7f7f7f7f7fULL) + 0x7f7f7f7f7f7f7f7fULL) | a) >> 7 & ones;
        u64 bn = (((b & 0x7f7f7f7f7f7f7f7fULL) + 0x7f7f7f7f7f7f7f7fULL) | b) >> 7 & ones;
        *(u64 *)(acc + g * 8) = (logic == RFX_AND) ? (an & bn) : (an | bn);
    }
    if (blockIdx.x == 0) {
        for (i64 r = n8 * 8 + threadIdx.x; r < n; r += RFX_BLOCK) {
            int a = acc[r] != 0, b = next ? (next[r] != 0) : (scalar != 0);
            acc[r] = (int8_t)((logic == RFX_AND) ? (a && b) : (a || b));
        }
    }
}

extern "C" int rfx_hip_mask_logic(rfx_ctx_t *c, int logic, int8_t *d_acc, const int8_t *d_next, int scalar, int64_t nrows) {
    RFX_REQUIRE(c && (d_acc || nrows == 0), RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(logic == RFX_AND || logic == RFX_OR, RFX_EINVAL, "logic must be RFX_AND or RFX_OR");
    if (nrows == 0) return RFX_OK;
    c->ext_i[0]++; // RFX_STAT_MASK_PASSES
    RFX_REQUIRE(((uintptr_t)d_acc & 7) == 0 && ((uintptr_t)d_next & 7) == 0, RFX_EINVAL, "masks must be 8-byte aligned");
    hipLaunchKernelGGL(k_mask_logic, dim3(rfx_grid(c)), dim3(RFX_BLOCK), 0, c->stream, d_acc, d_next, scalar, logic, (i64)nrows);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
