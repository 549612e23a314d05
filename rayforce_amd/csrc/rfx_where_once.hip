// rfx_where_once.hip -- K3 `where` in ONE pass: predicates -> wave ballots -> decoupled look-back -> ascending row ids.
//
// ops_where (core/ops.c:254-273) counts the set bytes of a B8 mask, allocates, and walks the mask a second time; the mask itself was
// written by cmp_map / logic_map passes before (core/cmp.c:35-68, core/logic.c:34-86).  rfx_where.hip already fused the comparisons into
// the pass and kept the selection as 1 bit per row, but still went bitmap -> per-chunk counts -> three scan kernels -> emit: two kernels
// over the data and a bitmap round trip (1e9 rows, 10 % selected: 1.55 + 0.57 ms, 51-53 % of the 8.8 B/row roofline).
//
// Here a workgroup takes TILES of 65 536 rows in ticket order (one device atomic per tile): its four waves evaluate 16 384 rows each and
// keep the ballots as the tile's selection bits in LDS (8 KB); the tile's count is published as an AGGREGATE, wave 0 looks back over the
// predecessors' status words (64 tiles per load, until it meets an INCLUSIVE prefix) and publishes the tile's inclusive prefix; then every
// wave turns its bits into row ids through a wave-private LDS ring -- its run of the output is contiguous, so the ids leave as whole
// 512-byte-aligned stores.  Nothing but the ids is written and the input is read once; the selection never leaves the CU.
// Measured on 1e9 rows, 10 % selected (MI355X): the load phase alone 1.33 ms (6 TB/s), + look-back 0.25, + ids 0.3 = 2.0 ms per query
// against 2.17 for the two-pass form; 1 % selected 1.74 ms.  What the phases cost when run back to back in one wave is NOT hidden behind
// other waves' loads at 4-5 waves per SIMD (96-110 registers): the versions on the way here -- 16 384-row tiles with the look-back between
// the phases (2.10 ms), ballot arithmetic per 128-row group for the ids (0.5 ms for them alone) -- are described where they were replaced.
// The caller sizes the output by a sampled estimate (rfx_hip_where_estimate); the count is exact either way.
#include "rfx_where_once_kernel.hpp"
#include <math.h>
#include <stdlib.h>

// One translation unit per column count (make: -DWO_NC=1..4, the kernels; -DWO_NC=0, the host side): the instantiations compile in
// parallel (minutes each).
#ifndef WO_NC
#define WO_NC 0
#endif

int rfx_rtc_where_once(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid); // rfx_rtc.hip: RFX_OK = launched
void rfx_where_once_launch_nc1(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);
void rfx_where_once_launch_nc2(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);
void rfx_where_once_launch_nc3(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);
void rfx_where_once_launch_nc4(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);

#if WO_NC > 0
template <int NC, int NP>
__global__ __launch_bounds__(WO_T) __attribute__((amdgpu_waves_per_eu(WO_NC <= 2 ? 5 : 3))) void k_where_once(const Plan P, const WoArgs A) {
    where_once_body<NC, NP>(P, P, A);
}

#define WO_CAT2(a, b) a##b
#define WO_CAT(a, b) WO_CAT2(a, b)
void WO_CAT(rfx_where_once_launch_nc, WO_NC)(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid) {
    if (P.npred <= 1) hipLaunchKernelGGL((k_where_once<WO_NC, 1>), dim3(grid), dim3(WO_T), 0, c->stream, P, A);
    else hipLaunchKernelGGL((k_where_once<WO_NC, 4>), dim3(grid), dim3(WO_T), 0, c->stream, P, A);
}

#else // WO_NC == 0: estimate + entry points

// ---- the estimate: the predicates over 2^15 strided rows ----
#define WO_NSAMP (1 << 15)
template <int NC>
__global__ __launch_bounds__(RFX_BLOCK) void k_where_sample(const Plan P, i64 stride, i64 nsamp, unsigned *__restrict__ hits) {
    PredSet<4> S;
    predset_load<4>(P, S);
    unsigned h = 0;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < nsamp; i += (i64)gridDim.x * RFX_BLOCK) {
        u64 v[NC][1];
        bool valid[1] = {true}, sel[1];
#pragma unroll
        for (int c = 0; c < NC; c++) v[c][0] = P.cols[c][i * stride];
        eval_sel<NC, 1, 4>(S, v, valid, sel);
        h += sel[0] ? 1u : 0u;
    }
    for (int m = 32; m >= 1; m >>= 1) h += __shfl_xor(h, m, 64);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(hits, h);
}

extern "C" int rfx_hip_where_estimate(rfx_ctx_t *c, const rfx_pred_t *preds, int npred, int logic, int64_t nrows, int64_t *upper) {
    RFX_REQUIRE(c && upper, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nrows >= 0, RFX_EINVAL, "nrows < 0");
    *upper = nrows;
    if (nrows <= 4 * WO_NSAMP || npred == 0) return RFX_OK;
    Plan P;
    int rc = rfx_plan_build(&P, preds, npred, logic, NULL, 0, NULL, NULL, nrows, 0);
    if (rc != RFX_OK) return rc;
    if (P.ncols > 4 || P.npred > 4) return RFX_OK; // (the sampler has the one-pass kernel's shapes; wider filters: the whole column)
    // The estimate of the SAME predicates over the SAME columns is remembered (four per context; the sample kernel, its copy back and the
    // wait are 25 us of a 1.7 ms query).  It only sizes the id buffer: a remembered figure that no longer fits the data comes back from
    // rfx_hip_where_once as RFX_ELIMIT with the exact count, like any underestimate; uploads through the context drop it (rfx_hip_h2d).
    u64 sig = 0xCBF29CE484222325ULL;
    {
        auto mix = [&](u64 v) { sig = (sig ^ v) * 0x100000001B3ULL; sig ^= sig >> 29; };
        mix((u64)nrows); mix((u64)P.npred); mix((u64)P.logic); mix((u64)P.ncols);
        for (int i = 0; i < P.ncols; i++) mix((u64)(uintptr_t)P.cols[i]);
        for (int i = 0; i < P.npred; i++) {
            const PlanPred &q = P.preds[i];
            mix((u64)q.col | ((u64)(unsigned)q.rhs_col << 8) | ((u64)q.op << 16) | ((u64)q.dom_f64 << 20) | ((u64)q.lhs_cvt << 21) | ((u64)q.rhs_cvt << 22) | ((u64)q.more << 23) | ((u64)(unsigned)q.tree << 24));
            mix(q.rhs_bits);
        }
        if (sig == 0) sig = 1;
    }
    struct EstMemo { u64 sig; i64 upper; unsigned age, uses; }; // 4 x 24 bytes <= the 256 rfx_ctx.hip clears
    static_assert(4 * sizeof(EstMemo) <= 256, "memo block");
    if (!c->ext_p[6]) c->ext_p[6] = calloc(1, 256);
    EstMemo *memo = (EstMemo *)c->ext_p[6];
    static unsigned memo_clock = 0; // (only orders the entries of each context)
    for (int i = 0; memo && i < 4 && !getenv("RFX_NO_SAMPLE_MEMO"); i++)
        if (memo[i].sig == sig) {
            if (++memo[i].uses > 32u) { memo[i].sig = 0; break; } // (serves 32 queries, then the column is sampled again: staleness is bounded)
            memo[i].age = __atomic_add_fetch(&memo_clock, 1u, __ATOMIC_RELAXED);
            *upper = memo[i].upper;
            return RFX_OK;
        }
    rc = rfx_ws_reserve(c, 256);
    if (rc != RFX_OK) return rc;
    unsigned *hits = (unsigned *)c->d_ws;
    RFX_HIP_CHECK(hipMemsetAsync(hits, 0, 4, c->stream));
    const i64 stride = nrows / WO_NSAMP;
    const int grid = 32;
    switch (P.ncols) {
        case 1: hipLaunchKernelGGL(k_where_sample<1>, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, stride, (i64)WO_NSAMP, hits); break;
        case 2: hipLaunchKernelGGL(k_where_sample<2>, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, stride, (i64)WO_NSAMP, hits); break;
        case 3: hipLaunchKernelGGL(k_where_sample<3>, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, stride, (i64)WO_NSAMP, hits); break;
        default: hipLaunchKernelGGL(k_where_sample<4>, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, stride, (i64)WO_NSAMP, hits); break;
    }
    RFX_HIP_CHECK(hipGetLastError());
    unsigned *h = (unsigned *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, hits, 4, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    // the sampled fraction + four standard deviations of the sample + 1 % of the column: clustered selections can still exceed it --
    // rfx_hip_where_once then says how many ids there are and the caller runs it again
    const double f = (double)h[0] / (double)WO_NSAMP;
    const double sd = sqrt(f * (1.0 - f) / (double)WO_NSAMP);
    double up = ((f + 4.0 * sd + 0.01) * (double)nrows) + 1024.0;
    if (up > (double)nrows) up = (double)nrows;
    *upper = (int64_t)up;
    if (memo) { // replaces the entry used longest ago
        int at = 0;
        for (int i = 1; i < 4; i++)
            if (memo[i].age < memo[at].age) at = i;
        memo[at] = EstMemo{sig, *upper, __atomic_add_fetch(&memo_clock, 1u, __ATOMIC_RELAXED), 0u};
    }
    return RFX_OK;
}

extern "C" int rfx_hip_where_once(rfx_ctx_t *c, const rfx_pred_t *preds, int npred, int logic, int64_t nrows, int64_t row0, int64_t *d_ids,
                                  int64_t cap, int64_t *count) {
    RFX_REQUIRE(c && count, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nrows >= 0 && cap >= 0, RFX_EINVAL, "nrows / cap < 0");
    RFX_REQUIRE(npred > 0 && preds, RFX_EINVAL, "where_once takes predicates (a byte mask goes through where_begin / where_emit)");
    RFX_REQUIRE(cap == 0 || d_ids, RFX_EINVAL, "d_ids is NULL");
    *count = 0;
    c->where_n = -1; // a pending where_begin does not survive (the workspace is shared)
    if (nrows == 0) return RFX_OK;
    Plan P;
    int rc = rfx_plan_build(&P, preds, npred, logic, NULL, 0, NULL, NULL, nrows, 0);
    if (rc != RFX_OK) return rc;
    if (P.ncols > 4 || P.npred > 4 || (c->flags & RFX_TUNE_NO_WHERE_ONCE)) { // shapes the one-pass kernel is not instantiated for (and A/B): the two-pass form, same contract
        int64_t n = 0;
        rc = rfx_hip_where_begin(c, preds, npred, logic, NULL, nrows, &n);
        if (rc != RFX_OK) return rc;
        *count = n;
        if (n > cap) {
            rfx_set_error("where_once: %lld ids for a buffer of %lld", (long long)n, (long long)cap);
            return RFX_ELIMIT;
        }
        return rfx_hip_where_emit(c, row0, d_ids);
    }
    const i64 ntiles = (nrows + WO_TILE - 1) / WO_TILE;
    rc = rfx_ws_reserve(c, 256 + (size_t)ntiles * 8);
    if (rc != RFX_OK) return rc;
    WoArgs A;
    memset(&A, 0, sizeof(A));
    A.ticket = (unsigned *)c->d_ws;
    A.total = (i64 *)((char *)c->d_ws + 8);
    A.status = (u64 *)((char *)c->d_ws + 256);
    A.out = (i64 *)d_ids;
    A.cap = cap;
    A.row0 = row0;
    A.ntiles = ntiles;
    A.delay = 6; /* (0 .. 20 measured: 1.955 / 1.93 / 1.92 / 1.925 / 1.95 ms per 1e9 rows, 10 % selected) */
    RFX_HIP_CHECK(hipMemsetAsync(c->d_ws, 0, 256 + (size_t)ntiles * 8, c->stream));
    int grid = c->num_cus * (P.ncols <= 2 ? 4 : 2); // workgroups a CU holds (5 waves each)
    if ((i64)grid > ntiles) grid = (int)ntiles;
    c->ext_p[4] = (void *)((uintptr_t)c->ext_p[4] + 1); // RFX_STAT_WHERE_ONCE: k_where_once launches
    RFX_KERNEL_BEGIN(c);
    if (rfx_rtc_where_once(c, P, A, grid) != RFX_OK) { // the kernel compiled for this plan, when there is one; else the prebuilt instantiation
        switch (P.ncols) {
            case 1: rfx_where_once_launch_nc1(c, P, A, grid); break;
            case 2: rfx_where_once_launch_nc2(c, P, A, grid); break;
            case 3: rfx_where_once_launch_nc3(c, P, A, grid); break;
            default: rfx_where_once_launch_nc4(c, P, A, grid); break;
        }
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    i64 *h = (i64 *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, A.total, 8, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    *count = h[0];
    if (h[0] > cap) {
        rfx_set_error("where_once: %lld ids for a buffer of %lld", (long long)h[0], (long long)cap);
        return RFX_ELIMIT;
    }
    return RFX_OK;
}
#endif
