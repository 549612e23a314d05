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
#include "rfx_scalar_kernel.hpp"
#include <math.h>
#include <stdlib.h>

// One translation unit per column count (make: -DWO_NC=1..4, the kernels; -DWO_NC=0, the host side): the instantiations compile in
// parallel (minutes each).
#ifndef WO_NC
#define WO_NC 0
#endif

#define WO_DWAVES 4                   /* data waves of a workgroup */
#define WO_T ((WO_DWAVES + 1) * RFX_WAVE) /* ... and one control wave */
#define WO_CHUNK 512                  /* rows per wave step, as everywhere: lane l holds rows 2l, 2l + 1 of four 128-row groups */
#define WO_WCHUNKS 32                 /* chunks per wave and tile */
#define WO_WROWS (WO_WCHUNKS * WO_CHUNK) /* 16 384 rows per wave and tile */
#define WO_TILE (WO_DWAVES * WO_WROWS) /* 65 536 rows per tile: 8 KB of selection bits in LDS */
#define WO_RING 1024                  /* ids of a wave on their way out: what is left of the last store (< 64) + one chunk (<= 512) */

#define WO_AGG (1ULL << 62)
#define WO_INC (2ULL << 62)
#define WO_VAL ((1ULL << 62) - 1)

struct WoArgs {
    u64 *status;      // [ntiles] 0: nothing yet | WO_AGG + the tile's count | WO_INC + count of all tiles up to and including this one
    unsigned *ticket; // next tile
    i64 *total;       // out: the inclusive prefix of the last tile
    i64 *out;
    i64 cap;          // ids beyond `cap` are counted, not written
    i64 row0;
    i64 ntiles;
    int delay;        // look-back starts this many 3.4-us naps after the tile's count went out
};

void rfx_where_once_launch_nc1(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);
void rfx_where_once_launch_nc2(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);
void rfx_where_once_launch_nc3(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);
void rfx_where_once_launch_nc4(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid);

#if WO_NC > 0
__device__ __forceinline__ u64 wo_lanemask_lt() {
    const unsigned l = threadIdx.x & 63;
    return (l == 0) ? 0ULL : (~0ULL >> (64 - l));
}

// A whole chunk: four 16-byte loads per lane and column, unconditional (the caller clamps q to a whole chunk of the column).
template <int NC>
__device__ __forceinline__ void wo_chunk_load_whole(const Plan &P, i64 q, int lane, u64 (&v)[NC][8]) {
    const i64 base = q * WO_CHUNK + lane * 2;
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u64x2 t = rfx_ld2(P.cols[c] + base + j * 128);
            v[c][2 * j] = t.x;
            v[c][2 * j + 1] = t.y;
        }
    }
}
// The column's ragged last chunk, row by row; returns the rows that exist (bit e = row e of this lane).
template <int NC>
__device__ __forceinline__ unsigned wo_chunk_load_ragged(const Plan &P, i64 q, int lane, u64 (&v)[NC][8]) {
    const i64 base = q * WO_CHUNK + lane * 2;
    unsigned vm = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const i64 row = base + (e >> 1) * 128 + (e & 1);
        const bool in = row < P.nrows;
        vm |= (unsigned)in << e;
#pragma unroll
        for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
    }
    return vm;
}
// the chunk's eight ballots, ballot e parked in lane `lane0 + e` of `word`; returns the number of selected rows (wave-uniform)
template <int NC, int NP>
__device__ __forceinline__ unsigned wo_chunk_bits(const PredSet<NP> &S, const u64 (&v)[NC][8], unsigned vm, int lane, int lane0, u64 &word) {
    bool ok[8], sel[8];
#pragma unroll
    for (int e = 0; e < 8; e++) ok[e] = (vm >> e) & 1u;
    eval_sel<NC, 8, NP>(S, v, ok, sel);
    unsigned n = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const u64 b = __ballot(sel[e]);
        n += (unsigned)__popcll(b);
        word = (lane == lane0 + e) ? b : word;
    }
    return n;
}

__device__ __forceinline__ u64 wo_uniform(u64 x) { // a wave-uniform 64-bit value into scalar registers
    return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
__device__ __forceinline__ unsigned wo_rank(u64 m) { // set bits of the (uniform) mask below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Workgroup = four DATA waves + one CONTROL wave, software-pipelined over tiles: while the data waves evaluate tile i (phase 1), the control
// wave draws the next ticket and looks back for tile i - 1 (whose aggregate it published right after the previous barrier); after the
// barrier the data waves write the ids of tile i - 1 (phase 3) and go straight on to tile i + 1.  A look-back (up to ~16 rounds of a
// device-scope load each, ~1 us a round) has a whole tile time (~90 us) to finish: measured, a kernel that waited for it between phase 1
// and phase 3 lost 0.3-0.5 ms per 1e9 rows to it.  ONE barrier per tile; LDS slots are double (bits) / triple (counts, tile ids,
// prefixes) buffered so that nobody needs a second one.
template <int NC, int NP>
__global__ __launch_bounds__(WO_T) __attribute__((amdgpu_waves_per_eu(WO_NC <= 2 ? 5 : 3))) void k_where_once(const Plan P, const WoArgs A) {
    __shared__ u64 bits[2][WO_DWAVES][WO_WCHUNKS * 8]; // a tile's selection, 1 bit per row ("pair-split 128": word 2g even rows, 2g + 1 odd rows of group g)
    __shared__ unsigned ring[WO_DWAVES][WO_RING];      // ids on their way out, as offsets from the tile's first row
    __shared__ unsigned wcnt[3][WO_DWAVES];
    __shared__ i64 s_tile[3], s_excl[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool ctrl = wv == WO_DWAVES;
    const i64 nchunks = (P.nrows + WO_CHUNK - 1) / WO_CHUNK, nwhole = P.nrows / WO_CHUNK;
    PredSet<NP> S;
    predset_load<NP>(P, S);
    if (tid == 0) s_tile[0] = (i64)atomicAdd(A.ticket, 1u);
    __syncthreads();
    i64 prev = A.ntiles; // the tile of the previous round (>= ntiles: none)
    u64 agg_prev = 0;    // (control wave) its count
    for (int it = 0;; it++) {
        const int b3 = it % 3, p3 = (it + 2) % 3, n3 = (it + 1) % 3;
        const i64 tile = s_tile[b3];
        const bool have = tile < A.ntiles, havep = prev < A.ntiles;
        if (!have && !havep) break;
        if (ctrl) {
            if (lane == 0) s_tile[n3] = have ? (i64)atomicAdd(A.ticket, 1u) : A.ntiles;
            if (havep) { // the previous tile's place in the output
                u64 excl = 0;
                // lane l looks at tiles j - l, j - 64 - l, j - 128 - l, j - 192 - l (four loads in flight: a round covers 256 predecessors --
                // with ~1 000 workgroups in flight the nearest tile that knows its prefix is up to that far back, and a device-scope load
                // under full HBM load takes microseconds); before tile 0 lies an inclusive prefix of zero
                // (no hurry: the answer is needed a whole tile time from now, and a look-back that starts at once finds its neighbours'
                // counts not published yet and polls -- a thousand control waves reading the same few status lines device-wide)
                for (int z = 0; z < A.delay; z++) __builtin_amdgcn_s_sleep(127);
                for (i64 j = prev - 1;; j -= 256) {
                    u64 s[4];
                    bool again = false, done;
                    u64 add;
                    do {
                        if (again) __builtin_amdgcn_s_sleep(127);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const i64 idx = j - 64 * k - lane;
                            s[k] = idx >= 0 ? __hip_atomic_load(&A.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : WO_INC;
                        }
                        again = false;
                        done = false;
                        add = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (done || again) continue; // wave-uniform
                            const u64 inc = __ballot((s[k] >> 62) == 2ULL);
                            const unsigned first_inc = inc ? (unsigned)__builtin_ctzll(inc) : 64u;   // the nearest one that knows its prefix
                            const u64 need = first_inc >= 63u ? ~0ULL : ((2ULL << first_inc) - 1ULL); // it, and every tile between it and us
                            if (__ballot((s[k] >> 62) == 0ULL) & need) again = true;                // somebody has not even counted yet
                            else {
                                add += ((unsigned)lane <= first_inc) ? (s[k] & WO_VAL) : 0ULL;
                                done = inc != 0;
                            }
                        }
                    } while (again);
                    for (int m = 32; m >= 1; m >>= 1) add += rfx_shfl_xor_u64(add, m);
                    excl += add;
                    if (done) break;
                }
                if (lane == 0) {
                    __hip_atomic_store(&A.status[prev], WO_INC | (excl + agg_prev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_excl[p3] = (i64)excl;
                    if (prev == A.ntiles - 1) *A.total = (i64)(excl + agg_prev);
                }
            }
        } else if (have) {
            // ---- phase 1: this wave's 16 384 rows -> selection bits in LDS + their count ----
            u64 *mybits = bits[it & 1][wv];
            const i64 q0 = (tile * WO_TILE + (i64)wv * WO_WROWS) / WO_CHUNK;
            unsigned cnt = 0; // wave-uniform
            if (q0 + WO_WCHUNKS <= nwhole) {
                // 32 whole chunks: the next chunk's loads are in flight while this one is evaluated (two register sets, the loop unrolled
                // by two so that they swap by name; the loads are unconditional -- past the wave's last chunk the load is clamped and
                // ignored -- so that the compiler counts what is in flight instead of waiting for everything)
                u64 va[NC][8], vb[NC][8];
                wo_chunk_load_whole<NC>(P, q0, lane, va);
#pragma unroll 1
                for (int i = 0; i < WO_WCHUNKS / 2; i++) {
                    u64 word = 0;
                    wo_chunk_load_whole<NC>(P, q0 + 2 * i + 1, lane, vb);
                    cnt += wo_chunk_bits<NC, NP>(S, va, 0xffu, lane, 0, word);
                    const i64 qn = q0 + 2 * i + 2;
                    wo_chunk_load_whole<NC>(P, qn < q0 + WO_WCHUNKS ? qn : q0, lane, va);
                    cnt += wo_chunk_bits<NC, NP>(S, vb, 0xffu, lane, 8, word);
                    if (lane < 16) mybits[16 * i + lane] = word;
                }
            } else {
                // the column's last tile: chunk by chunk, the ragged one row by row, absent ones as zeros
#pragma unroll 1
                for (int i = 0; i < WO_WCHUNKS / 2; i++) {
                    u64 word = 0;
#pragma unroll 1
                    for (int h = 0; h < 2; h++) {
                        const i64 q = q0 + 2 * i + h;
                        if (q >= nchunks) continue; // wave-uniform
                        u64 v[NC][8];
                        unsigned vm = 0xffu;
                        if (q < nwhole) wo_chunk_load_whole<NC>(P, q, lane, v);
                        else vm = wo_chunk_load_ragged<NC>(P, q, lane, v);
                        cnt += wo_chunk_bits<NC, NP>(S, v, vm, lane, 8 * h, word);
                    }
                    if (lane < 16) mybits[16 * i + lane] = word;
                }
            }
            if (lane == 0) wcnt[b3][wv] = cnt;
        }
        __syncthreads();
        if (ctrl) {
            if (have) { // this tile's count goes out at once: successors sum aggregates while we are still waiting for our own prefix
                agg_prev = 0;
#pragma unroll
                for (int w = 0; w < WO_DWAVES; w++) agg_prev += (u64)wcnt[b3][w];
                if (lane == 0) __hip_atomic_store(&A.status[tile], WO_AGG | agg_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (havep) {
            // ---- phase 3: the previous tile's bits -> ids.  A wave's rows are contiguous, so is its output run: the ids of a chunk go into
            // a wave-private ring (rank = popcounts of the uniform bit words: no atomics) and leave as whole 512-byte-aligned 64-id stores
            // (a short first store reaches the alignment, the last one drains the ring) ----
            const u64 *mybits = bits[(it + 1) & 1][wv];
            unsigned *R = ring[wv];
            i64 gpos = s_excl[p3];
#pragma unroll
            for (int w = 0; w < WO_DWAVES; w++)
                if (w < wv) gpos += (i64)wcnt[p3][w];
            unsigned head = 0, fill = 0;
            const i64 tbase = A.row0 + prev * WO_TILE + (i64)wv * WO_WROWS;
            auto flush = [&]() __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the wave's ring writes have landed before other lanes read them
                while (fill >= 64) {
                    const unsigned k = 64 - (unsigned)(gpos & 63); // short first store, then whole aligned 64-id lines
                    if ((unsigned)lane < k && gpos + lane < A.cap) A.out[gpos + lane] = tbase + (i64)R[(head + lane) & (WO_RING - 1)];
                    gpos += k;
                    head += k;
                    fill -= k;
                }
                asm volatile("" ::: "memory");
            };
            // A LANE per 128-row group (its even-row and odd-row words), 64 groups = 8 192 rows a round: the lane walks the set bits of its
            // two words and writes its ids at (exclusive scan of the groups' counts) into the ring.  The walk costs ~25 instructions per
            // selected row PAIR of the busiest lane, not ~25 per 128 rows whatever they hold (a first version went group by group with
            // wave-wide ballot arithmetic: 0.5 ms per 1e9 rows, as much as two fifths of the load phase).  A round that selects more than
            // the ring holds goes through it in windows.
#pragma unroll 1
            for (int round = 0; round < WO_WCHUNKS * 4 / 64; round++) {
                const unsigned g = (unsigned)round * 64u + (unsigned)lane;
                const u64 w0 = mybits[2 * g], w1 = mybits[2 * g + 1];
                const unsigned n = (unsigned)(__popcll(w0) + __popcll(w1));
                unsigned inc = n;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const unsigned t = (unsigned)__shfl_up((int)inc, sft, 64);
                    if (lane >= sft) inc += t;
                }
                const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
                unsigned off = inc - n; // index, within the round, of this lane's next id
                u64 m = w0 | w1;
                const unsigned grow = g * 128u;
                unsigned wstart = 0; // ids of the round below this index are in the ring (or out)
                while (wstart < total) { // wave-uniform
                    const unsigned wend = wstart + (WO_RING - fill);
                    const unsigned at = head + fill - wstart;
                    while (m != 0) {
                        const unsigned b = (unsigned)__builtin_ctzll(m);
                        const unsigned e0 = (unsigned)(w0 >> b) & 1u, e1 = (unsigned)(w1 >> b) & 1u;
                        if (off + e0 + e1 > wend) break; // the window is full: this pair waits for the next one
                        m &= m - 1;
                        const unsigned row = grow + 2u * b;
                        if (e0) R[(at + off) & (WO_RING - 1)] = row;
                        if (e1) R[(at + off + e0) & (WO_RING - 1)] = row + 1u;
                        off += e0 + e1;
                    }
                    // how far the window got: the first id somebody still holds back (lanes before that one are through, lanes after it
                    // have not started)
                    unsigned nxt = m != 0 ? off : total;
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1) {
                        const unsigned o = (unsigned)__shfl_xor((int)nxt, sft, 64);
                        nxt = o < nxt ? o : nxt;
                    }
                    fill += nxt - wstart;
                    wstart = nxt;
                    flush();
                }
            }
            if ((unsigned)lane < fill && gpos + lane < A.cap) A.out[gpos + lane] = tbase + (i64)R[(head + lane) & (WO_RING - 1)];
        }
        prev = tile;
    }
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
    switch (P.ncols) {
        case 1: rfx_where_once_launch_nc1(c, P, A, grid); break;
        case 2: rfx_where_once_launch_nc2(c, P, A, grid); break;
        case 3: rfx_where_once_launch_nc3(c, P, A, grid); break;
        default: rfx_where_once_launch_nc4(c, P, A, grid); break;
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
