// rfx_where.hip -- K3 ordered stream compaction (`where`), K4 gather, and the chunk-count scan they share with
// the group-rank step.
//
// ops_where (core/ops.c:254-273) is a single-threaded count pass + emit pass over a byte mask.  Here:
//   pass A  one streaming read of the predicate columns (or of the byte mask): every wave evaluates 512 rows,
//           turns its per-lane results into 64-bit ballots and stores 8 bitmap words + one popcount.  Nothing else is
//           written: the selection lives in 1 bit per row (1/64 of an i64 column).
//   scan    exclusive prefix over the per-512-row counts (three tiny kernels).
//   pass B  reads only the bitmap (0.125 B/row) and writes the ascending row ids (8 B per selected row); the rank
//           of a row inside its wave comes from popcounts of the ballot words (mbcnt-style), no LDS, no barrier.
// Total traffic: 8 B/row in + 8 B/selected row out + 0.25 B/row bitmap, against the reference's 1 B/row mask
// written + read twice on top of the 8 B/row compare.
//
// Bitmap layout ("pair-split 128"): rows are grouped by 128; lane l of a wave owns rows 2l and 2l+1 of a group
// (one 16-byte load); word 2g holds the even rows of group g (bit l <-> row 128g + 2l), word 2g+1 the odd rows.
#include "rfx_scalar_kernel.hpp"

#define RFX_CHUNK 512 /* rows per wave step = 4 groups of 128 = 8 bitmap words */

__device__ __forceinline__ u64 lanemask_lt() {
    const unsigned l = threadIdx.x & 63;
    return (l == 0) ? 0ULL : (~0ULL >> (64 - l));
}

// ---------------- pass A: predicates -> bitmap + per-chunk counts ----------------
template <int NC, int NP>
__device__ __forceinline__ void sel_chunk_load(const Plan &P, i64 q, int lane, u64 (&v)[NC][8], bool (&valid)[8]) {
    const i64 base = q * RFX_CHUNK + lane * 2;
    if (q * RFX_CHUNK + RFX_CHUNK <= P.nrows) {
#pragma unroll
        for (int e = 0; e < 8; e++) valid[e] = true;
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u64x2 t = rfx_ld2(P.cols[c] + base + j * 128);
                v[c][2 * j] = t.x;
                v[c][2 * j + 1] = t.y;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const i64 row = base + (e >> 1) * 128 + (e & 1);
            valid[e] = row < P.nrows;
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][e] = valid[e] ? P.cols[c][row] : 0ULL;
        }
    }
}
template <int NC, int NP>
__device__ __forceinline__ u64 sel_chunk_words(const PredSet<NP> &S, int lane, int lane_off, const u64 (&v)[NC][8], const bool (&valid)[8], u64 mine) {
    bool sel[8];
    eval_sel<NC, 8, NP>(S, v, valid, sel);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const u64 b = __ballot(sel[e]); // the compare's lane mask itself
        mine = (lane == e + lane_off) ? b : mine;
    }
    return mine;
}

// A wave step = TWO ADJACENT 512-row chunks: 8 x 16-byte loads per lane per column in flight, and the 16 bitmap words of
// the pair leave as ONE aligned 128-byte store (lanes 0..15).  Writing the chunks' 64 bytes separately, plus an 8-byte
// count each, meant two partial cache lines per chunk (read-modify-write at the memory side): 1.5 ms for the 8 GB pass
// against 1.2 ms for the same read in K1.  The per-chunk counts come from k_chunk_counts over the bitmap afterwards.
template <int NC, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_sel_bitmap(const Plan P, u64 *__restrict__ bitmap, i64 *__restrict__ chunk_cnt) {
    (void)chunk_cnt;
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nchunks = (P.nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    // A wave step = FOUR adjacent chunk pairs (4096 rows): each pair's 16 bitmap words land in lanes 16i .. 16i+15 of `mine`, and the
    // 64 words leave as ONE 512-byte store (eight pairs with two stores per step: slower again, 1.59 ms).  Stores share the vector-memory counter with the loads the wave waits on: with one
    // 128-byte store per pair the pass took 1.53 ms, without any store 1.21 ms (measured) -- so store rarely, and whole.
    const i64 nquads = (nchunks + 7) / 8;
    for (i64 p4 = wave_id; p4 < nquads; p4 += nwaves) {
        u64 mine = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const i64 q = 8 * p4 + 2 * i, q2 = q + 1;
            if (q >= nchunks) break; // wave-uniform
            u64 v0[NC][8], v1[NC][8];
            bool ok0[8], ok1[8];
            sel_chunk_load<NC, NP>(P, q, lane, v0, ok0);
            if (q2 < nchunks) sel_chunk_load<NC, NP>(P, q2, lane, v1, ok1);
            mine = sel_chunk_words<NC, NP>(S, lane, 16 * i, v0, ok0, mine);
            if (q2 < nchunks) mine = sel_chunk_words<NC, NP>(S, lane, 16 * i + 8, v1, ok1, mine);
        }
        bitmap[p4 * 64 + lane] = mine; // the bitmap is sized in whole 4096-row steps; absent chunks store zeros
    }
}

#ifndef WH_NC // (the TUs that only instantiate k_sel_bitmap for one column count -- build/rfx_where_nc<k>.o -- stop before the rest)
// ---------------- pass A': byte mask -> bitmap + per-chunk counts ----------------
__global__ __launch_bounds__(RFX_BLOCK) void k_mask_bitmap(const int8_t *__restrict__ mask, i64 nrows, u64 *__restrict__ bitmap,
                                                         i64 *__restrict__ chunk_cnt) {
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    for (i64 q = wave_id; q < nchunks; q += nwaves) {
        const i64 base = q * RFX_CHUNK + lane * 2;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            i64 row = base + (e >> 1) * 128 + (e & 1);
            if (row < nrows) m |= (unsigned)(mask[row] != 0) << e;
        }
        u64 mine = 0;
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            u64 b = __ballot((m >> e) & 1u);
            cnt += __popcll(b);
            mine = (lane == e) ? b : mine;
        }
        if (lane < 8) bitmap[q * 8 + lane] = mine;
        if (lane == 0) chunk_cnt[q] = cnt;
    }
}

// ---------------- exclusive scan of the chunk counts (in place), total -> *total ----------------
#define SCAN_SPAN 2048 /* entries per workgroup in the first and third kernels */
__device__ __forceinline__ i64 block_exclusive_scan(i64 x, i64 *lds /* >= 4 entries */, i64 *block_total) {
    // inclusive wave scan
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i64 inc = x;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        i64 o = (i64)(((u64)(unsigned)__shfl_up((unsigned)(inc >> 32), s, 64) << 32) | (unsigned)__shfl_up((unsigned)inc, s, 64));
        if (lane >= s) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    i64 wbase = 0, tot = 0;
    for (int w = 0; w < RFX_BLOCK / RFX_WAVE; w++) {
        i64 t = lds[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return wbase + inc - x;
}

// (n_eff, when given: a device-side bound on n -- entries beyond it count as zero and are neither read nor written; rfx_rank_slots)
__global__ __launch_bounds__(RFX_BLOCK) void k_scan_partial(const i64 *__restrict__ cnt, i64 n, i64 *__restrict__ span_sum, const i64 *__restrict__ n_eff) {
    __shared__ i64 lds[4];
    const i64 base = (i64)blockIdx.x * SCAN_SPAN;
    if (n_eff && *n_eff < n) n = *n_eff;
    if (base >= n) { // (wave-uniform)
        if (threadIdx.x == 0) span_sum[blockIdx.x] = 0;
        return;
    }
    i64 s = 0;
    for (int i = threadIdx.x; i < SCAN_SPAN; i += RFX_BLOCK) {
        i64 idx = base + i;
        s += (idx < n) ? cnt[idx] : 0;
    }
    for (int m = 32; m >= 1; m >>= 1) s += (i64)rfx_shfl_xor_u64((u64)s, m);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) span_sum[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// single workgroup: exclusive scan of span sums (nspans arbitrary), total -> *total
__global__ __launch_bounds__(RFX_BLOCK) void k_scan_spans(i64 *__restrict__ span_sum, i64 nspans, i64 *__restrict__ total) {
    __shared__ i64 lds[4];
    i64 carry = 0;
    for (i64 b = 0; b < nspans; b += RFX_BLOCK) {
        i64 idx = b + threadIdx.x;
        i64 x = (idx < nspans) ? span_sum[idx] : 0;
        i64 tot;
        i64 ex = block_exclusive_scan(x, lds, &tot);
        if (idx < nspans) span_sum[idx] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(RFX_BLOCK) void k_scan_apply(i64 *__restrict__ cnt, i64 n, const i64 *__restrict__ span_off, const i64 *__restrict__ n_eff) {
    __shared__ i64 lds[4];
    const i64 base = (i64)blockIdx.x * SCAN_SPAN;
    if (n_eff && *n_eff < n) n = *n_eff;
    if (base >= n) return;
    i64 carry = span_off[blockIdx.x];
    // each thread owns SCAN_SPAN / RFX_BLOCK = 8 consecutive entries
    constexpr int PER = SCAN_SPAN / RFX_BLOCK;
    i64 x[PER], s = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        i64 idx = base + threadIdx.x * PER + i;
        x[i] = (idx < n) ? cnt[idx] : 0;
        s += x[i];
    }
    i64 tot;
    i64 ex = block_exclusive_scan(s, lds, &tot) + carry;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        i64 idx = base + threadIdx.x * PER + i;
        if (idx < n) cnt[idx] = ex;
        ex += x[i];
    }
}

// Exclusive scan of d_cnt[0..n) in place; total written to d_total (device).  Uses the tail of d_cnt's
// allocation?  No: span sums live in the context workspace.
int rfx_scan_counts(rfx_ctx *c, i64 *d_cnt, i64 n, i64 *d_total, const i64 *d_n_eff) {
    const i64 nspans = (n + SCAN_SPAN - 1) / SCAN_SPAN;
    int rc = rfx_ws_reserve(c, (size_t)(nspans + 8) * 8);
    if (rc != RFX_OK) return rc;
    i64 *spans = (i64 *)c->d_ws;
    if (nspans > 0) {
        hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nspans), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_cnt, n, spans, d_n_eff);
    }
    hipLaunchKernelGGL(k_scan_spans, dim3(1), dim3(RFX_BLOCK), 0, c->stream, spans, nspans, d_total);
    if (nspans > 0) {
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nspans), dim3(RFX_BLOCK), 0, c->stream, d_cnt, n, (const i64 *)spans, d_n_eff);
    }
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---------------- pass B: bitmap + scanned counts -> ascending ids ----------------
__global__ __launch_bounds__(RFX_BLOCK) void k_emit_ids(const u64 *__restrict__ bitmap, const i64 *__restrict__ chunk_off, i64 nrows,
                                                      i64 row0, i64 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    const u64 below = lanemask_lt();
    for (i64 q = wave_id; q < nchunks; q += nwaves) {
        i64 pos = chunk_off[q];
        const u64 *w = bitmap + q * 8;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const u64 w0 = w[2 * g], w1 = w[2 * g + 1];
            if ((w0 | w1) == 0) continue;
            const unsigned s0 = (unsigned)(w0 >> lane) & 1u, s1 = (unsigned)(w1 >> lane) & 1u;
            const i64 r = pos + __popcll(w0 & below) + __popcll(w1 & below);
            const i64 row = row0 + q * RFX_CHUNK + g * 128 + lane * 2;
            if (s0) out[r] = row;
            if (s1) out[r + s0] = row + 1;
            pos += __popcll(w0) + __popcll(w1);
        }
    }
}

// Write-combining form of pass B.  k_emit_ids stores each 128-row group's ids as two masked store instructions of ~6 active
// lanes at 10 % selectivity: every store is a partial cache line (0.81 ms for 0.8 GB of ids).  Here a wave owns a CONTIGUOUS
// range of chunks, so its output is one contiguous run: ids go through a 256-entry wave-private LDS ring and leave as whole,
// 512-byte-aligned 64-id stores (the first flush is short to reach alignment, the last one drains the ring).  Wave-private:
// no workgroup barrier, waves may run out of chunks independently.
#define EMIT_RING 256
__global__ __launch_bounds__(RFX_BLOCK) void k_emit_ids_wc(const u64 *__restrict__ bitmap, const i64 *__restrict__ chunk_off, i64 nrows, i64 row0,
                                                         i64 *__restrict__ out) {
    __shared__ i64 ring[RFX_BLOCK / RFX_WAVE][EMIT_RING];
    const int lane = threadIdx.x & 63;
    i64 *R = ring[threadIdx.x >> 6];
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    const i64 per = (nchunks + nwaves - 1) / nwaves;
    const i64 q0 = wave_id * per, q1 = (q0 + per < nchunks) ? q0 + per : nchunks;
    if (q0 >= q1) return;
    const u64 below = lanemask_lt();
    i64 gpos = chunk_off[q0]; // global output index of the ring's head
    unsigned head = 0, fill = 0;
    u64 nxt[8]; // the next chunk's bitmap words, fetched while the current chunk goes through the ring
#pragma unroll
    for (int i = 0; i < 8; i++) nxt[i] = bitmap[q0 * 8 + i];
    for (i64 q = q0; q < q1; q++) {
        u64 w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = nxt[i];
        if (q + 1 < q1) {
#pragma unroll
            for (int i = 0; i < 8; i++) nxt[i] = bitmap[(q + 1) * 8 + i];
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const u64 w0 = w[2 * g], w1 = w[2 * g + 1];
            if ((w0 | w1) == 0) continue;
            const unsigned s0 = (unsigned)(w0 >> lane) & 1u, s1 = (unsigned)(w1 >> lane) & 1u;
            const unsigned r = head + fill + (unsigned)(__popcll(w0 & below) + __popcll(w1 & below));
            const i64 row = row0 + q * RFX_CHUNK + g * 128 + lane * 2;
            if (s0) R[r & (EMIT_RING - 1)] = row;
            if (s1) R[(r + s0) & (EMIT_RING - 1)] = row + 1;
            fill += (unsigned)(__popcll(w0) + __popcll(w1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the wave's ring writes have landed before other lanes read them
            while (fill >= 64) {
                const unsigned k = 64 - (unsigned)(gpos & 63); // short first flush, then whole aligned 64-id lines
                if ((unsigned)lane < k) out[gpos + lane] = R[(head + lane) & (EMIT_RING - 1)];
                gpos += k;
                head += k;
                fill -= k;
            }
            asm volatile("" ::: "memory");
        }
    }
    if ((unsigned)lane < fill) out[gpos + lane] = R[(head + lane) & (EMIT_RING - 1)];
}

#endif
// k_sel_bitmap is instantiated per column count in its OWN translation unit (rfx_where.hip compiled with -DWH_NC=<k>: 24 instantiations in
// one TU took 20 minutes of a 16-way build); this TU calls them through rfx_launch_sel_bitmap<NC>
template <int NC>
void rfx_launch_sel_bitmap(rfx_ctx *c, const Plan &P, int grid)
#ifdef WH_NC
{
    if (P.npred <= 1) hipLaunchKernelGGL((k_sel_bitmap<NC, 1>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, c->d_bitmap, c->d_blksum);
    else if (P.npred <= 4) hipLaunchKernelGGL((k_sel_bitmap<NC, 4>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, c->d_bitmap, c->d_blksum);
    else hipLaunchKernelGGL((k_sel_bitmap<NC, RFX_MAX_PREDS>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, c->d_bitmap, c->d_blksum);
}
template void rfx_launch_sel_bitmap<WH_NC>(rfx_ctx *, const Plan &, int);
#else
;
extern template void rfx_launch_sel_bitmap<1>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<2>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<3>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<4>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<5>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<6>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<7>(rfx_ctx *, const Plan &, int);
extern template void rfx_launch_sel_bitmap<8>(rfx_ctx *, const Plan &, int);
#endif

#ifndef WH_NC
static int where_reserve(rfx_ctx *c, i64 nrows) {
    int rc = rfx_bitmap_reserve(c, ((nrows + 4095) / 4096) * 4096); // whole 4096-row wave steps of k_sel_bitmap
    if (rc != RFX_OK) return rc;
    // blksum is sized per 2048 rows by rfx_bitmap_reserve; we need one entry per 512 rows (+1 for the total)
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    if (c->blksum_cap < (size_t)nchunks + 2) {
        RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_blksum) RFX_HIP_CHECK(hipFree(c->d_blksum));
        c->d_blksum = NULL;
        c->blksum_cap = 0;
        RFX_HIP_CHECK(hipMalloc((void **)&c->d_blksum, ((size_t)nchunks + 2) * 8));
        c->blksum_cap = (size_t)nchunks + 2;
    }
    return RFX_OK;
}

// P must hold only the predicate columns (every column of P is loaded).
static void where_launch_bitmap(rfx_ctx *c, const Plan &P) {
    const i64 nchunks = (P.nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    int grid = c->num_cus * 4;
    if ((i64)grid * 4 > nchunks) grid = (int)((nchunks + 3) / 4);
    switch (P.ncols) {
        case 1: rfx_launch_sel_bitmap<1>(c, P, grid); break;
        case 2: rfx_launch_sel_bitmap<2>(c, P, grid); break;
        case 3: rfx_launch_sel_bitmap<3>(c, P, grid); break;
        case 4: rfx_launch_sel_bitmap<4>(c, P, grid); break;
        case 5: rfx_launch_sel_bitmap<5>(c, P, grid); break;
        case 6: rfx_launch_sel_bitmap<6>(c, P, grid); break;
        case 7: rfx_launch_sel_bitmap<7>(c, P, grid); break;
        default: rfx_launch_sel_bitmap<8>(c, P, grid); break;
    }
}

// per-chunk popcounts of the selection bitmap (full-line stores; the bitmap pass itself writes no counts)
__global__ __launch_bounds__(RFX_BLOCK) void k_chunk_counts(const u64 *__restrict__ bitmap, i64 nchunks, i64 *__restrict__ cnt) {
    for (i64 q = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; q < nchunks; q += (i64)gridDim.x * RFX_BLOCK) {
        const u64 *w = bitmap + q * 8;
        int s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += __popcll(w[i]);
        cnt[q] = s;
    }
}
static int where_scan_total(rfx_ctx *c, i64 nrows, i64 *count, bool counts_from_bitmap) {
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    if (counts_from_bitmap) {
        i64 cb = (nchunks + RFX_BLOCK - 1) / RFX_BLOCK;
        int grid = rfx_grid(c) * 4;
        if (cb < grid) grid = (int)cb;
        hipLaunchKernelGGL(k_chunk_counts, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)c->d_bitmap, nchunks, c->d_blksum);
        RFX_HIP_CHECK(hipGetLastError());
    }
    i64 *d_total = c->d_blksum + nchunks;
    int rc = rfx_scan_counts(c, c->d_blksum, nchunks, d_total, NULL);
    if (rc != RFX_OK) return rc;
    i64 *h = (i64 *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    *count = h[0];
    return RFX_OK;
}

// Internal (partitioned group-by under a selective filter): the predicates of `Pfull` -> selection bitmap + scanned
// per-chunk offsets in the context, *count = selected rows.  Invalidates a pending where_begin.  (syncs)
int rfx_where_bitmap_of_plan(rfx_ctx *c, const Plan &Pfull, i64 *count) {
    c->where_n = -1;
    int rc = where_reserve(c, Pfull.nrows);
    if (rc != RFX_OK) return rc;
    Plan Ph = Pfull; // keep only the columns the predicates read
    int map[RFX_MAX_COLS], nh = 0;
    for (int i = 0; i < RFX_MAX_COLS; i++) map[i] = -1;
    for (int i = 0; i < Pfull.npred; i++) {
        const int cs[2] = {Pfull.preds[i].col, Pfull.preds[i].rhs_col};
        for (int j = 0; j < 2; j++)
            if (cs[j] >= 0 && map[cs[j]] < 0) {
                map[cs[j]] = nh;
                Ph.cols[nh++] = Pfull.cols[cs[j]];
            }
    }
    for (int i = 0; i < Pfull.npred; i++) {
        Ph.preds[i].col = map[Pfull.preds[i].col];
        if (Pfull.preds[i].rhs_col >= 0) Ph.preds[i].rhs_col = map[Pfull.preds[i].rhs_col];
    }
    Ph.ncols = nh;
    Ph.nagg = 0;
    where_launch_bitmap(c, Ph);
    RFX_HIP_CHECK(hipGetLastError());
    return where_scan_total(c, Pfull.nrows, count, true);
}

// Internal: a selection bitmap is already in the context (written by the fused scope pass): per-chunk counts + scan.  (syncs)
int rfx_where_counts_of_bitmap(rfx_ctx *c, i64 nrows, i64 *count) {
    c->where_n = -1;
    int rc = where_reserve(c, nrows);
    if (rc != RFX_OK) return rc;
    return where_scan_total(c, nrows, count, true);
}

// Ordered compaction of whole columns by the context's bitmap: dst[c][r] = src[c][row] for the r-th selected row, and
// rows32[r] = row (local).  A lane fetches its 16-byte row pair only when one of the two rows is selected.
struct CompactArgs {
    const u64 *src[4];
    u64 *dst[4];
    unsigned *rows32;
};
template <int NCOL>
__global__ __launch_bounds__(RFX_BLOCK) void k_compact_cols(const u64 *__restrict__ bitmap, const i64 *__restrict__ chunk_off, i64 nrows,
                                                          const CompactArgs A) {
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    const u64 below = lanemask_lt();
    for (i64 q = wave_id; q < nchunks; q += nwaves) {
        i64 pos = chunk_off[q];
        const u64 *w = bitmap + q * 8;
        u64 ws[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ws[i] = w[i];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const u64 w0 = ws[2 * g], w1 = ws[2 * g + 1];
            if ((w0 | w1) == 0) continue;
            const unsigned s0 = (unsigned)(w0 >> lane) & 1u, s1 = (unsigned)(w1 >> lane) & 1u;
            const i64 r = pos + __popcll(w0 & below) + __popcll(w1 & below);
            const i64 row = q * RFX_CHUNK + g * 128 + lane * 2;
            if (s0 | s1) {
                const bool pair_ok = row + 1 < nrows;
#pragma unroll
                for (int c = 0; c < NCOL; c++) {
                    u64 x, y = 0;
                    if (pair_ok) {
                        const u64x2 t = rfx_ld2(A.src[c] + row);
                        x = t.x;
                        y = t.y;
                    } else x = A.src[c][row];
                    if (s0) A.dst[c][r] = x;
                    if (s1) A.dst[c][r + s0] = y;
                }
                if (s0) A.rows32[r] = (unsigned)row;
                if (s1) A.rows32[r + s0] = (unsigned)(row + 1);
            }
            pos += __popcll(w0) + __popcll(w1);
        }
    }
}

int rfx_where_compact_cols(rfx_ctx *c, i64 nrows, const u64 *const *src, u64 *const *dst, int ncol, unsigned *d_rows32) {
    RFX_REQUIRE(ncol >= 1 && ncol <= 4, RFX_ELIMIT, "1..4 columns");
    CompactArgs A;
    memset(&A, 0, sizeof(A));
    for (int i = 0; i < ncol; i++) {
        A.src[i] = src[i];
        A.dst[i] = dst[i];
    }
    A.rows32 = d_rows32;
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    int grid = c->num_cus * 16;
    if ((i64)grid * 4 > nchunks) grid = (int)((nchunks + 3) / 4);
    const u64 *bm = (const u64 *)c->d_bitmap;
    const i64 *off = (const i64 *)c->d_blksum;
    switch (ncol) {
        case 1: hipLaunchKernelGGL((k_compact_cols<1>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, bm, off, nrows, A); break;
        case 2: hipLaunchKernelGGL((k_compact_cols<2>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, bm, off, nrows, A); break;
        case 3: hipLaunchKernelGGL((k_compact_cols<3>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, bm, off, nrows, A); break;
        default: hipLaunchKernelGGL((k_compact_cols<4>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, bm, off, nrows, A); break;
    }
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_where_begin(rfx_ctx_t *c, const rfx_pred_t *preds, int npred, int logic, const int8_t *d_mask,
                                   int64_t nrows, int64_t *count) {
    RFX_REQUIRE(c && count, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nrows >= 0, RFX_EINVAL, "nrows < 0");
    c->where_n = -1;
    c->pc_bitmap = 0; // the context's bitmap is about to be overwritten: a scope pass's selection is gone
    *count = 0;
    if (nrows == 0) {
        c->where_n = 0;
        c->where_count = 0;
        return RFX_OK;
    }
    RFX_REQUIRE((d_mask != NULL) != (npred > 0), RFX_EINVAL, "give either predicates or a byte mask");
    int rc = where_reserve(c, nrows);
    if (rc != RFX_OK) return rc;
    const i64 nchunks = (nrows + RFX_CHUNK - 1) / RFX_CHUNK;
    RFX_KERNEL_BEGIN(c);
    if (d_mask) {
        int grid = c->num_cus * 4;
        if ((i64)grid * 4 > nchunks) grid = (int)((nchunks + 3) / 4);
        hipLaunchKernelGGL(k_mask_bitmap, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, d_mask, (i64)nrows, c->d_bitmap, c->d_blksum);
    } else {
        Plan P;
        rc = rfx_plan_build(&P, preds, npred, logic, NULL, 0, NULL, NULL, nrows, 0);
        if (rc != RFX_OK) return rc;
        where_launch_bitmap(c, P);
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    i64 total = 0;
    rc = where_scan_total(c, nrows, &total, d_mask == NULL);
    if (rc != RFX_OK) return rc;
    c->where_n = nrows;
    c->where_count = total;
    *count = total;
    return RFX_OK;
}

extern "C" int rfx_hip_where_emit(rfx_ctx_t *c, int64_t row0, int64_t *d_ids) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(c->where_n >= 0, RFX_ESTATE, "where_emit without a successful where_begin");
    if (c->where_count == 0) return RFX_OK;
    RFX_REQUIRE(d_ids, RFX_EINVAL, "d_ids is NULL");
    const i64 nchunks = (c->where_n + RFX_CHUNK - 1) / RFX_CHUNK;
    int grid = c->num_cus * 16;
    if ((i64)grid * 4 > nchunks) grid = (int)((nchunks + 3) / 4);
    if (c->flags & RFX_TUNE_NO_EMIT_WC)
        hipLaunchKernelGGL(k_emit_ids, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)c->d_bitmap, (const i64 *)c->d_blksum, c->where_n,
                           (i64)row0, (i64 *)d_ids);
    else
        hipLaunchKernelGGL(k_emit_ids_wc, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)c->d_bitmap, (const i64 *)c->d_blksum, c->where_n,
                           (i64)row0, (i64 *)d_ids);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---------------- K4: gather ----------------
__global__ __launch_bounds__(RFX_BLOCK) void k_gather8(const u64 *__restrict__ col, const i64 *__restrict__ ids, i64 m, u64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < m; i += (i64)gridDim.x * RFX_BLOCK) out[i] = col[ids[i]];
}

// ... several columns at the same ids in one launch: the ids are read once (the row-hash route's key columns at the groups' first rows: six gathers, five
// of whose index reads were as large as the gathered column)
struct GatherMany {
    const u64 *col[RFX_MAX_KEYS];
    u64 *out[RFX_MAX_KEYS];
};
template <int N>
__global__ __launch_bounds__(RFX_BLOCK) void k_gather_many(const GatherMany A, const i64 *__restrict__ ids, i64 m) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < m; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 r = ids[i];
        u64 v[N];
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = A.col[k][r];
#pragma unroll
        for (int k = 0; k < N; k++) A.out[k][i] = v[k];
    }
}
extern "C" int rfx_hip_gather_many(rfx_ctx_t *c, const void *const *d_cols, int ncols, const int64_t *d_ids, int64_t m, void *const *d_outs) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(ncols >= 1 && ncols <= RFX_MAX_KEYS && d_cols && d_outs, RFX_EINVAL, "1..RFX_MAX_KEYS columns");
    if (m <= 0) return RFX_OK;
    RFX_REQUIRE(d_ids, RFX_EINVAL, "NULL argument");
    GatherMany A;
    for (int k = 0; k < ncols; k++) {
        RFX_REQUIRE(d_cols[k] && d_outs[k], RFX_EINVAL, "NULL column");
        A.col[k] = (const u64 *)d_cols[k];
        A.out[k] = (u64 *)d_outs[k];
    }
    for (int k = ncols; k < RFX_MAX_KEYS; k++) A.col[k] = NULL, A.out[k] = NULL;
    const i64 blocks = (m + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c) * 4;
    if (blocks < grid) grid = (int)blocks;
#define GM(N) case N: hipLaunchKernelGGL((k_gather_many<N>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, A, (const i64 *)d_ids, (i64)m); break;
    switch (ncols) {
        GM(1) GM(2) GM(3) GM(4) GM(5) GM(6) GM(7)
        default: hipLaunchKernelGGL((k_gather_many<8>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, A, (const i64 *)d_ids, (i64)m); break;
    }
#undef GM
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_gather(rfx_ctx_t *c, const void *d_col, const int64_t *d_ids, int64_t m, void *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (m <= 0) return RFX_OK;
    RFX_REQUIRE(d_col && d_ids && d_out, RFX_EINVAL, "NULL argument");
    i64 blocks = (m + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c) * 4;
    if (blocks < grid) grid = (int)blocks;
    hipLaunchKernelGGL(k_gather8, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_col, (const i64 *)d_ids, (i64)m, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

#endif // WH_NC