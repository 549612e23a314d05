// rfx_where_once_kernel.hpp -- the body of K3's one-pass `where` (see rfx_where_once.hip for what it does and why): a header so that
// rfx_rtc.hip can compile it for ONE plan at run time, like rfx_scalar_kernel.hpp's K1.
#pragma once
#include "rfx_scalar_kernel.hpp"

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

typedef i64 i64x2 __attribute__((ext_vector_type(2)));

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
// D: the plan's DESCRIPTORS, P: its run-time values (column pointers, row count, the predicates' atoms) -- the prebuilt kernels pass one
// plan for both, a kernel compiled at run time for one plan (rfx_rtc.hip) passes a constexpr D: the comparison's operator, domain and
// conversions become constants (96 -> 69 registers and no scratch for one i64 comparison, 1.92 -> 1.81 ms per 1e9 rows).
template <int NC, int NP>
__device__ __forceinline__ void where_once_body(const Plan &D, const Plan &P, const WoArgs &A) {
    __shared__ u64 bits[2][WO_DWAVES][WO_WCHUNKS * 8]; // a tile's selection, 1 bit per row ("pair-split 128": word 2g even rows, 2g + 1 odd rows of group g)
    __shared__ unsigned ring[WO_DWAVES][WO_RING];      // ids on their way out, as offsets from the tile's first row
    __shared__ unsigned wcnt[3][WO_DWAVES];
    __shared__ i64 s_tile[3], s_excl[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool ctrl = wv == WO_DWAVES;
    const i64 nchunks = (P.nrows + WO_CHUNK - 1) / WO_CHUNK, nwhole = P.nrows / WO_CHUNK;
    PredSet<NP> S;
    predset_load<NP>(D, S);
#pragma unroll
    for (int i = 0; i < NP; i++) S.p[i].rhs = P.preds[i].rhs_bits;
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
            // ids leave as NON-TEMPORAL stores, 16 bytes per lane once the output position sits on a 1 KB line: measured on 1e9 rows, 10 %
            // selected, the 0.8 GB of ids cost 0.39 ms on top of the 8 GB read (computed but not stored: nothing) -- plain 8-byte stores
            // 1.83 ms per query, non-temporal 1.75-1.78
            auto put = [&](unsigned k) __attribute__((always_inline)) { // k <= 64 ids, one per lane
                if ((unsigned)lane < k && gpos + lane < A.cap) __builtin_nontemporal_store(tbase + (i64)R[(head + lane) & (WO_RING - 1)], &A.out[gpos + lane]);
                gpos += k;
                head += k;
                fill -= k;
            };
            auto flush = [&]() __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the wave's ring writes have landed before other lanes read them
                while (fill >= 64) {
                    const unsigned mis = (unsigned)(((u64)(A.out + gpos) >> 3) & 127); // ids past the last 1 KB boundary of the output
                    if (mis != 0 || gpos + 128 > A.cap) put(64 - (mis & 63)); // a short store to the next 512-byte boundary, or a whole 512-byte one
                    else if (fill >= 128) {
                        i64x2 t;
                        t.x = tbase + (i64)R[(head + 2 * lane) & (WO_RING - 1)];
                        t.y = tbase + (i64)R[(head + 2 * lane + 1) & (WO_RING - 1)];
                        __builtin_nontemporal_store(t, (i64x2 *)(A.out + gpos + 2 * lane));
                        gpos += 128;
                        head += 128;
                        fill -= 128;
                    } else break; // on a line boundary with less than a line: wait for the rest
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
            while (fill > 0) { // (wave-uniform) what is left: at most two stores
                const unsigned room = 64 - (unsigned)(gpos & 63);
                put(fill < room ? fill : room);
            }
        }
        prev = tile;
    }
}

