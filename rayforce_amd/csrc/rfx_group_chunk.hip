// rfx_group_chunk.hip -- ONE-PASS radix partitioning for the dense group-by whose tables do not fit one workgroup's LDS
// (BASELINE C3: 1e9 rows, 1e6 i64 keys, sum(f64); C3w: the same under `where a < k`).
//
// rfx_group_part.hip needs the exact size of every (workgroup, partition) region before it can scatter, i.e. a histogram
// pass over the key column in front (8 B/row), and the key scope (index_scope_i64, core/index.c:376-435) before that.  Here
// the scope pass IS the scatter:
//
//   k_chunk_scatter   reads predicate + key + value columns ONCE; partition = key & 255 needs no kmin, the record header
//                     keeps (key >> 8) mod 2^32 instead of a kmin-relative slot, and space comes in CHUNKS of 256 records
//                     (4 KB) that a (workgroup, partition) takes from its workgroup's slab (slabs of 256 chunks come from one
//                     device-wide cursor: ~1 returning global atomic per 32 tiles).  Same tile sort + 128-byte write
//                     combining as k_part_scatter_wc.  Side product: min / max / count of the selected keys.
//   k_chunk_offsets   per partition: exclusive scan of the chunk counts over workgroups (one 1024-lane workgroup)
//   k_chunk_place     chunk list of every partition, ordered by (workgroup, allocation order) -- no atomics
//   k_chunk_aggregate one or two workgroups per partition walk its chunk list and aggregate into LDS tables
//                     (slot = kh + ((p - kmin) >> 8), exact in 32-bit modular arithmetic), then merge into the global tables.
//
// Bytes per row: 16 read + 16 written + 16 read = 48 (was 8 + 16 + 16 + 16 = 56 plus a second scope read under a filter);
// under a 10 % filter 24 + 1.6 + 1.6.  The host learns the scope from the same pass, decides dense / sparse exactly as before
// (core/index.c:2013) and rfx_hip_group_dense_accumulate then finds the partitions ready (or falls back to the column passes
// when the tables turn out not to fit: nothing here changes a result).
#include "rfx_part_common.hpp"

#define CK_FREE 0xFFFFFFFFFFFFFFFFULL
#define CK_PARTS 256
#define CK_SLAB_BYTES (1u << 20) /* a workgroup takes chunks from the device-wide cursor one megabyte at a time */

struct ChunkArgs {
    int key_idx, vcol, nwg, chs; // chs: log2(records per chunk), 8..12
    i64 tiles_per_wg;            // in 2048-row tiles; 0: grid-stride (tile t of workgroup b = b + t * nwg)
    u64x2 *pool;         // chunk c = records [c << chs, (c + 1) << chs)
    u64 *meta;           // per chunk: partition | workgroup << 8 | records << 20 | ordinal within (workgroup, partition) << 40
    unsigned *ctl;       // [0] chunk cursor, [1] pool exhausted
    unsigned *wcount;    // [nwg][256] chunks of (workgroup, partition); k_chunk_offsets turns them into exclusive offsets
    u64 *part_start;     // [257] first chunk-list entry of every partition
    u64 *plist;          // chunk lists: chunk | records << 32
    ScopePart *parts;    // [nwg]
    unsigned max_chunks;
};

// Per-workgroup state of the chunk-allocating write combiner.  Lane p OWNS partition p: its running state (records carried, room
// in the current chunk, chunk count ...) lives in that lane's registers (CkLane); LDS holds only what the other lanes need to place
// a record -- one 16-byte read each: pi[p] = {exclusive batch offset, carried, first place in the flush order, room},
// di[p] = {cursor, first new chunk}.
struct CkDst {
    u64 cursor;    // next record index inside the partition's current chunk (pool coordinates)
    unsigned noff; // this batch: first new chunk of the partition, relative to tile_alloc
    unsigned _pad;
};
#define CK_MAX_GROUPS ((PART_TILE_ROWS + CK_PARTS * (WC_B - 1)) / WC_B) /* whole 128-byte groups one batch can complete */
#define CK_REL 0x80000000u
struct CkLds {
    unsigned cnt[CK_PARTS];  // per batch: record count per partition (rank atomics)
    unsigned excl[CK_PARTS]; // exclusive offset of the partition's run in the staged batch
    // one descriptor per flushing 128-byte group -- x, y: pool index of its first record (y & CK_REL: relative to the batch's first
    // new chunk), z: staging index of its record 0 (may be negative: the first `cut` records come from the carry buffer instead),
    // w: cut | partition << 8
    uint4 gd[CK_MAX_GROUPS + 32];
    u64x2 carry[CK_PARTS][WC_B];
    unsigned scan_w[RFX_BLOCK / RFX_WAVE];
    unsigned tile_groups, tile_alloc, slab_next, slab_end, dead;
    ScopePart red[RFX_BLOCK / RFX_WAVE];
};
struct CkLane {
    unsigned pre, room, nch, cur; // carried records, room in the current chunk (0 only before the first batch), chunks so far, current chunk
    u64 cursor;
    unsigned x, pf, noff, excl;   // this batch: records, records flushing, first new chunk, offset of the partition's run
};

template <typename LDS>
__device__ __forceinline__ void ck_init(LDS &L, CkLane &M) {
    M.pre = M.room = M.nch = M.cur = 0;
    M.cursor = 0;
    M.x = M.pf = M.noff = M.excl = 0;
    if (threadIdx.x == 0) {
        L.slab_next = 0;
        L.slab_end = 0;
        L.dead = 0;
    }
}
__device__ __forceinline__ u64 ck_meta(unsigned p, unsigned wg, unsigned nrec, unsigned ord) {
    return (u64)p | ((u64)wg << 8) | ((u64)nrec << 20) | ((u64)ord << 40);
}
// LDS-only barrier: the phases exchange data through LDS alone, global loads / stores in flight need not land first
__device__ __forceinline__ void ck_barrier() { __syncthreads(); }

// thread RFX_BLOCK-1 / CK_PARTS-1, with the batch's chunk request: a slab refill when the workgroup's slab is used up
template <typename LDS>
__device__ __forceinline__ void ck_request(LDS &L, const ChunkArgs &A, unsigned tn) {
    if (!tn) return;
    const unsigned sn = L.slab_next;
    if (sn + tn > L.slab_end) { // the rest of the old slab stays unused: its chunk metas keep CK_FREE
        const unsigned slab = (CK_SLAB_BYTES / 16u) >> A.chs;
        const unsigned g = tn > slab ? tn : slab;
        const unsigned fresh = atomicAdd(&A.ctl[0], g);
        // the returned value is consumed INSIDE this branch: a use after the join makes the compiler wait for every vector-memory
        // operation in flight (the next tile's loads among them) on every batch, refill or not
        asm volatile("" ::"v"(fresh));
        L.slab_end = fresh + g;
        if (fresh + g > A.max_chunks || fresh + g < fresh) {
            L.dead = 1;
            atomicExch(&A.ctl[1], 1u);
        }
        L.tile_alloc = fresh;
        L.slab_next = fresh + tn;
    } else {
        L.tile_alloc = sn;
        L.slab_next = sn + tn;
    }
}

// One lane per partition, after the batch's per-partition counts are in cnt[]: exclusive scan of (count | flushing groups << 12 |
// new chunks << 21), the flush decision, the chunk request, the partition of every flushing group.  Contains one barrier; the
// caller adds one after it.  cnt[] is zero again afterwards.
__device__ __forceinline__ void ck_plan(CkLds &L, CkLane &M, const ChunkArgs &A) {
    const int tid = threadIdx.x;
    const unsigned x = L.cnt[tid];
    L.cnt[tid] = 0;
    const unsigned pf = ((M.pre + x) / WC_B) * WC_B; // whole 128-byte groups leave, the rest waits in the carry buffer
    const unsigned rm = M.room;
    const unsigned need = (pf >= rm) ? ((pf - rm) >> A.chs) + 1 : 0;
    const unsigned packed = x | ((pf / WC_B) << 12) | (need << 21); // sums: <= 2048, <= CK_MAX_GROUPS (480), <= 256 + 2048 / 512
    unsigned inc = packed;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned o = __shfl_up(inc, d, 64);
        if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) L.scan_w[tid >> 6] = inc;
    ck_barrier();
    unsigned wbase = 0;
    for (int w = 0; w < (tid >> 6); w++) wbase += L.scan_w[w];
    const unsigned excl = wbase + inc - packed;
    const unsigned g0 = (excl >> 12) & 0x1FFu;
    M.x = x;
    M.pf = pf;
    M.noff = excl >> 21;
    M.excl = excl & 0xFFFu;
    L.excl[tid] = M.excl;
    for (unsigned g = 0; g < pf / WC_B; g++) {
        const unsigned pos0 = g * WC_B;
        const u64 dst = pos0 < rm ? M.cursor + pos0 : (((u64)M.noff << A.chs) + (pos0 - rm)) | ((u64)CK_REL << 32);
        L.gd[g0 + g] = make_uint4((unsigned)dst, (unsigned)(dst >> 32), M.excl - M.pre + pos0, (g == 0 ? M.pre : 0u) | ((unsigned)tid << 8));
    }
    if (tid == RFX_BLOCK - 1) {
        const unsigned tot = wbase + inc;
        L.tile_groups = (tot >> 12) & 0x1FFu;
        ck_request(L, A, tot >> 21);
    }
}
// where record `pos` of partition p's flush sequence goes (pos < pfl), given pi[p]
template <typename LDS>
__device__ __forceinline__ u64 ck_dst(const LDS &L, const ChunkArgs &A, unsigned p, unsigned pos, unsigned rm) {
    const CkDst d = L.di[p];
    return (pos < rm) ? d.cursor + pos : (((u64)L.tile_alloc + d.noff) << A.chs) + (pos - rm);
}
// one lane per partition, after the batch's records are placed
template <typename LDS>
__device__ __forceinline__ void ck_update(CkLane &M, const LDS &L, const ChunkArgs &A) {
    const int tid = threadIdx.x;
    const unsigned CH = 1u << A.chs;
    const unsigned pf = M.pf, rm = M.room;
    if (pf >= rm) { // the current chunk is full (or there is none yet): the partition moves on to its new chunks
        const unsigned k = ((pf - rm) >> A.chs) + 1;
        const unsigned first_new = L.tile_alloc + M.noff;
        for (unsigned j = 0; j < k; j++) A.meta[first_new + j] = ck_meta(tid, blockIdx.x, CH, M.nch + j);
        M.nch += k;
        const unsigned last = first_new + k - 1;
        const unsigned used = (pf - rm) - ((k - 1) << A.chs);
        M.cur = last;
        M.room = CH - used;
        M.cursor = ((u64)last << A.chs) + used;
    } else {
        M.room = rm - pf;
        M.cursor += pf;
    }
    M.pre = M.pre + M.x - pf;
}
// tails (one 128-byte group per partition that still carries records: its chunk has room, room > 0 after every batch), the
// final record count of every open chunk, the per-partition chunk counts and the workgroup's scope
template <bool CARRY, typename LDS>
__device__ __forceinline__ void ck_finish(LDS &L, const CkLane &M, const ChunkArgs &A, i64 mn, i64 mx, i64 sel, i64 nulls) {
    const int tid = threadIdx.x;
    const unsigned CH = 1u << A.chs;
    if constexpr (CARRY) {
        if (M.pre > 0) {
            for (unsigned j = 0; j < WC_B; j++) A.pool[M.cursor + j] = L.carry[tid][j]; // entries j >= pre are padding: the chunk's record count ends before them
        }
    }
    if (M.nch > 0) A.meta[M.cur] = ck_meta(tid, blockIdx.x, CH - M.room + M.pre, M.nch - 1);
    A.wcount[(size_t)blockIdx.x * CK_PARTS + tid] = M.nch;
    for (int s = 32; s >= 1; s >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s), omx = (i64)rfx_shfl_xor_u64((u64)mx, s);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sel += (i64)rfx_shfl_xor_u64((u64)sel, s);
        nulls += (i64)rfx_shfl_xor_u64((u64)nulls, s);
    }
    if ((tid & 63) == 0) L.red[tid >> 6] = ScopePart{mn, mx, sel, nulls};
    __syncthreads();
    if (tid == 0) {
        ScopePart r = L.red[0];
        for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) {
            r.mn = L.red[w].mn < r.mn ? L.red[w].mn : r.mn;
            r.mx = L.red[w].mx > r.mx ? L.red[w].mx : r.mx;
            r.sel += L.red[w].sel;
            r.nulls += L.red[w].nulls;
        }
        A.parts[blockIdx.x] = r;
    }
}

// 8 rows per lane of a full 2048-row tile as four 16-byte loads per column
template <int NC>
__device__ __forceinline__ void ck_load_full(const Plan &P, i64 tile, u64 (&v)[NC][8]) {
    const i64 base = tile * PART_TILE_ROWS + threadIdx.x * 2;
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u64x2 q = rfx_ld2(P.cols[c] + base + (i64)j * (RFX_BLOCK * 2));
            v[c][2 * j] = q.x;
            v[c][2 * j + 1] = q.y;
        }
    }
}
// the same for any tile; bit e of the result = row e exists
template <int NC>
__device__ __forceinline__ unsigned ck_load_tile(const Plan &P, i64 tile, u64 (&v)[NC][8]) {
    const i64 base = tile * PART_TILE_ROWS + threadIdx.x * 2;
    unsigned valid = 0xffu;
    if ((tile + 1) * PART_TILE_ROWS <= P.nrows) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u64x2 q = rfx_ld2(P.cols[c] + base + (i64)j * (RFX_BLOCK * 2));
                v[c][2 * j] = q.x;
                v[c][2 * j + 1] = q.y;
            }
        }
    } else {
        valid = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const i64 row = base + (i64)(e >> 1) * (RFX_BLOCK * 2) + (e & 1);
            const bool in = row < P.nrows;
            valid |= (unsigned)in << e;
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
        }
    }
    return valid;
}

// ---- every (or most) rows selected: 2048-row tiles sorted by partition in LDS, as k_part_scatter_wc ----
template <int NC, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_chunk_scatter(const Plan P, const ChunkArgs A) {
    __shared__ CkLds L;
    __shared__ u64x2 stag[PART_TILE_ROWS];
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x;
    CkLane M;
    ck_init(L, M);
    L.cnt[tid] = 0;
    ck_barrier();
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0, nulls = 0;
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    const i64 step = A.tiles_per_wg ? 1 : (i64)gridDim.x;
    const i64 t0 = A.tiles_per_wg ? (i64)blockIdx.x * A.tiles_per_wg : (i64)blockIdx.x;
    const i64 t1 = A.tiles_per_wg ? ((t0 + A.tiles_per_wg < ntiles) ? t0 + A.tiles_per_wg : ntiles) : ntiles;
    const int vc = A.vcol;
    // Two register sets, loop unrolled by two: the next tile's loads are issued before this tile's LDS phases and land while the
    // phases run (copying an in-flight set into "the current one" would wait for it: the roles alternate instead).
    u64 va[NC][8], vb[NC][8];
    bool alive = true;
    auto tile_body = [&](const u64 (&v)[NC][8], const unsigned valid, const i64 t) {
        const unsigned m = (NP == 0) ? valid : eval_preds<NC, 8, NP>(S, v, valid);
        u64 key[8], val[8];
        sel_col<NC, 8>(key, v, A.key_idx);
        sel_col<NC, 8>(val, v, vc);
        unsigned rank[8];
        const i64 base = t * PART_TILE_ROWS + tid * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            rank[e] = 0;
            if ((m >> e) & 1u) {
                const i64 k = (i64)key[e];
                rank[e] = atomicAdd(&L.cnt[key[e] & 255ULL], 1u);
                sel++;
                if (k == RFX_NULL_I64_D) nulls++;
                else {
                    mn = k < mn ? k : mn;
                    mx = k > mx ? k : mx;
                }
            }
        }
        ck_barrier();
        ck_plan(L, M, A);
        ck_barrier();
        if (L.dead) { // pool exhausted: the host sees ctl[1] and runs the column passes instead (uniform exit)
            alive = false;
            return;
        }
        // stage this tile's records in partition order
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (!((m >> e) & 1u)) continue;
            const unsigned p = (unsigned)(key[e] & 255ULL);
            const unsigned idx = L.excl[p] + rank[e];
            const u64 lrow = (u64)(base + (i64)(e >> 1) * (RFX_BLOCK * 2) + (e & 1));
            u64x2 r;
            r.x = (lrow << 32) | (u64)(unsigned)((i64)key[e] >> 8);
            r.y = val[e];
            stag[idx] = r;
        }
        ck_barrier();
        // Whole 128-byte groups leave: eight neighbouring lanes store one group = one full line, its first records from the carry
        // buffer, the rest from the staged tile.  (Carry and new records stored separately are two partial-line writes per group:
        // measured on C3, 1e9 records, those were 5 ms of the kernel's 8.)
        // The loop is unrolled over a whole tile's worth of places with the LDS reads of all eight issued together: one
        // descriptor + one record per lane and place, two dependent LDS latencies per tile instead of two per place.
        const unsigned nflush = L.tile_groups * WC_B;
        const u64 fresh = (u64)L.tile_alloc << A.chs;
        for (unsigned i0 = 0; i0 < nflush; i0 += PART_TILE_ROWS) {
            uint4 d[PART_TILE_ROWS / RFX_BLOCK];
            u64x2 r[PART_TILE_ROWS / RFX_BLOCK];
#pragma unroll
            for (int k = 0; k < PART_TILE_ROWS / RFX_BLOCK; k++) {
                const unsigned i = i0 + tid + k * RFX_BLOCK;
                d[k] = L.gd[i < nflush ? i / WC_B : 0];
            }
#pragma unroll
            for (int k = 0; k < PART_TILE_ROWS / RFX_BLOCK; k++) {
                const unsigned j = tid % WC_B;
                r[k] = j < (d[k].w & 0xffu) ? L.carry[d[k].w >> 8][j] : stag[(d[k].z + j) & (PART_TILE_ROWS - 1)];
            }
#pragma unroll
            for (int k = 0; k < PART_TILE_ROWS / RFX_BLOCK; k++) {
                const unsigned i = i0 + tid + k * RFX_BLOCK;
                if (i < nflush) {
                    const u64 at = (((u64)(d[k].y & ~CK_REL) << 32) | d[k].x) + ((d[k].y & CK_REL) ? fresh : 0ULL) + tid % WC_B;
                    A.pool[at] = r[k];
                }
            }
        }
        ck_barrier();
        // the partition's lane keeps what did not fill a group (< WC_B records): reads first, then writes
        {
            const unsigned left = M.pre + M.x - M.pf;
            const unsigned from = M.pf > 0 ? M.excl + (M.pf - M.pre) : M.excl, to = M.pf > 0 ? 0u : M.pre, n = M.pf > 0 ? left : M.x;
            u64x2 keep[WC_B - 1];
#pragma unroll
            for (int j = 0; j < WC_B - 1; j++) keep[j] = stag[(from + j) & (PART_TILE_ROWS - 1)];
#pragma unroll
            for (int j = 0; j < WC_B - 1; j++)
                if ((unsigned)j < n) L.carry[tid][to + j] = keep[j];
        }
        ck_update(M, L, A);
    };
    // The main loop takes full tiles only and ALWAYS issues the next tile's eight loads (a tile past the end is clamped to the last
    // one and never used): with a fixed number of loads behind the ones it waits for, the compiler waits with vmcnt(8); a
    // conditional or ragged load in the loop turns that into vmcnt(0) -- every tile then waits out its successor's loads.
    const i64 nfull = P.nrows / PART_TILE_ROWS;
    const i64 e1 = t1 < nfull ? t1 : nfull; // full tiles of this workgroup: t0, t0 + step, ... < e1
    if (t0 < e1) {
        const i64 last = t0 + ((e1 - 1 - t0) / step) * step;
        ck_load_full<NC>(P, t0, va);
        for (i64 t = t0;;) {
            i64 tn = t + step;
            ck_load_full<NC>(P, tn < e1 ? tn : last, vb);
            tile_body(va, 0xffu, t);
            if (!alive) return;
            t = tn;
            if (t >= e1) break;
            tn = t + step;
            ck_load_full<NC>(P, tn < e1 ? tn : last, va);
            tile_body(vb, 0xffu, t);
            if (!alive) return;
            t = tn;
            if (t >= e1) break;
        }
    }
    if (nfull < ntiles && nfull >= t0 && nfull < t1 && (nfull - t0) % step == 0) { // the ragged last tile is this workgroup's
        const unsigned valid = ck_load_tile<NC>(P, nfull, va);
        tile_body(va, valid, nfull);
        if (!alive) return;
    }
    ck_barrier();
    ck_finish<true>(L, M, A, mn, mx, sel, nulls);
}

#define CK_SEL_WROWS 512 /* rows per wave step: 8 per lane, four 16-byte loads per lane and column */
// ---- selective filters: ONE 1024-lane workgroup per CU (16 waves, as many as K1 keeps resident) with a 4096-record queue ----
// Waves run decoupled: each takes its own 512-row steps (grid-stride over waves), reserves queue space with one LDS atomic and never
// waits for the others -- until a reservation does not fit.  Then that wave raises a flag and parks at the barrier with its survivors
// still in registers; the others see the flag at their next step and join; the queue (the records before the first failed
// reservation) is ranked per partition, chunks are requested, every record is stored at its place in its partition's chunk; all
// resume.  The phases run once per 4096 survivors, not once per tile, and no carry buffers take LDS from the queue.
// Measured on C3w (1e9 rows x 24 B, 10 % selected; the same kernel dropping its survivors: 3.7-3.9 ms): this form 5.3 ms; 256-lane
// workgroups with a 1024..1792-record queue, waves coupled by a barrier per tile or decoupled, write-combined (carry buffers) or not
// 5.5-5.9 ms; per-tile LDS phases (the unfiltered kernel with predicates) 6.8 ms; lock-free placement without a shared queue 6.2 ms
// (single 16-byte record stores are partial-line writes: 2.0 ms per 1e8 records), with per-partition LDS combining buffers
// 6.1-6.6 ms; compacting first and partitioning the compacted rows with the unfiltered kernel 4.9 + 0.9 ms.
#define CK_SELQ 4096
#define CK_SELT 1024
struct CkSelLds {
    unsigned cnt[CK_PARTS];
    uint4 pi[CK_PARTS];
    CkDst di[CK_PARTS];
    unsigned scan_w[4];
    unsigned tile_total, tile_alloc, slab_next, slab_end, dead;
    unsigned q_tail, q_valid, q_flag, alive[2];
    ScopePart red[CK_SELT / RFX_WAVE];
};
template <int NC, int NP>
__global__ __launch_bounds__(CK_SELT) void k_chunk_scatter_sel(const Plan P, const ChunkArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ck_smem[];
    u64x2 *q = (u64x2 *)ck_smem;                                                   // [CK_SELQ]
    unsigned short *q_rank = (unsigned short *)(ck_smem + CK_SELQ * 16);           // [CK_SELQ]
    unsigned char *q_part = ck_smem + CK_SELQ * 18;                                // [CK_SELQ]
    u64x2(*carry)[WC_B] = (u64x2(*)[WC_B])(ck_smem + CK_SELQ * 19 + 64);           // [CK_PARTS][WC_B] records waiting for their 128-byte group
    CkSelLds &L = *(CkSelLds *)(ck_smem + CK_SELQ * 19 + 64 + CK_PARTS * WC_B * 16);
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    CkLane M; // lanes 0..255 own partition tid
    M.pre = M.room = M.nch = M.cur = 0;
    M.cursor = 0;
    M.x = M.pf = M.noff = 0;
    if (tid == 0) {
        L.slab_next = L.slab_end = L.dead = 0;
        L.q_tail = 0;
        L.q_valid = 0xFFFFFFFFu;
        L.q_flag = 0;
        L.alive[0] = L.alive[1] = 0;
    }
    __syncthreads();
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0, nulls = 0;
    const i64 nsteps = (P.nrows + CK_SEL_WROWS - 1) / CK_SEL_WROWS;
    const i64 nwaves = (i64)gridDim.x * (CK_SELT / RFX_WAVE);
    i64 ws = (i64)blockIdx.x * (CK_SELT / RFX_WAVE) + wv; // this wave's next step
    constexpr int VC = NC > 1 ? 1 : 0; // the host put the key in plan column 0 and the value in column 1 (0 when it is the key itself)
    u64 v[NC][8];
    u64 b[8];
    unsigned m = 0, wtot = 0;
    i64 base = 0;
    bool pending = false;
    unsigned round = 0;
    for (;;) {
        // ---- run ahead until a reservation fails, the flag is up, or this wave has nothing left ----
        for (;;) {
            if (!pending) {
                if (ws >= nsteps) break;
                base = ws * CK_SEL_WROWS + lane * 2;
                unsigned valid = 0xffu;
                if ((ws + 1) * CK_SEL_WROWS <= P.nrows) {
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
                    valid = 0;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const i64 row = base + (e >> 1) * 128 + (e & 1);
                        const bool in = row < P.nrows;
                        valid |= (unsigned)in << e;
#pragma unroll
                        for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
                    }
                }
                ws += nwaves;
                m = (NP == 0) ? valid : eval_preds<NC, 8, NP>(S, v, valid);
                wtot = 0;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    b[e] = __ballot((m >> e) & 1u);
                    wtot += (unsigned)__popcll(b[e]);
                }
                if (wtot == 0) continue;
                pending = true;
            }
            if (*(volatile unsigned *)&L.q_flag) break; // a drain has been called: join it, append afterwards
            unsigned wb = 0;
            if (lane == 0) wb = atomicAdd(&L.q_tail, wtot);
            wb = __shfl(wb, 0, 64);
            if (wb + wtot > CK_SELQ) { // does not fit: everything from wb on is invalid, call the drain
                if (lane == 0) {
                    atomicMin(&L.q_valid, wb);
                    *(volatile unsigned *)&L.q_flag = 1u;
                }
                break;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if ((m >> e) & 1u) {
                    const unsigned at = wb + __builtin_amdgcn_mbcnt_hi((unsigned)(b[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b[e], 0u));
                    const i64 k = (i64)v[0][e];
                    const u64 lrow = (u64)(base + (e >> 1) * 128 + (e & 1));
                    u64x2 r;
                    r.x = (lrow << 32) | (u64)(unsigned)(k >> 8);
                    r.y = A.vcol == 0 ? v[0][e] : v[VC][e];
                    q[at] = r;
                    q_part[at] = (unsigned char)(v[0][e] & 255ULL);
                    sel++;
                    if (k == RFX_NULL_I64_D) nulls++;
                    else {
                        mn = k < mn ? k : mn;
                        mx = k > mx ? k : mx;
                    }
                }
                wb += (unsigned)__popcll(b[e]);
            }
            pending = false;
        }
        // ---- drain round (all 16 waves) ----
        if (lane == 0 && (pending || ws < nsteps)) atomicAdd(&L.alive[round & 1], 1u);
        __syncthreads();
        const unsigned vq = L.q_valid, tq = L.q_tail;
        const unsigned n = vq != 0xFFFFFFFFu ? vq : tq;
        const unsigned still = L.alive[round & 1];
        if (n > 0) {
            if (tid < CK_PARTS) L.cnt[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < CK_SELQ / CK_SELT; k++) {
                const unsigned i = tid + k * CK_SELT;
                if (i < n) q_rank[i] = (unsigned short)atomicAdd(&L.cnt[q_part[i]], 1u);
            }
            __syncthreads();
            // one lane per partition (waves 0..3): chunk requests, as ck_plan<1>
            unsigned inc = 0, packed = 0;
            if (tid < CK_PARTS) {
                const unsigned x = L.cnt[tid];
                const unsigned rm = M.room;
                const unsigned pf = ((M.pre + x) / WC_B) * WC_B; // whole 128-byte groups leave, the rest waits in the carry buffer
                const unsigned need = (pf >= rm) ? ((pf - rm) >> A.chs) + 1 : 0;
                M.pf = pf;
                packed = x | (need << 16);
                inc = packed;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    unsigned o = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += o;
                }
                if (lane == 63) L.scan_w[wv] = inc;
            }
            __syncthreads();
            if (tid < CK_PARTS) {
                unsigned wbase = 0;
                for (int w = 0; w < wv; w++) wbase += L.scan_w[w];
                const unsigned excl = wbase + inc - packed;
                M.x = packed & 0xFFFFu;
                M.noff = excl >> 16;
                L.pi[tid] = make_uint4(excl & 0xFFFFu, M.pre, M.pf, M.room);
                CkDst d;
                d.cursor = M.cursor;
                d.noff = M.noff;
                d._pad = 0;
                L.di[tid] = d;
                if (tid == CK_PARTS - 1) {
                    const unsigned tn = (wbase + inc) >> 16;
                    if (tn) {
                        unsigned sn = L.slab_next;
                        if (sn + tn > L.slab_end) {
                            const unsigned slab = (CK_SLAB_BYTES / 16u) >> A.chs;
                            const unsigned g = tn > slab ? tn : slab;
                            sn = atomicAdd(&A.ctl[0], g);
                            L.slab_end = sn + g;
                            if (sn + g > A.max_chunks || sn + g < sn) {
                                L.dead = 1;
                                atomicExch(&A.ctl[1], 1u);
                            }
                        }
                        L.tile_alloc = sn;
                        L.slab_next = sn + tn;
                    }
                }
            }
            __syncthreads();
            if (L.dead) return;
            // old carry -> global for the partitions that flush (positions 0 .. pre-1 of their sequence)
#pragma unroll
            for (int k = 0; k < CK_PARTS * WC_B / CK_SELT; k++) {
                const int idx = tid + k * CK_SELT;
                const int p = idx / WC_B;
                const unsigned j = idx % WC_B;
                const uint4 pi = L.pi[p];
                if (j < pi.y && pi.z > 0) A.pool[ck_dst(L, A, p, j, pi.w)] = carry[p][j];
            }
            __syncthreads();
            // new records: the first (pfl - pre) of a partition complete its 128-byte groups, the rest is the new carry.  (Single 16-byte
            // record stores are partial-line writes: 1.5 ms of this kernel's 5.5 on C3w went to them before the groups were combined.)
#pragma unroll
            for (int k = 0; k < CK_SELQ / CK_SELT; k++) {
                const unsigned i = tid + k * CK_SELT;
                if (i < n) {
                    const unsigned p = q_part[i];
                    const uint4 pi = L.pi[p];
                    const unsigned pos = pi.y + q_rank[i];
                    if (pos < pi.z) A.pool[ck_dst(L, A, p, pos, pi.w)] = q[i];
                    else carry[p][pos - pi.z] = q[i];
                }
            }
            __syncthreads();
            if (tid < CK_PARTS) ck_update(M, L, A);
        }
        if (tid == 0) {
            L.q_tail = 0;
            L.q_valid = 0xFFFFFFFFu;
            L.q_flag = 0;
            L.alive[(round + 1) & 1] = 0;
        }
        __syncthreads();
        round++;
        if (still == 0) break; // nobody has rows or survivors left
    }
    // final record counts of the open chunks, chunk counts, scope
    if (tid < CK_PARTS) {
        const unsigned CH = 1u << A.chs;
        if (M.pre > 0) { // the last, padded group (its chunk has room: room > 0 after every drain; the record count ends before the padding)
            for (unsigned j = 0; j < WC_B; j++) A.pool[M.cursor + j] = carry[tid][j];
        }
        if (M.nch > 0) A.meta[M.cur] = ck_meta(tid, blockIdx.x, CH - M.room + M.pre, M.nch - 1);
        A.wcount[(size_t)blockIdx.x * CK_PARTS + tid] = M.nch;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s), omx = (i64)rfx_shfl_xor_u64((u64)mx, s);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sel += (i64)rfx_shfl_xor_u64((u64)sel, s);
        nulls += (i64)rfx_shfl_xor_u64((u64)nulls, s);
    }
    if (lane == 0) L.red[wv] = ScopePart{mn, mx, sel, nulls};
    __syncthreads();
    if (tid == 0) {
        ScopePart r = L.red[0];
        for (int w = 1; w < CK_SELT / RFX_WAVE; w++) {
            r.mn = L.red[w].mn < r.mn ? L.red[w].mn : r.mn;
            r.mx = L.red[w].mx > r.mx ? L.red[w].mx : r.mx;
            r.sel += L.red[w].sel;
            r.nulls += L.red[w].nulls;
        }
        A.parts[blockIdx.x] = r;
    }
}
// ---- selective filters, keys spread over the partitions: per-partition LDS bins instead of a queue ----
// The same 1024-lane workgroup per CU and the same decoupled waves, but a survivor goes straight into ITS PARTITION's bin (one
// returning LDS atomic for the place, one 16-byte LDS store); when a bin is full the wave keeps the rows that did not fit, raises the
// flag, and all waves meet.  Then every partition's whole 128-byte groups leave -- eight neighbouring lanes store one group from
// eight neighbouring bin entries: a full line, where the queue kernel's records (in arrival order) leave as lone 16-byte stores,
// partial-line writes -- and the < 8 records left move to the front of the bin.  No ranking, no staging, no carry buffers.
// A hot partition fills its bin every few records: the host picks this kernel only when the sample's low key bytes are spread.
#define CK_BIN_CAP 32
struct CkBinLds {
    unsigned tail[CK_PARTS]; // records in the bin (may overshoot CK_BIN_CAP: those lanes keep their rows)
    uint4 pd[CK_PARTS];      // drain: x, y: cursor, z: room | flushing << 16, w: first new chunk (relative)
    unsigned scan_w[4];
    unsigned tile_alloc, slab_next, slab_end, dead;
    unsigned q_flag, alive[2];
    ScopePart red[CK_SELT / RFX_WAVE];
};
#define CK_BIN_LDS (CK_PARTS * CK_BIN_CAP * 16 + sizeof(CkBinLds) + 64)
template <int NC, int NP>
__global__ __launch_bounds__(CK_SELT) void k_chunk_scatter_bin(const Plan P, const ChunkArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ck_smem[];
    u64x2 *bins = (u64x2 *)ck_smem; // [CK_PARTS][CK_BIN_CAP]
    CkBinLds &L = *(CkBinLds *)(ck_smem + CK_PARTS * CK_BIN_CAP * 16);
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    CkLane M; // lanes 0..255 own partition tid; M.pre = records waiting in the bin
    M.pre = M.room = M.nch = M.cur = 0;
    M.cursor = 0;
    M.x = M.pf = M.noff = M.excl = 0;
    if (tid < CK_PARTS) L.tail[tid] = 0;
    if (tid == 0) {
        L.slab_next = L.slab_end = L.dead = 0;
        L.q_flag = 0;
        L.alive[0] = L.alive[1] = 0;
    }
    __syncthreads();
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0, nulls = 0;
    const i64 nsteps = (P.nrows + CK_SEL_WROWS - 1) / CK_SEL_WROWS;
    const i64 nwaves = (i64)gridDim.x * (CK_SELT / RFX_WAVE);
    i64 ws = (i64)blockIdx.x * (CK_SELT / RFX_WAVE) + wv; // this wave's next step
    constexpr int VC = NC > 1 ? 1 : 0; // the host put the key in plan column 0 and the value in column 1 (0 when it is the key itself)
    u64 v[NC][8];
    unsigned m = 0; // rows of the current step still to be placed
    i64 base = 0;
    unsigned round = 0;
    for (;;) {
        // ---- run ahead until a bin is full, the flag is up, or this wave has nothing left ----
        for (;;) {
            if (!__any(m != 0)) {
                if (ws >= nsteps) break;
                base = ws * CK_SEL_WROWS + lane * 2;
                unsigned valid = 0xffu;
                if ((ws + 1) * CK_SEL_WROWS <= P.nrows) {
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
                    valid = 0;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const i64 row = base + (e >> 1) * 128 + (e & 1);
                        const bool in = row < P.nrows;
                        valid |= (unsigned)in << e;
#pragma unroll
                        for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
                    }
                }
                ws += nwaves;
                m = (NP == 0) ? valid : eval_preds<NC, 8, NP>(S, v, valid);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    if ((m >> e) & 1u) {
                        const i64 k = (i64)v[0][e];
                        sel++;
                        if (k == RFX_NULL_I64_D) nulls++;
                        else {
                            mn = k < mn ? k : mn;
                            mx = k > mx ? k : mx;
                        }
                    }
                }
                if (!__any(m != 0)) continue;
            }
            if (*(volatile unsigned *)&L.q_flag) break; // a drain has been called: join it, place afterwards
            bool full = false;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if ((m >> e) & 1u) {
                    const unsigned p = (unsigned)(v[0][e] & 255ULL);
                    const unsigned at = atomicAdd(&L.tail[p], 1u);
                    if (at < CK_BIN_CAP) {
                        const i64 k = (i64)v[0][e];
                        const u64 lrow = (u64)(base + (e >> 1) * 128 + (e & 1));
                        u64x2 r;
                        r.x = (lrow << 32) | (u64)(unsigned)(k >> 8);
                        r.y = A.vcol == 0 ? v[0][e] : v[VC][e];
                        bins[p * CK_BIN_CAP + at] = r;
                        m &= ~(1u << e);
                    } else full = true;
                }
            }
            if (__any(full)) {
                if (lane == 0) *(volatile unsigned *)&L.q_flag = 1u;
                break;
            }
        }
        // ---- drain round (all 16 waves) ----
        const bool waiting = __any(m != 0); // all lanes vote: lane 0 alone may have placed its rows
        if (lane == 0 && (waiting || ws < nsteps)) atomicAdd(&L.alive[round & 1], 1u);
        __syncthreads();
        const unsigned still = L.alive[round & 1];
        unsigned have = 0, inc = 0, need = 0;
        if (tid < CK_PARTS) { // one lane per partition (waves 0..3): what leaves, chunk requests
            const unsigned t = L.tail[tid];
            have = t < CK_BIN_CAP ? t : CK_BIN_CAP;
            const unsigned pf = (have / WC_B) * WC_B;
            need = (pf >= M.room) ? 1u : 0u; // a drain moves at most CK_BIN_CAP records of a partition: never more than one new chunk
            M.pf = pf;
            inc = need;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                unsigned o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            if (lane == 63) L.scan_w[wv] = inc;
        }
        __syncthreads();
        if (tid < CK_PARTS) {
            unsigned wbase = 0;
            for (int w = 0; w < wv; w++) wbase += L.scan_w[w];
            M.noff = wbase + inc - need;
            L.pd[tid] = make_uint4((unsigned)M.cursor, (unsigned)(M.cursor >> 32), M.room | (M.pf << 16), M.noff);
            if (tid == CK_PARTS - 1) ck_request(L, A, wbase + inc);
        }
        __syncthreads();
        if (L.dead) return;
        {
            const u64 fresh = (u64)L.tile_alloc << A.chs;
#pragma unroll
            for (int k = 0; k < CK_PARTS * CK_BIN_CAP / CK_SELT; k++) {
                const unsigned idx = tid + k * CK_SELT;
                const unsigned p = idx / CK_BIN_CAP, pos = idx % CK_BIN_CAP;
                const uint4 d = L.pd[p];
                const unsigned rm = d.z & 0xffffu;
                if (pos < (d.z >> 16)) {
                    const u64 at = pos < rm ? (((u64)d.y << 32) | d.x) + pos : fresh + ((u64)d.w << A.chs) + (pos - rm);
                    A.pool[at] = bins[idx];
                }
            }
        }
        __syncthreads();
        if (tid < CK_PARTS) {
            const unsigned left = have - M.pf; // < WC_B
            if (M.pf > 0) {
                u64x2 keep[WC_B - 1];
#pragma unroll
                for (int j = 0; j < WC_B - 1; j++) keep[j] = bins[tid * CK_BIN_CAP + ((M.pf + j) & (CK_BIN_CAP - 1))];
#pragma unroll
                for (int j = 0; j < WC_B - 1; j++)
                    if ((unsigned)j < left) bins[tid * CK_BIN_CAP + j] = keep[j];
            }
            L.tail[tid] = left;
            M.x = have - M.pre; // ck_update: pre <- pre + x - pf = left
            ck_update(M, L, A);
        }
        if (tid == 0) {
            L.q_flag = 0;
            L.alive[(round + 1) & 1] = 0;
        }
        __syncthreads();
        round++;
        if (still == 0) break; // nobody has rows left
    }
    // the last, padded group of every partition, final record counts of the open chunks, chunk counts, scope
    if (tid < CK_PARTS) {
        const unsigned CH = 1u << A.chs;
        if (M.pre > 0) { // its chunk has room (room > 0 after every drain); the record count ends before the padding
            for (unsigned j = 0; j < WC_B; j++) A.pool[M.cursor + j] = bins[tid * CK_BIN_CAP + j];
        }
        if (M.nch > 0) A.meta[M.cur] = ck_meta(tid, blockIdx.x, CH - M.room + M.pre, M.nch - 1);
        A.wcount[(size_t)blockIdx.x * CK_PARTS + tid] = M.nch;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s), omx = (i64)rfx_shfl_xor_u64((u64)mx, s);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sel += (i64)rfx_shfl_xor_u64((u64)sel, s);
        nulls += (i64)rfx_shfl_xor_u64((u64)nulls, s);
    }
    if (lane == 0) L.red[wv] = ScopePart{mn, mx, sel, nulls};
    __syncthreads();
    if (tid == 0) {
        ScopePart r = L.red[0];
        for (int w = 1; w < CK_SELT / RFX_WAVE; w++) {
            r.mn = L.red[w].mn < r.mn ? L.red[w].mn : r.mn;
            r.mx = L.red[w].mx > r.mx ? L.red[w].mx : r.mx;
            r.sel += L.red[w].sel;
            r.nulls += L.red[w].nulls;
        }
        A.parts[blockIdx.x] = r;
    }
}

#define CK_SEL_LDS (CK_SELQ * 19 + 64 + CK_PARTS * WC_B * 16 + sizeof(CkSelLds) + 64)

// wcount[w][p] -> exclusive offset of (w, p) inside partition p's chunk list; part_start[p] = first entry of partition p.
__global__ __launch_bounds__(1024) void k_chunk_offsets(const ChunkArgs A) {
    __shared__ unsigned slice_sum[4][CK_PARTS];
    __shared__ u64 tot[CK_PARTS];
    const int p = threadIdx.x & 255, sl = threadIdx.x >> 8;
    const int per = (A.nwg + 3) / 4;
    const int w0 = sl * per, w1 = (w0 + per < A.nwg) ? w0 + per : A.nwg;
    unsigned sum = 0;
    for (int w = w0; w < w1; w++) sum += A.wcount[(size_t)w * CK_PARTS + p];
    slice_sum[sl][p] = sum;
    __syncthreads();
    unsigned run = 0, total = 0;
    for (int s2 = 0; s2 < 4; s2++) {
        const unsigned v = slice_sum[s2][p];
        if (s2 < sl) run += v;
        total += v;
    }
    for (int w = w0; w < w1; w++) {
        const size_t i = (size_t)w * CK_PARTS + p;
        const unsigned c = A.wcount[i];
        A.wcount[i] = run;
        run += c;
    }
    if (sl == 0) tot[p] = total;
    __syncthreads();
    for (int s = 1; s < CK_PARTS; s <<= 1) {
        u64 add = 0;
        if (sl == 0 && p >= s) add = tot[p - s];
        __syncthreads();
        if (sl == 0) tot[p] += add;
        __syncthreads();
    }
    if (sl == 0) {
        A.part_start[p + 1] = tot[p];
        if (p == 0) A.part_start[0] = 0;
    }
}

__global__ __launch_bounds__(RFX_BLOCK) void k_chunk_place(const ChunkArgs A) {
    const unsigned n = A.ctl[0] < A.max_chunks ? A.ctl[0] : A.max_chunks;
    for (unsigned c = blockIdx.x * RFX_BLOCK + threadIdx.x; c < n; c += gridDim.x * RFX_BLOCK) {
        const u64 m = A.meta[c];
        if (m == CK_FREE) continue;
        const unsigned p = (unsigned)(m & 255ULL), w = (unsigned)((m >> 8) & 0xFFFULL), nrec = (unsigned)((m >> 20) & 0xFFFFFULL), ord = (unsigned)(m >> 40);
        A.plist[A.part_start[p] + A.wcount[(size_t)w * CK_PARTS + p] + ord] = (u64)c | ((u64)nrec << 32);
    }
}

// ---- per-partition LDS aggregation over a chunk list (one value plane) ----
struct ChunkAggArgs {
    i64 kmin, range;
    i64 local;  // table cells per partition: slots congruent to one residue mod 256
    int split;  // workgroups per partition
    int chs;    // log2(records per chunk)
    const u64x2 *pool;
    const u64 *part_start;
    const u64 *plist;
    u64 *first;
    u64 *acc[RFX_MAX_AGGS];
    u64 *cnt[RFX_MAX_AGGS];
};
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_chunk_aggregate(const Plan P, const ChunkAggArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x / A.split, s = blockIdx.x % A.split;
    const i64 local = A.local;
    int kind[RFX_MAX_AGGS], f64[RFX_MAX_AGGS], arr_of[RFX_MAX_AGGS], skip[RFX_MAX_AGGS], hasv[RFX_MAX_AGGS];
    {
        int arr = 1;
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            kind[a] = (a < P.nagg) ? P.aggs[a].kind : -1;
            f64[a] = (a < P.nagg) ? P.aggs[a].f64 : 0;
            skip[a] = (a < P.nagg) ? P.aggs[a].skipnull : 0;
            hasv[a] = (a < P.nagg) ? (P.aggs[a].col >= 0) : 0;
            arr_of[a] = arr;
            if (kind[a] >= 0) arr += agg_has_cnt(kind[a], f64[a]) ? 2 : 1;
        }
    }
    for (i64 i = tid; i < local; i += THREADS) smem[i] = (u64)RFX_INF_I64_D;
#pragma unroll
    for (int a = 0; a < RFX_MAX_AGGS; a++) {
        if (kind[a] < 0) continue;
        const u64 id = acc_identity(kind[a], f64[a]);
        for (i64 i = tid; i < local; i += THREADS) smem[(i64)arr_of[a] * local + i] = id;
        if (agg_has_cnt(kind[a], f64[a])) {
            for (i64 i = tid; i < local; i += THREADS) smem[(i64)(arr_of[a] + 1) * local + i] = 0;
        }
    }
    __syncthreads();
    const u64 beg = A.part_start[p], end = A.part_start[p + 1];
    const u64 len = end - beg;
    const u64 per = (len + A.split - 1) / A.split;
    const u64 b0 = beg + per * s;
    const u64 b1 = (b0 + per < end) ? (b0 + per) : end;
    const u64 row0 = (u64)P.row0;
    const unsigned shift = (unsigned)(((i64)p - A.kmin) >> 8); // slot = (key - kmin) >> 8 = kh + floor((p - kmin) / 256), key = kh * 256 + p
    constexpr int RU = 4; // records in flight per lane (8: the same 3.3 ms per 1e9 records -- the LDS atomics bound this pass)
    const unsigned chs = (unsigned)A.chs;
    auto apply = [&](const u64x2 &rec) {
        const u64 slot = (u64)(unsigned)((unsigned)rec.x + shift);
        if (slot >= (u64)local) return; // a key outside the scope the tables were sized for: not ours
        const u64 row = row0 + (rec.x >> 32);
        if (row < smem[slot]) atomicMin((unsigned long long *)&smem[slot], (unsigned long long)row);
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            if (kind[a] < 0) continue;
            group_apply(&smem[(i64)arr_of[a] * local + slot], &smem[(i64)(arr_of[a] + 1) * local + slot], kind[a], f64[a], hasv[a] ? rec.y : 0ULL, skip[a]);
        }
    };
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    if ((1u << chs) >= (unsigned)THREADS) {
        // a chunk is at least one record per lane: walk the list chunk by chunk (the entry is wave-uniform: a scalar load)
        for (u64 ci = b0; ci < b1; ci++) {
            const u64 e = A.plist[ci];
            const unsigned nrec = (unsigned)(e >> 32);
            const u64x2 *src = A.pool + ((e & 0xFFFFFFFFULL) << chs);
            for (unsigned r0 = 0; r0 < nrec; r0 += THREADS * RU) {
                u64x2 q[RU];
                bool in[RU];
#pragma unroll
                for (int r = 0; r < RU; r++) {
                    const unsigned rec = r0 + (unsigned)r * THREADS + tid;
                    in[r] = rec < nrec;
                    q[r].x = q[r].y = 0;
                    if (in[r]) {
                        const v2 t = __builtin_nontemporal_load((const v2 *)(src + rec));
                        q[r].x = t.x;
                        q[r].y = t.y;
                    }
                }
#pragma unroll
                for (int r = 0; r < RU; r++)
                    if (in[r]) apply(q[r]);
            }
        }
    } else {
        const u64 vend = (b1 - b0) << chs; // the chunk list as one virtual record range
        // chunk-list entries are fetched one step ahead: the record loads of a step never wait on a dependent load
        u64 ent[RU];
#pragma unroll
        for (int r = 0; r < RU; r++) {
            const u64 vi = (u64)r * THREADS + tid;
            ent[r] = (vi < vend) ? A.plist[b0 + (vi >> chs)] : 0ULL;
        }
        for (u64 v0 = 0; v0 < vend; v0 += (u64)THREADS * RU) {
            u64x2 q[RU];
            bool in[RU];
#pragma unroll
            for (int r = 0; r < RU; r++) {
                const u64 vi = v0 + (u64)r * THREADS + tid;
                const unsigned rec = (unsigned)vi & ((1u << chs) - 1u);
                in[r] = vi < vend && rec < (unsigned)(ent[r] >> 32);
                q[r].x = 0;
                q[r].y = 0;
                if (in[r]) {
                    const v2 t = __builtin_nontemporal_load((const v2 *)(A.pool + ((ent[r] & 0xFFFFFFFFULL) << chs) + rec));
                    q[r].x = t.x;
                    q[r].y = t.y;
                }
            }
#pragma unroll
            for (int r = 0; r < RU; r++) {
                const u64 vi = v0 + (u64)THREADS * RU + (u64)r * THREADS + tid;
                ent[r] = (vi < vend) ? A.plist[b0 + (vi >> chs)] : 0ULL;
            }
#pragma unroll
            for (int r = 0; r < RU; r++)
                if (in[r]) apply(q[r]);
        }
    }
    __syncthreads();
    for (i64 i = tid; i < local; i += THREADS) {
        const u64 f = smem[i];
        if (f == (u64)RFX_INF_I64_D) continue;
        const i64 g = (i << 8) | (i64)(((u64)p - (u64)A.kmin) & 255ULL);
        if (g >= A.range) continue;
        if (f < A.first[g]) atomicMin((unsigned long long *)&A.first[g], (unsigned long long)f);
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            if (kind[a] < 0) continue;
            const bool hc = agg_has_cnt(kind[a], f64[a]);
            group_merge_cell(&A.acc[a][g], hc ? &A.cnt[a][g] : (u64 *)0, kind[a], f64[a], smem[(i64)arr_of[a] * local + i],
                             hc ? smem[(i64)(arr_of[a] + 1) * local + i] : 0ULL);
        }
    }
}

// ---- strided sample: a first guess of the key range and of the filter's selectivity, only to choose the pass (never a result) ----
template <int NC>
__global__ __launch_bounds__(RFX_BLOCK) void k_scope_sample(const Plan P, int key_idx, i64 stride, i64 nsamp, i64 *__restrict__ out, unsigned *__restrict__ hist) {
    __shared__ i64 red[3][RFX_BLOCK / RFX_WAVE];
    PredSet<RFX_MAX_PREDS> S;
    predset_load<RFX_MAX_PREDS>(P, S);
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0;
    for (i64 i = (i64)blockIdx.x * RFX_BLOCK + threadIdx.x; i < nsamp; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 r = i * stride;
        if (r >= P.nrows) break;
        u64 v[NC][1];
#pragma unroll
        for (int c = 0; c < NC; c++) v[c][0] = P.cols[c][r];
        u64 key[1];
        sel_col<NC, 1>(key, v, key_idx);
        const i64 k = (i64)key[0];
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
        const unsigned hit = (P.npred == 0) ? 1u : (eval_preds<NC, 1, RFX_MAX_PREDS>(S, v, 1u) & 1u);
        sel += hit;
        if (hit) atomicAdd(&hist[(unsigned)((u64)k & 255ULL)], 1u); // how evenly the selection spreads over the 256 partitions
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s), omx = (i64)rfx_shfl_xor_u64((u64)mx, s);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sel += (i64)rfx_shfl_xor_u64((u64)sel, s);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = mn;
        red[1][threadIdx.x >> 6] = mx;
        red[2][threadIdx.x >> 6] = sel;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) {
            mn = red[0][w] < mn ? red[0][w] : mn;
            mx = red[1][w] > mx ? red[1][w] : mx;
            sel += red[2][w];
        }
        out[4 * blockIdx.x] = mn;
        out[4 * blockIdx.x + 1] = mx;
        out[4 * blockIdx.x + 2] = sel;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int rfx_chunk_reserve(rfx_ctx *c, size_t bytes);
// the plane form of this pass (rfx_group_plane.hip): 12-byte records in two planes, tried first where it applies
int rfx_plane_scope(rfx_ctx *c, const Plan &P, int key_idx, const void *d_key, int npred, int logic, unsigned long long est_range, double frac,
                    i64 *kmin, i64 *kmax, i64 *seen);
int rfx_plane_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t);
void rfx_plane_invalidate(rfx_ctx *c);

static void chunk_pred_sig(const Plan &P, u64 (*sig)[6]) {
    for (int i = 0; i < P.npred; i++) {
        const PlanPred &q = P.preds[i];
        sig[i][0] = (u64)(uintptr_t)P.cols[q.col];
        sig[i][1] = q.rhs_col >= 0 ? (u64)(uintptr_t)P.cols[q.rhs_col] : 0;
        sig[i][2] = (u64)q.op;
        sig[i][3] = (u64)(q.dom_f64 | (q.lhs_cvt << 1) | (q.rhs_cvt << 2) | (q.more << 3) | ((u64)q.tree << 4));
        sig[i][4] = q.rhs_bits;
        sig[i][5] = 0;
    }
}

static int narr_of(const Plan &P) {
    int narr = 1;
    for (int a = 0; a < P.nagg; a++) narr += 1 + (agg_has_cnt(P.aggs[a].kind, P.aggs[a].f64) ? 1 : 0);
    return narr;
}

// The single value plane of a plan: every aggregate reads the same column (or none).  -1: no plane / several.
static int single_value_col(const Plan &P) {
    int vc = -1;
    for (int a = 0; a < P.nagg; a++) {
        const PlanAgg &ag = P.aggs[a];
        if (ag.kind == RFX_AGG_COUNT || ag.kind == RFX_AGG_FIRST || ag.col < 0) continue;
        if (ag.col >= RFX_XCOL) return -1;
        if (vc >= 0 && vc != ag.col) return -1;
        vc = ag.col;
    }
    return vc;
}

template <int NC>
static void launch_chunk_scatter(rfx_ctx *c, const Plan &P, const ChunkArgs &A) {
    if (P.npred == 0) hipLaunchKernelGGL((k_chunk_scatter<NC, 0>), dim3(A.nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
    else if (P.npred <= 2) hipLaunchKernelGGL((k_chunk_scatter<NC, 2>), dim3(A.nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
    else hipLaunchKernelGGL((k_chunk_scatter<NC, RFX_MAX_PREDS>), dim3(A.nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
}
template <int NC, int NP>
static int launch_chunk_scatter_sel_np(rfx_ctx *c, const Plan &P, const ChunkArgs &A) {
    static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */ // per instantiation
    if (!((attr_set >> (c->device & 63)) & 1ull)) {
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chunk_scatter_sel<NC, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CK_SEL_LDS));
        __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);
    }
    hipLaunchKernelGGL((k_chunk_scatter_sel<NC, NP>), dim3(A.nwg), dim3(CK_SELT), CK_SEL_LDS, c->stream, P, A);
    return RFX_OK;
}
template <int NC, int NP>
static int launch_chunk_scatter_bin_np(rfx_ctx *c, const Plan &P, const ChunkArgs &A) {
    static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */ // per instantiation
    if (!((attr_set >> (c->device & 63)) & 1ull)) {
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chunk_scatter_bin<NC, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CK_BIN_LDS));
        __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);
    }
    hipLaunchKernelGGL((k_chunk_scatter_bin<NC, NP>), dim3(A.nwg), dim3(CK_SELT), CK_BIN_LDS, c->stream, P, A);
    return RFX_OK;
}
template <int NC>
static int launch_chunk_scatter_bin(rfx_ctx *c, const Plan &P, const ChunkArgs &A) {
    if (P.npred == 0) return launch_chunk_scatter_bin_np<NC, 0>(c, P, A);
    if (P.npred == 1) return launch_chunk_scatter_bin_np<NC, 1>(c, P, A);
    if (P.npred <= 3) return launch_chunk_scatter_bin_np<NC, 3>(c, P, A);
    return launch_chunk_scatter_bin_np<NC, RFX_MAX_PREDS>(c, P, A);
}
template <int NC>
static int launch_chunk_scatter_sel(rfx_ctx *c, const Plan &P, const ChunkArgs &A) {
    if (P.npred == 1) return launch_chunk_scatter_sel_np<NC, 1>(c, P, A);
    if (P.npred <= 3) return launch_chunk_scatter_sel_np<NC, 3>(c, P, A);
    return launch_chunk_scatter_sel_np<NC, RFX_MAX_PREDS>(c, P, A);
}
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// the scratch block: [ctl 256 B][scope partials][chunk counts][part_start][metas][chunk lists][record pool]
static void chunk_layout(rfx_ctx *c, int nwg, size_t max_chunks, ChunkArgs *A, size_t *total) {
    const size_t o_parts = 256, o_wcount = o_parts + al256((size_t)nwg * sizeof(ScopePart));
    const size_t o_start = o_wcount + al256((size_t)nwg * CK_PARTS * 4), o_meta = o_start + al256((CK_PARTS + 2) * 8);
    const size_t o_plist = o_meta + al256(max_chunks * 8), o_pool = o_plist + al256(max_chunks * 8);
    if (total) *total = o_pool + ((max_chunks << A->chs) * 16);
    char *w = (char *)c->d_chunk;
    A->nwg = nwg;
    A->ctl = (unsigned *)w;
    A->parts = (ScopePart *)(w + o_parts);
    A->wcount = (unsigned *)(w + o_wcount);
    A->part_start = (u64 *)(w + o_start);
    A->meta = (u64 *)(w + o_meta);
    A->plist = (u64 *)(w + o_plist);
    A->pool = (u64x2 *)(w + o_pool);
    A->max_chunks = (unsigned)max_chunks;
}

// Geometry of one tile-sorted scatter over `nrows` input rows of which about `est_rows` are selected; reserves the scratch block.
static int chunk_geometry(rfx_ctx *c, i64 nrows, i64 est_rows, ChunkArgs *A, i64 *tpw_out) {
    const i64 ntiles = (nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    int nwg = c->num_cus * 2; // what the kernel's LDS lets a CU hold
    if ((i64)nwg * 16 > ntiles) nwg = (int)((ntiles + 15) / 16);
    const i64 tpw = (ntiles + nwg - 1) / nwg;
    nwg = (int)((ntiles + tpw - 1) / tpw);
    // records per chunk: about a third of what one (workgroup, partition) will see, 256 .. 2048 (4 .. 32 KB)
    A->chs = 9; // >= 512 records: pass 2 then walks the list chunk by chunk with wave-uniform entries
    while (A->chs < 11 && (est_rows / ((i64)nwg * CK_PARTS)) / 3 >= (2LL << A->chs)) A->chs++;
    const size_t CH = (size_t)1 << A->chs;
    const size_t slab = (CK_SLAB_BYTES / 16) / CH;
    const size_t data_chunks = (size_t)((est_rows + (i64)CH - 1) / (i64)CH);
    const size_t max_chunks = data_chunks + data_chunks / 4 + (size_t)nwg * (CK_PARTS + 2 * slab) + 1024;
    if (max_chunks >= (1ULL << 32)) return RFX_ESTATE;
    size_t need = 0;
    chunk_layout(c, nwg, max_chunks, A, &need);
    if (rfx_chunk_reserve(c, need) != RFX_OK) return RFX_ESTATE; // no room for the pool: the column passes need less
    chunk_layout(c, nwg, max_chunks, A, NULL);
    A->tiles_per_wg = (c->flags & RFX_TUNE_CHUNK_CONTIG) ? tpw : 0;
    *tpw_out = tpw;
    return RFX_OK;
}

static void fold_scope(const ScopePart *h, int n, i64 *kmin, i64 *kmax, i64 *seen, i64 *nulls_out) {
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0, nulls = 0;
    for (int i = 0; i < n; i++) {
        mn = h[i].mn < mn ? h[i].mn : mn;
        mx = h[i].mx > mx ? h[i].mx : mx;
        sel += h[i].sel;
        nulls += h[i].nulls;
    }
    *seen = sel;
    if (nulls > 0) { // a null key is the value INT64_MIN for index_scope_i64
        mn = RFX_NULL_I64_D;
        if (nulls == sel) mx = RFX_NULL_I64_D;
    }
    *kmin = mn;
    *kmax = mx;
    *nulls_out = nulls;
}

// Scope pass that also partitions.  RFX_ESTATE: not applicable, the caller runs the plain scope pass.
int rfx_chunk_scope(rfx_ctx *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg, i64 nrows,
                    i64 *kmin, i64 *kmax, i64 *seen) {
    c->ck_valid = 0;
    rfx_plane_invalidate(c);
    if ((c->flags & (RFX_TUNE_NO_PARTITION | RFX_TUNE_NO_CHUNK)) || nrows >= (1LL << 32) || nrows < ((c->flags & RFX_TUNE_CHUNK_SMALL) ? (1LL << 16) : (1LL << 22)) || nagg < 1) return RFX_ESTATE;
    Plan P;
    int key_idx = 0;
    int rc = rfx_plan_build(&P, preds, npred, logic, aggs, nagg, d_key, &key_idx, nrows, 0);
    if (rc != RFX_OK) return RFX_ESTATE; // the accumulate call reports it
    if (P.nx > 0 || P.ncols > 6) return RFX_ESTATE;
    const int vc = single_value_col(P); // (< 0: several value columns -- only the plane kernels carry those)
    const int narr = narr_of(P);
    // a strided sample guesses the key range and the selectivity: only ranges whose tables overflow one workgroup's LDS but fit
    // 256 partitions come here (the LDS-direct kernel is one pass already; wider ranges need more partitions than the low 8 bits give)
    const i64 nsamp = 1 << 14;
    const int sgrid = 16;
    // The sample of the SAME query over the SAME columns is remembered (four of them per context: the kernel, its memset, the copy back and
    // the wait are 30 us of a 5 ms query).  Whatever comes out of it only sizes and routes -- ranges, capacities, which kernel -- and every
    // form downstream reports what did not fit, so a remembered sample that no longer describes the data (a column changed in place behind
    // the context's back) costs a fallback, never an answer; uploads through the context drop it (rfx_hip_h2d).
    u64 sig = 0xCBF29CE484222325ULL;
    {
        auto mix = [&](u64 v) { sig = (sig ^ v) * 0x100000001B3ULL; sig ^= sig >> 29; };
        mix((u64)(uintptr_t)d_key); mix((u64)nrows); mix((u64)P.npred); mix((u64)P.logic); mix((u64)P.ncols); mix((u64)key_idx);
        for (int i = 0; i < P.ncols; i++) mix((u64)(uintptr_t)P.cols[i]);
        for (int i = 0; i < P.npred; i++) {
            const PlanPred &q = P.preds[i];
            mix((u64)q.col | ((u64)(unsigned)q.rhs_col << 8) | ((u64)q.op << 16) | ((u64)q.dom_f64 << 20) | ((u64)q.lhs_cvt << 21) | ((u64)q.rhs_cvt << 22) | ((u64)q.more << 23) | ((u64)(unsigned)q.tree << 24));
            mix(q.rhs_bits);
        }
        if (sig == 0) sig = 1;
    }
    struct SampleMemo { u64 sig; i64 smn, smx, ssel; unsigned hot, age, uses, pad; }; // 4 x 48 bytes <= the 256 rfx_ctx.hip clears
    static_assert(4 * sizeof(SampleMemo) <= 256, "memo block");
    if (!c->ext_p[5]) c->ext_p[5] = calloc(1, 256);
    SampleMemo *memo = (SampleMemo *)c->ext_p[5];
    i64 smn = RFX_INF_I64_D, smx = RFX_NULL_I64_D, ssel = 0;
    unsigned hot = 0;
    bool remembered = false;
    int hit = -1;
    static unsigned memo_clock = 0; // (shared by the contexts of a process: it only orders the entries of each)
    for (int i = 0; memo && i < 4 && !getenv("RFX_NO_SAMPLE_MEMO"); i++)
        if (memo[i].sig == sig) {
            // (a remembered sample serves 32 queries, then the data is sampled again: what a stale one costs -- a slower route, e.g. the hashed
            //  tables for keys it calls sparse -- is bounded, and the 30 us come back once in 33 queries)
            if (++memo[i].uses > 32u) { memo[i].sig = 0; break; }
            smn = memo[i].smn, smx = memo[i].smx, ssel = memo[i].ssel, hot = memo[i].hot;
            memo[i].age = __atomic_add_fetch(&memo_clock, 1u, __ATOMIC_RELAXED);
            remembered = true;
            hit = i;
        }
    if (!remembered) {
    rc = rfx_ws_reserve(c, (size_t)sgrid * 32 + CK_PARTS * 4);
    if (rc != RFX_OK) return rc;
    unsigned *d_hist = (unsigned *)((char *)c->d_ws + (size_t)sgrid * 32);
    RFX_HIP_CHECK(hipMemsetAsync(d_hist, 0, CK_PARTS * 4, c->stream));
    switch (P.ncols) {
        case 1: hipLaunchKernelGGL(k_scope_sample<1>, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, nrows / nsamp, nsamp, (i64 *)c->d_ws, d_hist); break;
        case 2: hipLaunchKernelGGL(k_scope_sample<2>, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, nrows / nsamp, nsamp, (i64 *)c->d_ws, d_hist); break;
        case 3: hipLaunchKernelGGL(k_scope_sample<3>, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, nrows / nsamp, nsamp, (i64 *)c->d_ws, d_hist); break;
        case 4: hipLaunchKernelGGL(k_scope_sample<4>, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, nrows / nsamp, nsamp, (i64 *)c->d_ws, d_hist); break;
        case 5: hipLaunchKernelGGL(k_scope_sample<5>, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, nrows / nsamp, nsamp, (i64 *)c->d_ws, d_hist); break;
        default: hipLaunchKernelGGL(k_scope_sample<6>, dim3(sgrid), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, nrows / nsamp, nsamp, (i64 *)c->d_ws, d_hist); break;
    }
    RFX_HIP_CHECK(hipGetLastError());
    i64 *hs = (i64 *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(hs, c->d_ws, (size_t)sgrid * 32 + CK_PARTS * 4, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    const unsigned *hh = (const unsigned *)((const char *)hs + (size_t)sgrid * 32);
    for (int i = 0; i < CK_PARTS; i++) hot = hh[i] > hot ? hh[i] : hot;
    for (int i = 0; i < sgrid; i++) {
        smn = hs[4 * i] < smn ? hs[4 * i] : smn;
        smx = hs[4 * i + 1] > smx ? hs[4 * i + 1] : smx;
        ssel += hs[4 * i + 2];
    }
    if (memo) { // replaces the entry used longest ago
        int at = 0;
        for (int i = 1; i < 4; i++)
            if (memo[i].age < memo[at].age) at = i;
        memo[at] = SampleMemo{sig, smn, smx, ssel, hot, __atomic_add_fetch(&memo_clock, 1u, __ATOMIC_RELAXED), 0u, 0u};
    }
    }
    if (smn == RFX_NULL_I64_D || smx < smn || ssel == 0) return RFX_ESTATE;
    const unsigned long long est = (unsigned long long)smx - (unsigned long long)smn + 1ULL;
    if (est == 0 || est > (1ULL << 40)) return RFX_ESTATE;
    if ((size_t)est * 12 <= (size_t)150 * 1024) return RFX_ESTATE;                                    // LDS-direct territory
    const double frac = (double)ssel / (double)nsamp;
    i64 est_rows = (i64)((double)nrows * (frac * 1.5 + 0.02));
    if (est_rows > nrows || npred == 0) est_rows = nrows;
    if (est > (unsigned long long)est_rows * 2) return RFX_ESTATE;                                    // sparse keys: the hashed path
    // bins fill evenly when no partition takes much more than its 1/256 of the selection (sampled: the fullest bin against the mean)
    const bool spread = (double)hot <= 6.0 * ((double)ssel / CK_PARTS) + 24.0;
    // The bins kernel also takes UNFILTERED inputs with spread keys (NP = 0): C3 10.33 against 10.50 ms for the tile-sorted kernel --
    // two unlike designs within 2 % of each other, both at what a device copy of the records costs; the tile-sorted kernel keeps the
    // skewed and the mildly filtered inputs (RFX_TUNE_CHUNK_QUEUE: everywhere).
    const bool selective = (npred > 0 && frac <= 0.4) || (c->flags & RFX_TUNE_CHUNK_BINS) || (npred == 0 && spread && !(c->flags & RFX_TUNE_CHUNK_QUEUE));
    const bool bins = selective && (spread || (c->flags & RFX_TUNE_CHUNK_BINS)) && !(c->flags & RFX_TUNE_CHUNK_QUEUE);
    if (spread && !(c->flags & (RFX_TUNE_CHUNK_QUEUE | RFX_TUNE_CHUNK_BINS | RFX_TUNE_CHUNK_CONTIG))) {
        // spread keys: two planes of 8 + 4 bytes per record through decoupled LDS rings (40 instead of 48 bytes moved per row on C3)
        const int prc = rfx_plane_scope(c, P, key_idx, d_key, npred, logic, est, frac, kmin, kmax, seen);
        if (prc != RFX_ESTATE) return prc;
        if (remembered && hit >= 0) memo[hit].sig = 0; // the planes gave up on what a REMEMBERED sample promised (a region overflowed, ...): sample afresh next time
    }
    if (vc < 0 || P.ncols > 4) return RFX_ESTATE;                                                      // the chunk records carry one value
    if ((size_t)((est + 255) >> 8) * narr * 8 > (size_t)PART_LDS_BIG_BYTES) return RFX_ESTATE;         // needs more than 256 partitions
    ChunkArgs A;
    memset(&A, 0, sizeof(A));
    i64 tpw = 0, nulls = 0;
    Plan Pc = P; // key -> column 0, value -> column 1 (the selective kernels read them without a run-time select)
    if (selective) {
        int perm[RFX_MAX_COLS], n2 = 0; // perm[new] = old
        perm[n2++] = key_idx;
        if (vc != key_idx) perm[n2++] = vc;
        for (int i = 0; i < P.ncols; i++)
            if (i != key_idx && i != vc) perm[n2++] = i;
        int inv[RFX_MAX_COLS];
        for (int i = 0; i < n2; i++) {
            Pc.cols[i] = P.cols[perm[i]];
            inv[perm[i]] = i;
        }
        for (int i = 0; i < P.npred; i++) {
            Pc.preds[i].col = inv[P.preds[i].col];
            if (P.preds[i].rhs_col >= 0) Pc.preds[i].rhs_col = inv[P.preds[i].rhs_col];
        }
    }
    if (selective) {
        // ---- one kernel: filter, queue, place (1024-lane workgroups, one per CU) ----
        const int nwg = (int)((nrows / CK_SEL_WROWS / (CK_SELT / RFX_WAVE)) < c->num_cus ? (nrows / CK_SEL_WROWS / (CK_SELT / RFX_WAVE)) + 1 : c->num_cus);
        A.chs = 10; // few, large chunks: 65 K (workgroup, partition) streams share the selection
        while (A.chs < 11 && (est_rows / ((i64)nwg * CK_PARTS)) / 3 >= (2LL << A.chs)) A.chs++;
        const size_t CH = (size_t)1 << A.chs;
        const size_t slab = (CK_SLAB_BYTES / 16) / CH;
        const size_t data_chunks = (size_t)((est_rows + (i64)CH - 1) / (i64)CH);
        const size_t max_chunks = data_chunks + data_chunks / 4 + (size_t)nwg * (CK_PARTS + 2 * slab) + 1024;
        if (max_chunks >= (1ULL << 32)) return RFX_ESTATE;
        size_t need = 0;
        chunk_layout(c, nwg, max_chunks, &A, &need);
        if (rfx_chunk_reserve(c, need) != RFX_OK) return RFX_ESTATE;
        chunk_layout(c, nwg, max_chunks, &A, NULL);
        A.key_idx = 0;
        A.vcol = vc == key_idx ? 0 : 1;
        RFX_HIP_CHECK(hipMemsetAsync(A.ctl, 0, 256, c->stream));
        RFX_HIP_CHECK(hipMemsetAsync(A.meta, 0xFF, (size_t)A.max_chunks * 8, c->stream));
        c->ext_i[3 + RFX_STAT_CHUNK_SCATTER]++;
        RFX_KERNEL_BEGIN(c);
        if (bins) {
            switch (P.ncols) {
                case 1: rc = launch_chunk_scatter_bin<1>(c, Pc, A); break;
                case 2: rc = launch_chunk_scatter_bin<2>(c, Pc, A); break;
                case 3: rc = launch_chunk_scatter_bin<3>(c, Pc, A); break;
                default: rc = launch_chunk_scatter_bin<4>(c, Pc, A); break;
            }
        } else {
            switch (P.ncols) {
                case 1: rc = launch_chunk_scatter_sel<1>(c, Pc, A); break;
                case 2: rc = launch_chunk_scatter_sel<2>(c, Pc, A); break;
                case 3: rc = launch_chunk_scatter_sel<3>(c, Pc, A); break;
                default: rc = launch_chunk_scatter_sel<4>(c, Pc, A); break;
            }
        }
        RFX_KERNEL_END(c);
        if (rc != RFX_OK) return rc;
        RFX_HIP_CHECK(hipGetLastError());
        RFX_REQUIRE((size_t)A.nwg * sizeof(ScopePart) + 16 <= c->pin_bytes, RFX_ELIMIT, "pinned staging too small");
        ScopePart *h = (ScopePart *)c->h_pin;
        unsigned *hctl = (unsigned *)((char *)c->h_pin + (size_t)A.nwg * sizeof(ScopePart));
        RFX_HIP_CHECK(hipMemcpyAsync(h, A.parts, (size_t)A.nwg * sizeof(ScopePart), hipMemcpyDeviceToHost, c->stream));
        RFX_HIP_CHECK(hipMemcpyAsync(hctl, A.ctl, 8, hipMemcpyDeviceToHost, c->stream));
        RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (hctl[1]) return RFX_ESTATE; // pool exhausted (the sample underestimated the selection): plain passes
        fold_scope(h, A.nwg, kmin, kmax, seen, &nulls);
        if (*seen == 0 || nulls > 0) return RFX_OK;
        c->ck_valid = 1;
        c->ck_key = d_key;
        c->ck_val = (const void *)P.cols[vc];
        c->ck_nrows = nrows;
        c->ck_npred = npred;
        c->ck_logic = logic;
        c->ck_nwg = A.nwg;
        c->ck_tpw = 0;
        c->ck_chs = A.chs;
        c->ck_max_chunks = A.max_chunks;
        chunk_pred_sig(P, c->ck_sig);
        return RFX_OK;
    }
    // ---- every (or most) rows selected: the tile-sorted write-combining scatter ----
    rc = chunk_geometry(c, nrows, est_rows, &A, &tpw);
    if (rc != RFX_OK) return rc;
    A.key_idx = key_idx;
    A.vcol = vc;
    RFX_HIP_CHECK(hipMemsetAsync(A.ctl, 0, 256, c->stream));
    RFX_HIP_CHECK(hipMemsetAsync(A.meta, 0xFF, (size_t)A.max_chunks * 8, c->stream));
    c->ext_i[3 + RFX_STAT_CHUNK_SCATTER]++;
    RFX_KERNEL_BEGIN(c);
    switch (P.ncols) {
        case 1: launch_chunk_scatter<1>(c, P, A); break;
        case 2: launch_chunk_scatter<2>(c, P, A); break;
        case 3: launch_chunk_scatter<3>(c, P, A); break;
        default: launch_chunk_scatter<4>(c, P, A); break;
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    RFX_REQUIRE((size_t)A.nwg * sizeof(ScopePart) + 16 <= c->pin_bytes, RFX_ELIMIT, "pinned staging too small");
    ScopePart *h = (ScopePart *)c->h_pin;
    unsigned *hctl = (unsigned *)((char *)c->h_pin + (size_t)A.nwg * sizeof(ScopePart));
    RFX_HIP_CHECK(hipMemcpyAsync(h, A.parts, (size_t)A.nwg * sizeof(ScopePart), hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipMemcpyAsync(hctl, A.ctl, 8, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (hctl[1]) return RFX_ESTATE; // pool exhausted: plain passes
    fold_scope(h, A.nwg, kmin, kmax, seen, &nulls);
    if (*seen == 0 || nulls > 0) return RFX_OK;
    c->ck_valid = 1;
    c->ck_key = d_key;
    c->ck_val = (const void *)P.cols[vc];
    c->ck_nrows = nrows;
    c->ck_npred = npred;
    c->ck_logic = logic;
    c->ck_nwg = A.nwg;
    c->ck_tpw = tpw;
    c->ck_chs = A.chs;
    c->ck_max_chunks = A.max_chunks;
    chunk_pred_sig(P, c->ck_sig);
    return RFX_OK;
}

// Pass 2 over the partitions rfx_chunk_scope left, if they are the partitions of exactly this plan.  RFX_ESTATE: not so.
int rfx_chunk_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t) {
    {
        const int prc = rfx_plane_accumulate(c, P, key_idx, t);
        if (prc != RFX_ESTATE) {
            c->ck_valid = 0;
            return prc;
        }
    }
    if (!c->ck_valid) return RFX_ESTATE;
    bool ok = c->ck_key == (const void *)P.cols[key_idx] && c->ck_nrows == P.nrows && c->ck_npred == P.npred && c->ck_logic == P.logic && P.nx == 0;
    const int vc = ok ? single_value_col(P) : -1;
    ok = ok && vc >= 0 && c->ck_val == (const void *)P.cols[vc];
    if (ok && P.npred > 0) {
        u64 sig[RFX_MAX_PREDS][6];
        chunk_pred_sig(P, sig);
        ok = memcmp(sig, c->ck_sig, sizeof(u64) * 6 * (size_t)P.npred) == 0;
    }
    c->ck_valid = 0; // consumed (or stale) either way
    if (!ok || t->range <= 256) return RFX_ESTATE;
    const int narr = narr_of(P);
    const i64 local = (t->range + 255) >> 8;
    const size_t lds = (size_t)narr * (size_t)local * 8;
    if (lds > PART_LDS_BIG_BYTES) return RFX_ESTATE;
    ChunkArgs A;
    memset(&A, 0, sizeof(A));
    A.chs = c->ck_chs;
    chunk_layout(c, c->ck_nwg, c->ck_max_chunks, &A, NULL);
    ChunkAggArgs G;
    memset(&G, 0, sizeof(G));
    G.kmin = t->kmin;
    G.range = t->range;
    G.local = local;
    G.chs = A.chs;
    G.pool = A.pool;
    G.part_start = A.part_start;
    G.plist = A.plist;
    G.first = (u64 *)t->d_first;
    for (int a = 0; a < t->nagg; a++) {
        G.acc[a] = (u64 *)t->d_acc[a];
        G.cnt[a] = (u64 *)t->d_cnt[a];
    }
    c->ext_i[3 + RFX_STAT_CHUNK_AGGREGATE]++;
    RFX_KERNEL_BEGIN(c);
    hipLaunchKernelGGL(k_chunk_offsets, dim3(1), dim3(1024), 0, c->stream, A);
    int pgrid = (int)((c->ck_max_chunks + RFX_BLOCK - 1) / RFX_BLOCK);
    if (pgrid > c->num_cus * 8) pgrid = c->num_cus * 8;
    hipLaunchKernelGGL(k_chunk_place, dim3(pgrid), dim3(RFX_BLOCK), 0, c->stream, A);
    if (lds > PART_LDS_BYTES) {
        static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */
        if (!((attr_set >> (c->device & 63)) & 1ull)) {
            RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chunk_aggregate<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);
        }
        G.split = 1;
        hipLaunchKernelGGL((k_chunk_aggregate<1024>), dim3(CK_PARTS), dim3(1024), lds, c->stream, P, G);
    } else {
        G.split = (2 * c->num_cus + CK_PARTS - 1) / CK_PARTS;
        if (G.split < 1) G.split = 1;
        hipLaunchKernelGGL((k_chunk_aggregate<512>), dim3(CK_PARTS * G.split), dim3(512), lds, c->stream, P, G);
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
