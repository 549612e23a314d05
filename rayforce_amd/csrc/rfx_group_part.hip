// rfx_group_part.hip -- radix-partitioned dense group-by for key ranges whose tables do not fit one workgroup's LDS
// (e.g. BASELINE config C3: 1e9 rows, 1e6 distinct i64 keys, sum(f64)).
//
// Why: tools/probe_hw measured device-scope f64/u64 atomics into an 8 MB table at 23 G rows/s (12 G rows/s with the
// first-row atomicMin on top) -- 20-40x off the 16 B/row HBM roofline -- while LDS-privatised tables run at the HBM
// streaming rate (400+ G rows/s).  So rows are first routed to the workgroup that owns their key range:
//
//   pass 0  k_part_hist      read key (+ predicate columns): per-(workgroup, partition) row counts          8 B/row read
//           k_part_offsets   column-wise exclusive scan -> exact output offset of every (workgroup, partition)
//   pass 1  k_part_scatter   read key + value columns, build records {local_row:32 | local_slot:32} + values,
//                            sort each 2048-row tile by partition in LDS, write runs to the partition's region
//                            (coalesced, no atomics, no overflow: offsets are exact)                16 B/row read + 16 B/row write
//   pass 2  k_part_aggregate each partition (a contiguous record range) is streamed by SPLIT workgroups that
//                            aggregate into LDS tables (ds_add_f64 / ds_min_u64 ...) and merge them into the global
//                            tables once                                                              16 B/row read
// What bounds pass 1 (rocprofv3 PMC, profiles/): TCC_EA0_WRREQ_STALL ~ TCC_EA0_WRREQ -- the L2 -> memory write path is
// back-pressured by 64-byte write requests that arrive in no DRAM-page order (the same kernel writing each tile
// contiguously runs in 6.1 ms instead of ~9 ms).  Tried and measured without gain: 4096/8192-row tiles, 528-byte runs
// (123 partitions), a workgroup-major record layout, randomised segment padding, register prefetch of the next tile,
// direct (unsorted) stores.  Next lever: per-partition write-combining to >= 1 KB before the store.
// partition p = (key - kmin) >> lb owns slots [p << lb, (p+1) << lb): the per-partition LDS table covers them exactly.
// The global tables, the first-row ranking and the emit are those of rfx_group.hip, so results (group order included)
// are identical to the LDS-direct and atomic paths.
#include "rfx_part_common.hpp"

// ---- pass 0: per-(workgroup, partition) counts ----
template <int NC, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_part_hist(const Plan P, const PartArgs A) {
    __shared__ unsigned hist[PART_MAX];
    PredSet<NP> S;
    predset_load<NP>(P, S);
    for (int i = threadIdx.x; i < A.nparts; i += RFX_BLOCK) hist[i] = 0;
    __syncthreads();
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        u64 v[NC][8];
        const unsigned m = part_load_eval<NC, NP>(P, S, t, v);
        u64 key[8];
        sel_col<NC, 8>(key, v, A.key_idx);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 slot = key[e] - (u64)A.kmin;
            if (((m >> e) & 1u) && part_row_ok(A, slot)) atomicAdd(&hist[part_of(A, key[e], slot)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A.nparts; i += RFX_BLOCK) A.offsets[(size_t)blockIdx.x * A.nparts + i] = hist[i];
}

// ---- pass 0 fused with the key scope (index_scope_i64, core/index.c:376-435) ----
// Partitioning on the key's LOW 8 bits needs no kmin, so the histogram can be taken in the same streaming read that finds
// min / max: one 8 B/row pass instead of two.  Per workgroup: counts[256] + {min, max, selected, null keys}.
template <int NC, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_part_scope_hist(const Plan P, int key_idx, u64 *__restrict__ counts, ScopePart *__restrict__ parts, u64 *__restrict__ bitmap) {
    __shared__ unsigned hist[256];
    __shared__ ScopePart red[RFX_BLOCK / RFX_WAVE];
    PredSet<NP> S;
    predset_load<NP>(P, S);
    hist[threadIdx.x] = 0; // RFX_BLOCK == 256
    __syncthreads();
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0, nulls = 0;
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        u64 v[NC][8];
        const unsigned m = part_load_eval<NC, NP>(P, S, t, v);
        if (NP > 0 && bitmap) {
            // the selection as `where`'s bitmap ("pair-split 128": word 2g = even rows of the 128-row group g, word 2g+1 = odd
            // rows): a wave's rows of load j ARE one such group, so the two ballots are its two words.  A selective filter
            // then compacts by this bitmap instead of evaluating the predicates a second time.
            const i64 g0 = t * (PART_TILE_ROWS / 128) + (threadIdx.x >> 6);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u64 b0 = __ballot((m >> (2 * j)) & 1u), b1 = __ballot((m >> (2 * j + 1)) & 1u);
                if ((threadIdx.x & 63) == 0) {
                    u64x2 w;
                    w.x = b0;
                    w.y = b1;
                    *(u64x2 *)(bitmap + 2 * (g0 + j * 4)) = w;
                }
            }
        }
        u64 key[8];
        sel_col<NC, 8>(key, v, key_idx);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (!((m >> e) & 1u)) continue;
            const i64 k = (i64)key[e];
            atomicAdd(&hist[key[e] & 255ULL], 1u);
            sel++;
            if (k == RFX_NULL_I64_D) nulls++;
            else {
                mn = k < mn ? k : mn;
                mx = k > mx ? k : mx;
            }
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s), omx = (i64)rfx_shfl_xor_u64((u64)mx, s);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sel += (i64)rfx_shfl_xor_u64((u64)sel, s);
        nulls += (i64)rfx_shfl_xor_u64((u64)nulls, s);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ScopePart{mn, mx, sel, nulls};
    __syncthreads();
    counts[(size_t)blockIdx.x * 256 + threadIdx.x] = hist[threadIdx.x];
    if (threadIdx.x == 0) {
        ScopePart r = red[0];
        for (int w = 1; w < RFX_BLOCK / RFX_WAVE; w++) {
            r.mn = red[w].mn < r.mn ? red[w].mn : r.mn;
            r.mx = red[w].mx > r.mx ? red[w].mx : r.mx;
            r.sel += red[w].sel;
            r.nulls += red[w].nulls;
        }
        parts[blockIdx.x] = r;
    }
}

template <int NC>
static void launch_scope_hist(rfx_ctx *c, const Plan &P, int key_idx, int nwg, ScopePart *parts, u64 *bitmap) {
    if (P.npred == 0) hipLaunchKernelGGL((k_part_scope_hist<NC, 0>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, c->d_pc_counts, parts, (u64 *)0);
    else if (P.npred == 1) hipLaunchKernelGGL((k_part_scope_hist<NC, 1>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, c->d_pc_counts, parts, bitmap); // one predicate: the C3w shape
    else hipLaunchKernelGGL((k_part_scope_hist<NC, RFX_MAX_PREDS>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, key_idx, c->d_pc_counts, parts, bitmap);
}

static inline int part_nwg(const rfx_ctx *c) { return c->num_cus * ((c->flags & RFX_TUNE_PART_3WG) ? 3 : 2); } // in-process A/B: 2 per CU beat 3 by ~3 %

// predicate signature (column POINTERS, not plan indices: the scope plan and the accumulate plan order columns differently)
static void plan_pred_sig(const Plan &P, u64 (*sig)[6]) {
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

// Called by rfx_hip_scope_i64 for inputs the partitioned path will take.  RFX_ESTATE = not applicable.
int rfx_part_scope_hist(rfx_ctx *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, i64 nrows, i64 *kmin, i64 *kmax,
                        i64 *seen) {
    c->pc_valid = 0;
    if ((c->flags & (RFX_TUNE_NO_PARTITION | RFX_TUNE_NO_FUSED_SCOPE)) || nrows >= (1LL << 32) || nrows < (1 << 16)) return RFX_ESTATE;
    Plan P;
    int key_idx = 0;
    int rc = rfx_plan_build(&P, preds, npred, logic, NULL, 0, d_key, &key_idx, nrows, 0);
    if (rc != RFX_OK) return rc;
    if (P.ncols > 4) return RFX_ESTATE;
    const int nwg = part_nwg(c);
    if (!c->d_pc_counts) RFX_HIP_CHECK(hipMalloc((void **)&c->d_pc_counts, (size_t)1024 * 3 * 256 * 8));
    rc = rfx_ws_reserve(c, (size_t)nwg * sizeof(ScopePart));
    if (rc != RFX_OK) return rc;
    ScopePart *d_parts = (ScopePart *)c->d_ws;
    u64 *bitmap = NULL;
    c->pc_bitmap = 0;
    if (P.npred > 0) { // the selection bitmap as a side product (whole 2048-row tiles are written)
        // sized as where_reserve (rfx_where.hip) sizes it: rfx_where_counts_of_bitmap must find the buffer big enough, a re-allocation there would lose the bitmap
        rc = rfx_bitmap_reserve(c, ((nrows + 4095) / 4096) * 4096);
        if (rc != RFX_OK) return rc;
        c->where_n = -1; // a pending where_begin / where_emit pair loses its bitmap
        bitmap = c->d_bitmap;
    }
    RFX_KERNEL_BEGIN(c);
    switch (P.ncols) {
        case 1: launch_scope_hist<1>(c, P, key_idx, nwg, d_parts, bitmap); break;
        case 2: launch_scope_hist<2>(c, P, key_idx, nwg, d_parts, bitmap); break;
        case 3: launch_scope_hist<3>(c, P, key_idx, nwg, d_parts, bitmap); break;
        default: launch_scope_hist<4>(c, P, key_idx, nwg, d_parts, bitmap); break;
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    static_assert(sizeof(ScopePart) == 32, "ScopePart layout");
    RFX_REQUIRE((size_t)nwg * sizeof(ScopePart) <= c->pin_bytes, RFX_ELIMIT, "pinned staging too small");
    ScopePart *h = (ScopePart *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, d_parts, (size_t)nwg * sizeof(ScopePart), hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, sel = 0, nulls = 0;
    for (int w = 0; w < nwg; w++) {
        mn = h[w].mn < mn ? h[w].mn : mn;
        mx = h[w].mx > mx ? h[w].mx : mx;
        sel += h[w].sel;
        nulls += h[w].nulls;
    }
    *seen = sel;
    if (nulls > 0) { // a null key is the value INT64_MIN for index_scope_i64
        mn = RFX_NULL_I64_D;
        if (nulls == sel) mx = RFX_NULL_I64_D;
    }
    *kmin = mn;
    *kmax = mx;
    if (sel > 0 && nulls == 0) {
        c->pc_valid = 1;
        c->pc_bitmap = bitmap != NULL;
        c->pc_seen = sel;
        c->pc_key = d_key;
        c->pc_nrows = nrows;
        c->pc_npred = npred;
        c->pc_logic = logic;
        c->pc_nwg = nwg;
        plan_pred_sig(P, c->pc_sig);
    }
    return RFX_OK;
}

// ---- offsets: offsets[w][p] = part_start[p] + sum_{w' < w} counts[w'][p] ----
// Exclusive scan down every partition's column of the [nwg][nparts] count matrix.  One thread per partition walking nwg
// dependent loads was 0.18 ms; here a workgroup takes 32 partitions x 8 slices of workgroups: slice sums, an 8-step LDS scan,
// then the slice's own running offsets.
#define COLSCAN_P 32
#define COLSCAN_S (RFX_BLOCK / COLSCAN_P)
__global__ __launch_bounds__(RFX_BLOCK) void k_part_colscan(const PartArgs A, int nwg) {
    __shared__ u64 slice_sum[COLSCAN_S][COLSCAN_P];
    const int pl = threadIdx.x % COLSCAN_P, sl = threadIdx.x / COLSCAN_P;
    const int p = blockIdx.x * COLSCAN_P + pl;
    const int per = (nwg + COLSCAN_S - 1) / COLSCAN_S;
    const int w0 = sl * per, w1 = (w0 + per < nwg) ? w0 + per : nwg;
    const bool live = p < A.nparts;
    u64 sum = 0;
    if (live)
        for (int w = w0; w < w1; w++) {
            u64 c = A.offsets[(size_t)w * A.nparts + p];
            if (A.wc) c = (c + A.wc - 1) / A.wc * A.wc; // every (workgroup, partition) region is whole 128-byte store groups
            sum += c;
        }
    slice_sum[sl][pl] = sum;
    __syncthreads();
    u64 run = 0, total = 0;
    for (int s2 = 0; s2 < COLSCAN_S; s2++) {
        const u64 v = slice_sum[s2][pl];
        if (s2 < sl) run += v;
        total += v;
    }
    if (!live) return;
    for (int w = w0; w < w1; w++) {
        const size_t i = (size_t)w * A.nparts + p;
        u64 c = A.offsets[i];
        if (A.wc) c = (c + A.wc - 1) / A.wc * A.wc;
        A.offsets[i] = run;
        run += c;
    }
    if (sl == 0) A.part_start[p] = total; // column total, scanned by k_part_startscan
}
__global__ __launch_bounds__(PART_MAX) void k_part_startscan(const PartArgs A) {
    __shared__ u64 tmp[PART_MAX + 1];
    const int p = threadIdx.x;
    tmp[p] = (p < A.nparts) ? A.part_start[p] : 0;
    __syncthreads();
    // Hillis-Steele inclusive scan over <= 1024 entries
    for (int s = 1; s < PART_MAX; s <<= 1) {
        u64 add = (p >= s) ? tmp[p - s] : 0;
        __syncthreads();
        tmp[p] += add;
        __syncthreads();
    }
    if (p < A.nparts) A.part_start[p + 1] = tmp[p];
    if (p == 0) A.part_start[0] = 0;
}

// ---- pass 1: scatter records, tile sorted by partition in LDS ----
template <int NC, int NV, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_part_scatter(const Plan P, const PartArgs A) {
    __shared__ unsigned thist[PART_MAX]; // per tile: count, then exclusive tile offset
    __shared__ u64 cursor[PART_MAX];     // running global output position of (this workgroup, partition)
    __shared__ u64 stag[(1 + NV) * PART_TILE_ROWS];
    __shared__ unsigned short stag_p[PART_TILE_ROWS];
    __shared__ unsigned scan_w[RFX_BLOCK / RFX_WAVE];
    __shared__ unsigned tile_total;
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x;
    for (int i = tid; i < A.nparts; i += RFX_BLOCK) cursor[i] = A.part_start[i] + A.offsets[(size_t)blockIdx.x * A.nparts + i];
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int i = tid; i < A.nparts; i += RFX_BLOCK) thist[i] = 0;
        __syncthreads();
        u64 v[NC][8];
        const unsigned m0 = part_load_eval<NC, NP>(P, S, t, v);
        u64 key[8];
        sel_col<NC, 8>(key, v, A.key_idx);
        unsigned m = 0, part[8], rank[8];
        const i64 base = t * PART_TILE_ROWS + tid * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 slot = key[e] - (u64)A.kmin;
            part[e] = 0;
            rank[e] = 0;
            if (((m0 >> e) & 1u) && part_row_ok(A, slot)) {
                m |= 1u << e;
                part[e] = part_of(A, key[e], slot);
                rank[e] = atomicAdd(&thist[part[e]], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of thist over nparts (<= 1024): 4 entries per lane
        {
            unsigned x[4], s = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int idx = tid * 4 + i;
                x[i] = (idx < A.nparts) ? thist[idx] : 0;
                s += x[i];
            }
            unsigned inc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                unsigned o = __shfl_up(inc, d, 64);
                if ((tid & 63) >= d) inc += o;
            }
            if ((tid & 63) == 63) scan_w[tid >> 6] = inc;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < (tid >> 6); w++) wbase += scan_w[w];
            unsigned ex = wbase + inc - s;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int idx = tid * 4 + i;
                if (idx < A.nparts) thist[idx] = ex;
                ex += x[i];
            }
            if (tid == RFX_BLOCK - 1) tile_total = ex;
        }
        __syncthreads();
        // stage the records in partition order
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (!((m >> e) & 1u)) continue;
            const unsigned idx = thist[part[e]] + rank[e];
            const u64 slot = key[e] - (u64)A.kmin;
            const u64 lrow = (u64)(base + (i64)(e >> 1) * (RFX_BLOCK * 2) + (e & 1));
            stag[idx] = (lrow << 32) | (A.lowbit ? (slot >> 8) : (slot & ((1ULL << A.lb) - 1)));
            stag_p[idx] = (unsigned short)part[e];
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            u64 x[8];
            sel_col<NC, 8>(x, v, A.vcol[j]);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if ((m >> e) & 1u) stag[(1 + j) * PART_TILE_ROWS + thist[part[e]] + rank[e]] = x[e];
            }
        }
        __syncthreads();
        // write out: consecutive staged records of one partition go to consecutive global positions
        const unsigned total = tile_total;
        for (unsigned i = tid; i < total; i += RFX_BLOCK) {
            const unsigned p = stag_p[i];
            const u64 dst = cursor[p] + (i - thist[p]);
            if (NV == 1) {
                // array-of-structures: one 16-byte store per record {header, value}
                u64x2 r;
                r.x = stag[i];
                r.y = stag[PART_TILE_ROWS + i];
                *(u64x2 *)(A.recs + 2 * dst) = r;
            } else {
                A.recs[dst] = stag[i];
#pragma unroll
                for (int j = 0; j < NV; j++) A.recs[(size_t)(1 + j) * A.cap + dst] = stag[(1 + j) * PART_TILE_ROWS + i];
            }
        }
        __syncthreads();
        // advance the cursors by this tile's counts (count of p = next offset - this offset)
        for (int i = tid; i < A.nparts; i += RFX_BLOCK) {
            const unsigned nxt = (i + 1 < A.nparts) ? thist[i + 1] : total;
            cursor[i] += nxt - thist[i];
        }
        __syncthreads();
    }
}

// ---- pass 1, write-combining form (one value plane, <= 256 partitions: the C3 shape) ----
// rocprofv3 PMC A/B with IDENTICAL instruction streams (profiles/): the sorted-tile scatter above
// spends 11.7 ms where the same kernel storing each tile contiguously spends 6.3 ms.  The difference is entirely in the
// L2: a partition's run of ~8 records starts at an arbitrary 16-byte offset, so 64-byte sectors are completed by two
// different store instructions (or evicted half-written): TCC_WRITE 354 M vs 250 M requests, 91 M of the 370 M fabric
// writes are 32-byte partials, TCC_EA0_WRREQ_STALL 346 M vs 135 M cycles.
// Here each (workgroup, partition) keeps its last < 8 records in an LDS carry buffer and only ever stores whole,
// 128-byte aligned groups of 8 records; the tail is flushed once at the end, padded with sentinel records
// (slot field 0xFFFFFFFF) that pass 2 skips.  Every (workgroup, partition) region is padded to a multiple of 8 records.
template <int NC, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_part_scatter_wc(const Plan P, const PartArgs A) {
    __shared__ unsigned thist[WC_MAXP];   // per tile: count, then exclusive tile offset
    __shared__ unsigned tcnt[WC_MAXP];    // per tile: count
    __shared__ unsigned pre[WC_MAXP];     // records carried over from earlier tiles (< WC_B)
    __shared__ unsigned pfl[WC_MAXP];     // records of (carry ++ tile) that leave for global memory now (multiple of WC_B)
    __shared__ u64 cursor[WC_MAXP];       // next global record index of (this workgroup, partition), multiple of WC_B
    __shared__ u64x2 carry[WC_MAXP][WC_B];
    __shared__ u64x2 stag[PART_TILE_ROWS];
    __shared__ unsigned short stag_p[PART_TILE_ROWS];
    __shared__ unsigned scan_w[RFX_BLOCK / RFX_WAVE];
    __shared__ unsigned tile_total;
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x;
    const int np = A.nparts;
    u64x2 *__restrict__ recs = (u64x2 *)A.recs;
    if (tid < np) {
        cursor[tid] = A.part_start[tid] + A.offsets[(size_t)blockIdx.x * np + tid];
        pre[tid] = 0;
    }
    __syncthreads(); // workgroups beyond the last tile skip the loop below: the tail pass must still see pre[] == 0
    const int vc = A.vcol[0];
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (tid < np) thist[tid] = 0;
        __syncthreads();
        u64 v[NC][8];
        const unsigned m0 = part_load_eval<NC, NP>(P, S, t, v);
        u64 key[8], val[8];
        sel_col<NC, 8>(key, v, A.key_idx);
        sel_col<NC, 8>(val, v, vc);
        unsigned m = 0, part[8], rank[8];
        const i64 base = t * PART_TILE_ROWS + tid * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 slot = key[e] - (u64)A.kmin;
            part[e] = 0;
            rank[e] = 0;
            if (((m0 >> e) & 1u) && part_row_ok(A, slot)) {
                m |= 1u << e;
                part[e] = part_of(A, key[e], slot);
                rank[e] = atomicAdd(&thist[part[e]], 1u);
            }
        }
        __syncthreads();
        // one lane per partition: exclusive scan of the tile counts + the flush decision
        {
            const unsigned x = (tid < np) ? thist[tid] : 0;
            unsigned inc = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                unsigned o = __shfl_up(inc, d, 64);
                if ((tid & 63) >= d) inc += o;
            }
            if ((tid & 63) == 63) scan_w[tid >> 6] = inc;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < (tid >> 6); w++) wbase += scan_w[w];
            if (tid < np) {
                thist[tid] = wbase + inc - x;
                tcnt[tid] = x;
                pfl[tid] = ((pre[tid] + x) / WC_B) * WC_B;
            }
            if (tid == RFX_BLOCK - 1) tile_total = wbase + inc;
        }
        __syncthreads();
        // stage this tile's records in partition order
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (!((m >> e) & 1u)) continue;
            const unsigned idx = thist[part[e]] + rank[e];
            const u64 slot = key[e] - (u64)A.kmin;
            const u64 lrow = (u64)(base + (i64)(e >> 1) * (RFX_BLOCK * 2) + (e & 1));
            u64x2 r;
            r.x = (lrow << 32) | (A.lowbit ? (slot >> 8) : (slot & ((1ULL << A.lb) - 1)));
            r.y = val[e];
            stag[idx] = r;
            stag_p[idx] = (unsigned short)part[e];
        }
        // old carry -> global for the partitions that flush (positions 0 .. pre-1 of their sequence)
        for (int idx = tid; idx < np * WC_B; idx += RFX_BLOCK) {
            const int p = idx / WC_B, j = idx % WC_B;
            if ((unsigned)j < pre[p] && pfl[p] > 0) recs[cursor[p] + j] = carry[p][j];
        }
        __syncthreads();
        // new records: the first (pfl - pre) of a partition complete the 128-byte groups, the rest is the new carry
        const unsigned total = tile_total;
        for (unsigned i = tid; i < total; i += RFX_BLOCK) {
            const unsigned p = stag_p[i];
            const unsigned pos = pre[p] + (i - thist[p]);
            if (pos < pfl[p]) recs[cursor[p] + pos] = stag[i];
            else carry[p][pos - pfl[p]] = stag[i];
        }
        __syncthreads();
        if (tid < np) {
            const unsigned tot = pre[tid] + tcnt[tid];
            cursor[tid] += pfl[tid];
            pre[tid] = tot - pfl[tid];
        }
        __syncthreads();
    }
    // tails: one padded 128-byte group per partition that still carries records
    for (int idx = tid; idx < np * WC_B; idx += RFX_BLOCK) {
        const int p = idx / WC_B, j = idx % WC_B;
        if (pre[p] > 0) {
            u64x2 r;
            r.x = WC_SENTINEL;
            r.y = 0;
            recs[cursor[p] + j] = ((unsigned)j < pre[p]) ? carry[p][j] : r;
        }
    }
}

// ---- pass 1, direct write-combining form (1..3 value planes, <= 256 partitions) ----
// Same 128-byte store groups as k_part_scatter_wc, but records go from registers straight to their final place
// (global group or LDS carry): position inside the partition = carried records + rank from the LDS counter, so no tile
// sort, no staging buffer, no scan.  LDS: 37 KB (4 workgroups per CU instead of 2), 5 barriers per tile instead of 7.
// Records are array-of-structures: {header, v0} = 16 B for one value plane, {header, v0, v1, v2} = 32 B for two or three.
template <int NC, int NV, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_part_scatter_dwc(const Plan P, const PartArgs A) {
    constexpr int RSU = (NV == 1) ? 2 : 4;  // u64 per record
    constexpr int GB = 16 / RSU;            // records per 128-byte store group
    __shared__ unsigned cnt[WC_MAXP];       // records of this tile per partition
    __shared__ unsigned pre[WC_MAXP];       // records carried over (< GB)
    __shared__ unsigned pfl[WC_MAXP];       // records of (carry ++ tile) that leave now (multiple of GB)
    __shared__ u64 cursor[WC_MAXP];         // next global record index, multiple of GB
    __shared__ __attribute__((aligned(16))) u64 carry[WC_MAXP][16];
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x;
    const int np = A.nparts;
    u64 *__restrict__ recs = A.recs;
    if (tid < np) {
        cursor[tid] = A.part_start[tid] + A.offsets[(size_t)blockIdx.x * np + tid];
        pre[tid] = 0;
        cnt[tid] = 0;
    }
    __syncthreads();
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        u64 v[NC][8];
        const unsigned m0 = part_load_eval<NC, NP>(P, S, t, v);
        u64 key[8];
        sel_col<NC, 8>(key, v, A.key_idx);
        unsigned m = 0, part[8], rank[8];
        const i64 base = t * PART_TILE_ROWS + tid * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 slot = key[e] - (u64)A.kmin;
            part[e] = 0;
            rank[e] = 0;
            if (((m0 >> e) & 1u) && part_row_ok(A, slot)) {
                m |= 1u << e;
                part[e] = part_of(A, key[e], slot);
                rank[e] = atomicAdd(&cnt[part[e]], 1u);
            }
        }
        __syncthreads();
        if (tid < np) pfl[tid] = ((pre[tid] + cnt[tid]) / GB) * GB;
        __syncthreads();
        // old carry -> global for the partitions that complete at least one group now
        for (int idx = tid; idx < np * GB; idx += RFX_BLOCK) {
            const int p = idx / GB, j = idx % GB;
            if ((unsigned)j < pre[p] && pfl[p] > 0) {
                u64 *dst = recs + (cursor[p] + j) * RSU;
#pragma unroll
                for (int q = 0; q < RSU; q += 2) *(u64x2 *)(dst + q) = *(const u64x2 *)(&carry[p][j * RSU + q]);
            }
        }
        __syncthreads();
        // this tile's records, straight from registers
        u64 val[NV][8];
#pragma unroll
        for (int j = 0; j < NV; j++) sel_col<NC, 8>(val[j], v, A.vcol[j]);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (!((m >> e) & 1u)) continue;
            const unsigned p = part[e];
            const unsigned pos = pre[p] + rank[e];
            const u64 slot = key[e] - (u64)A.kmin;
            const u64 lrow = (u64)(base + (i64)(e >> 1) * (RFX_BLOCK * 2) + (e & 1));
            u64 rec[4];
            rec[0] = (lrow << 32) | (A.lowbit ? (slot >> 8) : (slot & ((1ULL << A.lb) - 1)));
            rec[1] = val[0][e];
            rec[2] = (NV > 1) ? val[NV > 1 ? 1 : 0][e] : 0;
            rec[3] = (NV > 2) ? val[NV > 2 ? 2 : 0][e] : 0;
            if (pos < pfl[p]) {
                u64 *dst = recs + (cursor[p] + pos) * RSU;
#pragma unroll
                for (int q = 0; q < RSU; q += 2) {
                    u64x2 w;
                    w.x = rec[q];
                    w.y = rec[q + 1];
                    *(u64x2 *)(dst + q) = w;
                }
            } else {
                const unsigned ci = (pos - pfl[p]) * RSU;
#pragma unroll
                for (int q = 0; q < RSU; q++) carry[p][ci + q] = rec[q];
            }
        }
        __syncthreads();
        if (tid < np) {
            const unsigned tot = pre[tid] + cnt[tid];
            cursor[tid] += pfl[tid];
            pre[tid] = tot - pfl[tid];
            cnt[tid] = 0;
        }
        __syncthreads();
    }
    // tails: one padded 128-byte group per partition that still carries records
    for (int idx = tid; idx < np * GB; idx += RFX_BLOCK) {
        const int p = idx / GB, j = idx % GB;
        if (pre[p] > 0) {
            u64 *dst = recs + (cursor[p] + j) * RSU;
#pragma unroll
            for (int q = 0; q < RSU; q += 2) {
                u64x2 w;
                if ((unsigned)j < pre[p]) w = *(const u64x2 *)(&carry[p][j * RSU + q]);
                else { w.x = WC_SENTINEL; w.y = 0; }
                *(u64x2 *)(dst + q) = w;
            }
        }
    }
}

// ---- pass 1, tile-sorted write-combining form for SEVERAL value planes (structure of arrays) ----
// Two or three value planes made the records 32 bytes and the scatter register-direct (k_part_scatter_dwc): every record a
// 32-byte store of its own, 20 ms per 1e9 rows.  Here the tile is sorted by partition once (as in k_part_scatter_wc), then
// each PLANE -- headers, then every value plane -- goes through the same 16 KB staging buffer and leaves as whole aligned
// 64-byte groups of 8 x 8 bytes, with an 8-record LDS carry per (partition, plane) between tiles.  LDS 26 KB + 16 KB per
// plane: two workgroups per CU with two value planes.
template <int NC, int NV, int NP>
__global__ __launch_bounds__(RFX_BLOCK) void k_part_scatter_soa(const Plan P, const PartArgs A) {
    constexpr int NPL = 1 + NV;
    __shared__ unsigned thist[WC_MAXP];   // per tile: count, then exclusive tile offset
    __shared__ unsigned tcnt[WC_MAXP];
    __shared__ unsigned pre[WC_MAXP];     // records carried over from earlier tiles (< WC_B)
    __shared__ unsigned pfl[WC_MAXP];     // records of (carry ++ tile) that leave now (multiple of WC_B)
    __shared__ u64 cursor[WC_MAXP];       // next record index of (this workgroup, partition), multiple of WC_B
    __shared__ u64 carry[NPL][WC_MAXP][WC_B];
    __shared__ u64 stag[PART_TILE_ROWS];
    __shared__ unsigned short stag_p[PART_TILE_ROWS];
    __shared__ unsigned scan_w[RFX_BLOCK / RFX_WAVE];
    __shared__ unsigned tile_total;
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x;
    const int np = A.nparts;
    u64 *__restrict__ recs = A.recs;
    const size_t cap = (size_t)A.cap;
    if (tid < np) {
        cursor[tid] = A.part_start[tid] + A.offsets[(size_t)blockIdx.x * np + tid];
        pre[tid] = 0;
    }
    __syncthreads();
    const i64 ntiles = (P.nrows + PART_TILE_ROWS - 1) / PART_TILE_ROWS;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (tid < np) thist[tid] = 0;
        __syncthreads();
        u64 v[NC][8];
        const unsigned m0 = part_load_eval<NC, NP>(P, S, t, v);
        u64 key[8];
        sel_col<NC, 8>(key, v, A.key_idx);
        unsigned m = 0, part[8], rank[8];
        const i64 base = t * PART_TILE_ROWS + tid * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 slot = key[e] - (u64)A.kmin;
            part[e] = 0;
            rank[e] = 0;
            if (((m0 >> e) & 1u) && part_row_ok(A, slot)) {
                m |= 1u << e;
                part[e] = part_of(A, key[e], slot);
                rank[e] = atomicAdd(&thist[part[e]], 1u);
            }
        }
        __syncthreads();
        {
            const unsigned x = (tid < np) ? thist[tid] : 0;
            unsigned inc = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                unsigned o = __shfl_up(inc, d, 64);
                if ((tid & 63) >= d) inc += o;
            }
            if ((tid & 63) == 63) scan_w[tid >> 6] = inc;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < (tid >> 6); w++) wbase += scan_w[w];
            if (tid < np) {
                thist[tid] = wbase + inc - x;
                tcnt[tid] = x;
                pfl[tid] = ((pre[tid] + x) / WC_B) * WC_B;
            }
            if (tid == RFX_BLOCK - 1) tile_total = wbase + inc;
        }
        __syncthreads();
        const unsigned total = tile_total;
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            // this plane's values of the tile, in partition order
            u64 x[8];
            if (pl == 0) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const u64 slot = key[e] - (u64)A.kmin;
                    const u64 lrow = (u64)(base + (i64)(e >> 1) * (RFX_BLOCK * 2) + (e & 1));
                    x[e] = (lrow << 32) | (A.lowbit ? (slot >> 8) : (slot & ((1ULL << A.lb) - 1)));
                }
            } else sel_col<NC, 8>(x, v, A.vcol[pl - 1]);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (!((m >> e) & 1u)) continue;
                const unsigned idx = thist[part[e]] + rank[e];
                stag[idx] = x[e];
                if (pl == 0) stag_p[idx] = (unsigned short)part[e];
            }
            u64 *__restrict__ out = recs + (size_t)pl * cap;
            // old carry -> global for the partitions that complete a group now
            for (int idx = tid; idx < np * WC_B; idx += RFX_BLOCK) {
                const int p = idx / WC_B, j = idx % WC_B;
                if ((unsigned)j < pre[p] && pfl[p] > 0) out[cursor[p] + j] = carry[pl][p][j];
            }
            __syncthreads();
            for (unsigned i = tid; i < total; i += RFX_BLOCK) {
                const unsigned p = stag_p[i];
                const unsigned pos = pre[p] + (i - thist[p]);
                if (pos < pfl[p]) out[cursor[p] + pos] = stag[i];
                else carry[pl][p][pos - pfl[p]] = stag[i];
            }
            __syncthreads();
        }
        if (tid < np) {
            const unsigned tot = pre[tid] + tcnt[tid];
            cursor[tid] += pfl[tid];
            pre[tid] = tot - pfl[tid];
        }
        __syncthreads();
    }
    // tails: one padded group per partition that still carries records (sentinel headers; the value planes' padding is never read)
    for (int idx = tid; idx < np * WC_B; idx += RFX_BLOCK) {
        const int p = idx / WC_B, j = idx % WC_B;
        if (pre[p] > 0) {
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) recs[(size_t)pl * cap + cursor[p] + j] = ((unsigned)j < pre[p]) ? carry[pl][p][j] : (pl == 0 ? WC_SENTINEL : 0ULL);
        }
    }
}
template <int NC, int NV>
static void launch_scatter_soa(rfx_ctx *c, const Plan &P, const PartArgs &A, int nwg) {
    if (P.npred == 0) hipLaunchKernelGGL((k_part_scatter_soa<NC, NV, 0>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
    else hipLaunchKernelGGL((k_part_scatter_soa<NC, NV, RFX_MAX_PREDS>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
}

// ---- pass 2: per-partition LDS aggregation ----
template <int NV, int THREADS>
__global__ __launch_bounds__(THREADS) void k_part_aggregate(const Plan P, const PartArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x / A.split, s = blockIdx.x % A.split;
    const i64 local = 1LL << A.lb;
    // descriptors into registers once (static indices only inside the record loop)
    int kind[RFX_MAX_AGGS], f64[RFX_MAX_AGGS], plane[RFX_MAX_AGGS], arr_of[RFX_MAX_AGGS], skip[RFX_MAX_AGGS];
    {
        int arr = 1;
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            kind[a] = (a < P.nagg) ? P.aggs[a].kind : -1;
            f64[a] = (a < P.nagg) ? P.aggs[a].f64 : 0;
            skip[a] = (a < P.nagg) ? P.aggs[a].skipnull : 0;
            plane[a] = (a < P.nagg) ? A.agg_plane[a] : -1;
            arr_of[a] = arr;
            if (kind[a] >= 0) arr += agg_has_cnt(kind[a], f64[a]) ? 2 : 1;
        }
    }
    // LDS layout: [first | acc0 | (cnt0) | acc1 ...] each `local` cells
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
    const u64 *__restrict__ recs = A.recs;
    const size_t cap = (size_t)A.cap;
    constexpr int RU = 4; // records in flight per lane
    for (u64 i0 = b0; i0 < b1; i0 += (u64)THREADS * RU) {
        u64 h[RU], val[RU][NV > 0 ? NV : 1];
        bool in[RU];
#pragma unroll
        for (int r = 0; r < RU; r++) {
            const u64 i = i0 + (u64)r * THREADS + tid;
            in[r] = i < b1;
            h[r] = 0;
            if (!in[r]) continue;
            if (NV == 1) {
                typedef u64 v2 __attribute__((ext_vector_type(2)));
                const v2 q = __builtin_nontemporal_load((const v2 *)(recs + 2 * i));
                h[r] = q.x;
                val[r][0] = q.y;
            } else if (NV > 1 && A.wc && !A.soa) {
                typedef u64 v2 __attribute__((ext_vector_type(2)));
                const v2 q0 = __builtin_nontemporal_load((const v2 *)(recs + 4 * i));
                const v2 q1 = __builtin_nontemporal_load((const v2 *)(recs + 4 * i + 2));
                h[r] = q0.x;
                val[r][0] = q0.y;
                if (NV > 1) val[r][NV > 1 ? 1 : 0] = q1.x;
                if (NV > 2) val[r][NV > 2 ? 2 : 0] = q1.y;
            } else {
                h[r] = __builtin_nontemporal_load(&recs[i]);
#pragma unroll
                for (int j = 0; j < NV; j++) val[r][j] = __builtin_nontemporal_load(&recs[(size_t)(1 + j) * cap + i]);
            }
        }
#pragma unroll
        for (int r = 0; r < RU; r++) {
            if (!in[r] || (unsigned)h[r] == 0xffffffffu) continue; // padding of the write-combining scatter
            const u64 slot = h[r] & 0xffffffffULL;
            const u64 row = row0 + (h[r] >> 32);
            if (row < smem[slot]) atomicMin((unsigned long long *)&smem[slot], (unsigned long long)row);
#pragma unroll
            for (int a = 0; a < RFX_MAX_AGGS; a++) {
                if (kind[a] < 0) continue;
                u64 x = 0;
#pragma unroll
                for (int j = 0; j < NV; j++)
                    if (plane[a] == j) x = val[r][j];
                group_apply(&smem[(i64)arr_of[a] * local + slot], &smem[(i64)(arr_of[a] + 1) * local + slot], kind[a], f64[a], x, skip[a]);
            }
        }
    }
    __syncthreads();
    // merge into the global tables (several workgroups may share a partition: atomics)
    const i64 gbase = (i64)p << A.lb;
    for (i64 i = tid; i < local; i += THREADS) {
        const u64 f = smem[i];
        if (f == (u64)RFX_INF_I64_D) continue;
        const i64 g = A.lowbit ? ((i << 8) | (i64)(((u64)p - (u64)A.kmin) & 255ULL)) : (gbase + i);
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

template <int NC>
static void launch_hist(rfx_ctx *c, const Plan &Ph, const PartArgs &Ah, int nwg) {
    // the predicate-free instantiations are the C3 shape: without the generic predicate code the kernels are ~10x smaller
    if (Ph.npred == 0) hipLaunchKernelGGL((k_part_hist<NC, 0>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, Ph, Ah);
    else hipLaunchKernelGGL((k_part_hist<NC, RFX_MAX_PREDS>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, Ph, Ah);
}

template <int NC, int NV>
static void launch_scatter(rfx_ctx *c, const Plan &P, const PartArgs &A, int nwg) {
    if (P.npred == 0) hipLaunchKernelGGL((k_part_scatter<NC, NV, 0>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
    else hipLaunchKernelGGL((k_part_scatter<NC, NV, RFX_MAX_PREDS>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
}

template <int NC, int NV>
static void launch_scatter_dwc(rfx_ctx *c, const Plan &P, const PartArgs &A, int nwg) {
    if (P.npred == 0) hipLaunchKernelGGL((k_part_scatter_dwc<NC, NV, 0>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
    else hipLaunchKernelGGL((k_part_scatter_dwc<NC, NV, RFX_MAX_PREDS>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
}

template <int NC>
static int launch_part(rfx_ctx *c, const Plan &P, const PartArgs &A, int nwg) {
    // pass 0 reads only the key and the predicate columns: a reduced plan without the value columns
    Plan Ph = P;
    PartArgs Ah = A;
    int map[RFX_MAX_COLS], nh = 0;
    for (int i = 0; i < RFX_MAX_COLS; i++) map[i] = -1;
    auto use = [&](int col) {
        if (col >= 0 && map[col] < 0) {
            map[col] = nh;
            Ph.cols[nh++] = P.cols[col];
        }
    };
    use(A.key_idx);
    for (int i = 0; i < P.npred; i++) {
        use(P.preds[i].col);
        use(P.preds[i].rhs_col);
    }
    for (int i = 0; i < P.npred; i++) {
        Ph.preds[i].col = map[P.preds[i].col];
        if (P.preds[i].rhs_col >= 0) Ph.preds[i].rhs_col = map[P.preds[i].rhs_col];
    }
    Ph.ncols = nh;
    Ph.nagg = 0;
    Ah.key_idx = map[A.key_idx];
    if (!A.lowbit) {
        switch (nh) {
            case 1: launch_hist<1>(c, Ph, Ah, nwg); break;
            case 2: launch_hist<2>(c, Ph, Ah, nwg); break;
            case 3: launch_hist<3>(c, Ph, Ah, nwg); break;
            default: launch_hist<4>(c, Ph, Ah, nwg); break;
        }
    }
    hipLaunchKernelGGL(k_part_colscan, dim3((A.nparts + COLSCAN_P - 1) / COLSCAN_P), dim3(RFX_BLOCK), 0, c->stream, A, nwg);
    hipLaunchKernelGGL(k_part_startscan, dim3(1), dim3(PART_MAX), 0, c->stream, A);
    switch (A.nv) {
        case 0: launch_scatter<NC, 0>(c, P, A, nwg); break;
        case 1:
            if (A.wc && !(c->flags & RFX_TUNE_DIRECT_WC)) { // in-process A/B at one value plane: tile-sorted 11.9 ms vs direct 13.3 ms per C3 query
                if (P.npred == 0) hipLaunchKernelGGL((k_part_scatter_wc<NC, 0>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
                else hipLaunchKernelGGL((k_part_scatter_wc<NC, RFX_MAX_PREDS>), dim3(nwg), dim3(RFX_BLOCK), 0, c->stream, P, A);
            } else if (A.wc) launch_scatter_dwc<NC, 1>(c, P, A, nwg);
            else launch_scatter<NC, 1>(c, P, A, nwg);
            break;
        case 2:
            if (A.soa) launch_scatter_soa<NC, 2>(c, P, A, nwg);
            else if (A.wc) launch_scatter_dwc<NC, 2>(c, P, A, nwg);
            else launch_scatter<NC, 2>(c, P, A, nwg);
            break;
        default:
            if (A.soa) launch_scatter_soa<NC, 3>(c, P, A, nwg);
            else if (A.wc) launch_scatter_dwc<NC, 3>(c, P, A, nwg);
            else launch_scatter<NC, 3>(c, P, A, nwg);
            break;
    }
    return RFX_OK;
}

// ---- sparse keys: per-partition LDS hash tables in front of the device-wide one (K9, partitioned form) ----
// range > rows: the reference switches to open addressing (index_group_i64_unscoped, core/index.c:1959-1977).  One
// device-wide CAS table takes ~21 G rows/s whatever its size (memory-side atomics, section 3 of DESIGN.md): 48 ms per 1e9
// rows.  Same cure as for the dense path: partition -- here by the TOP 8 bits of hash_index_u64(key), records carry the key
// as value plane 0 -- then one workgroup per partition aggregates its records in an LDS open-addressed table
// {key, first, acc..} (ds_cmpst_rtn_b64 insert, ds atomics) and merges every occupied entry into the device-wide table
// once: the global atomics drop from one per row to one per (partition, distinct key).  A key that finds no free LDS entry
// within PH_PROBES steps (more distinct keys in the partition than the table holds) is applied to the device-wide table
// directly; the null key (its bit pattern is the empty marker) always is.
#define PH_THREADS 1024
#define PH_PROBES 48
struct PartHashArgs {
    HashArgs H;       // the device-wide table
    unsigned lcap;    // LDS table entries
    int narr;         // 8-byte arrays per entry besides key and first (acc + cnt per aggregate)
    int *overflow;    // device-wide table full
};
template <int NVT>
__global__ __launch_bounds__(PH_THREADS) void k_part_hash_aggregate(const Plan P, const PartArgs A, const PartHashArgs X) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const unsigned C = X.lcap;
    u64 *lkey = smem;                              // [C]
    u64 *larr = smem + C;                          // [narr][C]
    unsigned *lfirst = (unsigned *)(smem + (size_t)(1 + X.narr) * C); // [C] local row of the first occurrence
    int kind[RFX_MAX_AGGS], f64[RFX_MAX_AGGS], plane[RFX_MAX_AGGS], arr_of[RFX_MAX_AGGS], skip[RFX_MAX_AGGS];
    {
        int arr = 0;
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            kind[a] = (a < P.nagg) ? P.aggs[a].kind : -1;
            f64[a] = (a < P.nagg) ? P.aggs[a].f64 : 0;
            skip[a] = (a < P.nagg) ? P.aggs[a].skipnull : 0;
            plane[a] = (a < P.nagg) ? A.agg_plane[a] : -1;
            arr_of[a] = arr;
            if (kind[a] >= 0) arr += agg_has_cnt(kind[a], f64[a]) ? 2 : 1;
        }
    }
    for (unsigned i = tid; i < C; i += PH_THREADS) {
        lkey[i] = (u64)RFX_NULL_I64_D;
        lfirst[i] = 0xffffffffu;
    }
#pragma unroll
    for (int a = 0; a < RFX_MAX_AGGS; a++) {
        if (kind[a] < 0) continue;
        const u64 id = acc_identity(kind[a], f64[a]);
        for (unsigned i = tid; i < C; i += PH_THREADS) larr[(size_t)arr_of[a] * C + i] = id;
        if (agg_has_cnt(kind[a], f64[a]))
            for (unsigned i = tid; i < C; i += PH_THREADS) larr[(size_t)(arr_of[a] + 1) * C + i] = 0;
    }
    __syncthreads();
    const u64 beg = A.part_start[p], end = A.part_start[p + 1];
    const u64 row0 = (u64)P.row0;
    const u64 *__restrict__ recs = A.recs;
    constexpr int RSU = (NVT == 1) ? 2 : 4;
    const size_t cap = (size_t)A.cap;
    constexpr int RU = 4; // records in flight per lane (8: 15.5 ms against 10.4 -- registers): the loads of a batch are issued before any of them is hashed
    for (u64 i0 = beg; i0 < end; i0 += (u64)PH_THREADS * RU) {
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        u64 hs[RU], keys[RU], vals[RU][2];
        bool live[RU];
#pragma unroll
        for (int r = 0; r < RU; r++) {
            const u64 i = i0 + (u64)r * PH_THREADS + tid;
            live[r] = i < end;
            hs[r] = WC_SENTINEL;
            keys[r] = 0;
            vals[r][0] = vals[r][1] = 0;
            if (!live[r]) continue;
            if (A.soa) { // planes: headers, keys, values
                hs[r] = __builtin_nontemporal_load(&recs[i]);
                keys[r] = __builtin_nontemporal_load(&recs[cap + i]);
                if (NVT > 1) vals[r][0] = __builtin_nontemporal_load(&recs[2 * cap + i]);
                if (NVT > 2) vals[r][1] = __builtin_nontemporal_load(&recs[3 * cap + i]);
            } else {
                const v2 q0 = __builtin_nontemporal_load((const v2 *)(recs + RSU * i));
                hs[r] = q0.x;
                keys[r] = q0.y;
                if (NVT > 1) {
                    const v2 q1 = __builtin_nontemporal_load((const v2 *)(recs + RSU * i + 2));
                    vals[r][0] = q1.x;
                    vals[r][1] = q1.y;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RU; r++) {
        const u64 h = hs[r], key = keys[r];
        const u64 val[2] = {vals[r][0], vals[r][1]};
        if (!live[r] || h == WC_SENTINEL) continue; // beyond the partition / padding of the write-combining scatter
        const unsigned lrow = (unsigned)(h >> 32);
        // find-or-insert in the LDS table
        int idx = -1;
        if ((i64)key != RFX_NULL_I64_D) {
            const u64 hh = rfx_hash_index_u64(RFX_U64_HASH_SEED, key);
            unsigned s = (unsigned)((((hh >> 16) & 0xffffffffULL) * (u64)C) >> 32); // bits 16..47: independent of the partition bits
            for (int probe = 0; probe < PH_PROBES; probe++) {
                const u64 k = lkey[s];
                if (k == key) { idx = (int)s; break; }
                if ((i64)k == RFX_NULL_I64_D) {
                    const u64 old = atomicCAS((unsigned long long *)&lkey[s], (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key);
                    if ((i64)old == RFX_NULL_I64_D || old == key) { idx = (int)s; break; }
                }
                s = (s + 1 == C) ? 0 : s + 1;
            }
        }
        if (idx >= 0) {
            atomicMin(&lfirst[idx], lrow);
#pragma unroll
            for (int a = 0; a < RFX_MAX_AGGS; a++) {
                if (kind[a] < 0) continue;
                const u64 x = (plane[a] == 1) ? val[0] : ((plane[a] == 2) ? val[1] : 0ULL);
                group_apply(&larr[(size_t)arr_of[a] * C + idx], &larr[(size_t)(arr_of[a] + 1) * C + idx], kind[a], f64[a], x, skip[a]);
            }
        } else {
            // no room in LDS for this key (or the null key): straight to the device-wide table
            const i64 g = hash_slot(X.H.keys, X.H.capacity, key);
            if (g < 0) {
                atomicExch(X.overflow, 1);
                continue;
            }
            const u64 row = row0 + lrow;
            if (row < X.H.first[g]) atomicMin((unsigned long long *)&X.H.first[g], (unsigned long long)row);
#pragma unroll
            for (int a = 0; a < RFX_MAX_AGGS; a++) {
                if (kind[a] < 0) continue;
                const u64 x = (plane[a] == 1) ? val[0] : ((plane[a] == 2) ? val[1] : 0ULL);
                group_apply(&X.H.acc[a][g], X.H.cnt[a] ? &X.H.cnt[a][g] : (u64 *)0, kind[a], f64[a], x, skip[a]);
            }
        }
        }
    }
    __syncthreads();
    // merge the partition's groups into the device-wide table: one insert per distinct key
    for (unsigned i = tid; i < C; i += PH_THREADS) {
        const u64 key = lkey[i];
        if ((i64)key == RFX_NULL_I64_D) continue;
        const i64 g = hash_slot(X.H.keys, X.H.capacity, key);
        if (g < 0) {
            atomicExch(X.overflow, 1);
            continue;
        }
        const u64 row = row0 + lfirst[i];
        if (row < X.H.first[g]) atomicMin((unsigned long long *)&X.H.first[g], (unsigned long long)row);
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            if (kind[a] < 0) continue;
            const bool hc = agg_has_cnt(kind[a], f64[a]);
            group_merge_cell(&X.H.acc[a][g], hc ? &X.H.cnt[a][g] : (u64 *)0, kind[a], f64[a], larr[(size_t)arr_of[a] * C + i],
                             hc ? larr[(size_t)(arr_of[a] + 1) * C + i] : 0ULL);
        }
    }
}

// ---- selective filters: compact first, partition what is left ----
// The scatter pays its LDS phases per 2048-row tile whether 2048 or 200 rows survive the predicates (C3 under a 10 %
// filter: 11.8 ms in the scatter alone, slower than unfiltered; storing the survivors directly as 16-byte records is
// slower still -- 15 ms -- because every store is a partial cache line).  So when at most half the rows pass:
//   1. predicates -> selection bitmap + per-chunk counts (`where` pass A, reads only the predicate columns), scan;
//   2. ordered compaction of the key, the value planes and the local row ids by that bitmap (coalesced 8-byte stores);
//   3. the unfiltered partitioned pipeline over the compacted columns -- same tables, `first` into a scratch array;
//   4. first[slot] = min(first[slot], row0 + rows32[scratch_first[slot]]): compaction is order preserving, so the smallest
//      compacted position is the first occurrence (the reference does the same through its filter ids, core/query.c:65-72).
int rfx_where_bitmap_of_plan(rfx_ctx *c, const Plan &Pfull, i64 *count);
int rfx_where_counts_of_bitmap(rfx_ctx *c, i64 nrows, i64 *count);
int rfx_where_compact_cols(rfx_ctx *c, i64 nrows, const u64 *const *src, u64 *const *dst, int ncol, unsigned *d_rows32);

__global__ __launch_bounds__(RFX_BLOCK) void k_first_translate(const u64 *__restrict__ tmp, i64 range, const unsigned *__restrict__ rows32, i64 row0,
                                                             u64 *__restrict__ first) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < range; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 f = tmp[i];
        if (f == (u64)RFX_INF_I64_D) continue;
        const u64 row = (u64)(row0 + (i64)rows32[f]);
        if (row < first[i]) atomicMin((unsigned long long *)&first[i], (unsigned long long)row);
    }
}

int rfx_group_part_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t);
int rfx_chunk_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t); // rfx_group_chunk.hip

// RFX_OK: done.  RFX_ESTATE: the partitioned path does not apply (caller uses device atomics).  1: not selective after
// all, continue with the filtered write-combining scatter.
static int part_accumulate_selective(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t, const PartArgs &A, bool have_bitmap) {
    i64 nsel = 0;
    // the fused scope pass has just left this very selection as a bitmap: count and scan it; else evaluate the predicates
    int rc = have_bitmap ? rfx_where_counts_of_bitmap(c, P.nrows, &nsel) : rfx_where_bitmap_of_plan(c, P, &nsel);
    if (rc != RFX_OK) return rc;
    if (nsel * 2 > P.nrows) return 1;
    if (nsel < (1 << 16)) return RFX_ESTATE;
    const int ncol = 1 + A.nv;
    const size_t col_bytes = (((size_t)nsel * 8) + 255) & ~(size_t)255;
    const size_t rows_bytes = (((size_t)nsel * 4) + 255) & ~(size_t)255;
    const size_t first_bytes = (size_t)t->range * 8;
    rc = rfx_sel_reserve(c, col_bytes * ncol + rows_bytes + first_bytes);
    if (rc != RFX_OK) return rc;
    char *w = (char *)c->d_sel;
    const u64 *src[4];
    u64 *dst[4];
    src[0] = P.cols[key_idx];
    for (int j = 0; j < A.nv; j++) src[1 + j] = P.cols[A.vcol[j]];
    for (int j = 0; j < ncol; j++) {
        dst[j] = (u64 *)w;
        w += col_bytes;
    }
    unsigned *rows32 = (unsigned *)w;
    w += rows_bytes;
    u64 *tmp_first = (u64 *)w;
    rc = rfx_where_compact_cols(c, P.nrows, src, dst, ncol, rows32);
    if (rc != RFX_OK) return rc;
    Plan P2;
    memset(&P2, 0, sizeof(P2));
    P2.ncols = ncol;
    for (int j = 0; j < ncol; j++) P2.cols[j] = dst[j];
    P2.npred = 0;
    P2.logic = RFX_AND;
    P2.nagg = P.nagg;
    for (int a = 0; a < RFX_MAX_AGGS; a++) {
        P2.aggs[a] = P.aggs[a];
        if (a < P.nagg) P2.aggs[a].col = (A.agg_plane[a] >= 0) ? 1 + A.agg_plane[a] : -1;
    }
    P2.nrows = nsel;
    P2.row0 = 0;
    rfx_group_tables_t t2 = *t;
    t2.d_first = (int64_t *)tmp_first;
    rc = rfx_fill_u64(c, tmp_first, t->range, (u64)RFX_INF_I64_D);
    if (rc != RFX_OK) return rc;
    rc = rfx_group_part_accumulate(c, P2, 0, &t2);
    if (rc != RFX_OK) return rc; // RFX_ESTATE included: nothing has touched the real tables yet
    int grid = (int)((t->range + RFX_BLOCK - 1) / RFX_BLOCK);
    if (grid > rfx_grid(c) * 4) grid = rfx_grid(c) * 4;
    hipLaunchKernelGGL(k_first_translate, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)tmp_first, (i64)t->range, (const unsigned *)rows32, P.row0,
                       (u64 *)t->d_first);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// Returns RFX_ESTATE when this path does not apply (caller falls back to device-scope atomics).
int rfx_group_part_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t) {
    if (P.nrows >= (1LL << 32) || P.nrows < (1 << 16)) return RFX_ESTATE; // 32-bit local rows; tiny inputs are not worth 4 passes
    if (P.nx > 0) {
        // expression aggregates: the records carry plain values, so evaluate the expressions into scratch columns first
        // (what the reference does for every query, core/math.c binop_map) and partition those
        if (P.ncols + P.nx > RFX_MAX_COLS) return RFX_ESTATE;
        Plan Pm = P;
        const int rc = rfx_plan_materialise_exprs(c, &Pm);
        if (rc != RFX_OK) return rc;
        return rfx_group_part_accumulate(c, Pm, key_idx, t);
    }
    {
        // did rfx_hip_group_scope leave exactly this plan's rows partitioned already?  Then only pass 2 is left.
        const int crc = rfx_chunk_accumulate(c, P, key_idx, t);
        if (crc != RFX_ESTATE) {
            c->pc_valid = 0;
            return crc;
        }
    }
    PartArgs A;
    memset(&A, 0, sizeof(A));
    int narr = 1;
    A.nv = 0;
    for (int a = 0; a < P.nagg; a++) {
        const PlanAgg ag = P.aggs[a];
        narr += 1 + (agg_has_cnt(ag.kind, ag.f64) ? 1 : 0);
        A.agg_plane[a] = -1;
        if (ag.kind == RFX_AGG_COUNT || ag.kind == RFX_AGG_FIRST || ag.col < 0) continue;
        int j = 0;
        for (; j < A.nv; j++)
            if (A.vcol[j] == ag.col) break;
        if (j == A.nv) {
            if (A.nv >= 3) return RFX_ESTATE; // records carry at most 3 value planes
            A.vcol[A.nv++] = ag.col;
        }
        A.agg_plane[a] = j;
    }
    if (P.npred > 0 && A.nv >= 1 && !(c->flags & RFX_TUNE_NO_SEL_COMPACT)) {
        // rows that passed the predicates, if the fused scope pass has just counted exactly this selection
        bool known = c->pc_valid && c->pc_key == (const void *)P.cols[key_idx] && c->pc_nrows == P.nrows && c->pc_npred == P.npred && c->pc_logic == P.logic;
        if (known) {
            u64 sig[RFX_MAX_PREDS][6];
            plan_pred_sig(P, sig);
            known = memcmp(sig, c->pc_sig, sizeof(u64) * 6 * (size_t)P.npred) == 0;
        }
        if (!known || c->pc_seen * 2 <= P.nrows) {
            const int rc2 = part_accumulate_selective(c, P, key_idx, t, A, known && c->pc_bitmap);
            if (rc2 != 1) {
                c->pc_valid = 0;
                return rc2;
            }
        }
    }
    const int nwg = part_nwg(c);
    // Did rfx_hip_scope_i64 just leave the low-bit histogram of exactly these rows?  Then pass 0 is already done.
    int lowbit = 0, lb = 0;
    if (c->pc_valid && c->pc_key == (const void *)P.cols[key_idx] && c->pc_nrows == P.nrows && c->pc_npred == P.npred && c->pc_logic == P.logic &&
        c->pc_nwg == nwg && t->range > 256) {
        u64 sig[RFX_MAX_PREDS][6];
        plan_pred_sig(P, sig);
        lowbit = (P.npred == 0) || memcmp(sig, c->pc_sig, sizeof(u64) * 6 * (size_t)P.npred) == 0;
        if (lowbit) {
            const i64 per = (t->range + 255) >> 8; // slots per partition: those congruent to one residue mod 256
            while ((1LL << lb) < per) lb++;
            if ((1LL << lb) * narr * 8 > PART_LDS_BIG_BYTES) lowbit = 0;
        }
    }
    c->pc_valid = 0; // consumed (or stale) either way
    i64 nparts;
    if (lowbit) nparts = 256;
    else {
        // slots per partition: largest power of two whose tables fit the LDS budget -- the small budget if that keeps the
        // partition count within the write-combining scatter's reach, else the big one
        size_t budget = PART_LDS_BYTES;
        for (int attempt = 0; attempt < 2; attempt++) {
            lb = 0;
            while ((size_t)(1LL << (lb + 1)) * narr * 8 <= budget) lb++;
            nparts = (t->range + (1LL << lb) - 1) >> lb;
            if (nparts <= WC_MAXP || (c->flags & RFX_TUNE_NO_BIG_LDS)) break;
            budget = PART_LDS_BIG_BYTES;
        }
        if (lb < 8) return RFX_ESTATE;
        if (nparts > PART_MAX || nparts < 2) return RFX_ESTATE;
    }
    A.lowbit = lowbit;
    A.kmin = t->kmin;
    A.range = t->range;
    A.lb = lb;
    A.nparts = (int)nparts;
    A.key_idx = key_idx;
    A.narr = narr;
    // enough workgroups in pass 2 to fill the chip (two 512-thread workgroups per CU)
    A.split = (int)((2 * c->num_cus + nparts - 1) / nparts);
    if (A.split < 1) A.split = 1;
    const bool wc_ok = A.nv >= 1 && A.nv <= 3 && nparts <= WC_MAXP && !(c->flags & RFX_TUNE_NO_WRITE_COMBINE);
    // in-process A/B: two planes 35.6 -> 31.7 ms (K9), three planes 29 -> 38 ms (90 KB of LDS: one workgroup per CU) -- so two only
    A.soa = wc_ok && A.nv == 2 && !(c->flags & RFX_TUNE_NO_SOA_WC); // one plane as two 8-byte planes: scatter 10.8 ms against 7.4 ms for 16-byte records
    A.wc = wc_ok ? ((A.nv == 1 || A.soa) ? 8 : 4) : 0;
    const int rsu = (A.wc && !A.soa) ? ((A.nv == 1) ? 2 : 4) : (1 + A.nv); // u64 per record: AoS when write-combining 16/32-byte records, else planes
    A.cap = ((P.nrows + 63) / 64) * 64 + (A.wc ? (i64)nwg * nparts * A.wc : 0);
    const size_t off_bytes = (size_t)nwg * nparts * 8;
    const size_t start_bytes = (size_t)(nparts + 2) * 8;
    const size_t rec_bytes = (size_t)rsu * A.cap * 8;
    const size_t need = ((off_bytes + 255) & ~(size_t)255) + ((start_bytes + 255) & ~(size_t)255) + rec_bytes;
    int rc = rfx_part_reserve(c, need);
    if (rc != RFX_OK) return rc;
    char *w = (char *)c->d_part;
    A.offsets = (u64 *)w;
    w += (off_bytes + 255) & ~(size_t)255;
    A.part_start = (u64 *)w;
    w += (start_bytes + 255) & ~(size_t)255;
    A.recs = (u64 *)w;
    if (lowbit) A.offsets = c->d_pc_counts; // [nwg][256] counts from the fused scope pass, scanned in place below
    A.first = (u64 *)t->d_first;
    for (int a = 0; a < t->nagg; a++) {
        A.acc[a] = (u64 *)t->d_acc[a];
        A.cnt[a] = (u64 *)t->d_cnt[a];
    }
    RFX_KERNEL_BEGIN(c);
    switch (P.ncols) {
        case 1: launch_part<1>(c, P, A, nwg); break;
        case 2: launch_part<2>(c, P, A, nwg); break;
        case 3: launch_part<3>(c, P, A, nwg); break;
        case 4: launch_part<4>(c, P, A, nwg); break;
        default: return RFX_ESTATE;
    }
    const size_t lds = (size_t)narr * (1ULL << lb) * 8;
    if (lds > PART_LDS_BYTES) {
        // one 1024-thread workgroup per partition (and per CU): dynamic LDS above 64 KB is opted into once per instance
        static unsigned long long attr_set[4] = {0};
#define RFX_PA(N)                                                                                                                                  \
    case N:                                                                                                                                        \
        if (!((attr_set[N] >> (c->device & 63)) & 1ull)) {                                                                                                                        \
            RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_part_aggregate<N, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
            __atomic_fetch_or(&attr_set[N], 1ull << (c->device & 63), __ATOMIC_RELAXED);                                                                                                                    \
        }                                                                                                                                          \
        hipLaunchKernelGGL((k_part_aggregate<N, 1024>), dim3((int)nparts), dim3(1024), lds, c->stream, P, A);                                     \
        break
        A.split = 1;
        switch (A.nv) {
            RFX_PA(0);
            RFX_PA(1);
            RFX_PA(2);
            default: RFX_PA(3);
        }
#undef RFX_PA
    } else {
        const int grid2 = (int)(nparts * A.split);
        switch (A.nv) {
            case 0: hipLaunchKernelGGL((k_part_aggregate<0, PART_AGG_THREADS>), dim3(grid2), dim3(PART_AGG_THREADS), lds, c->stream, P, A); break;
            case 1: hipLaunchKernelGGL((k_part_aggregate<1, PART_AGG_THREADS>), dim3(grid2), dim3(PART_AGG_THREADS), lds, c->stream, P, A); break;
            case 2: hipLaunchKernelGGL((k_part_aggregate<2, PART_AGG_THREADS>), dim3(grid2), dim3(PART_AGG_THREADS), lds, c->stream, P, A); break;
            default: hipLaunchKernelGGL((k_part_aggregate<3, PART_AGG_THREADS>), dim3(grid2), dim3(PART_AGG_THREADS), lds, c->stream, P, A); break;
        }
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// Sparse-key form.  RFX_ESTATE: not applicable (tiny input, too many value planes) -- the caller runs the direct kernel.
// How many distinct keys?  2^15 strided sample keys go into a 2^18-slot scratch table; with s samples of D equally likely keys
// about s^2 / 2D of them find their key already there (birthday), so D ~ s^2 / (2 x duplicates): 1e6 keys -> ~540 duplicates,
// 1e8 -> ~5.  Clustered or filtered inputs make the estimate too HIGH, which only sends the query to the device-wide table.
#define DSAMP_SLOTS (1 << 18)
__global__ __launch_bounds__(RFX_BLOCK) void k_distinct_sample(const u64 *__restrict__ keys, i64 stride, i64 nsamp, u64 *__restrict__ table,
                                                              unsigned *__restrict__ dups) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < nsamp; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 k = keys[i * stride];
        if ((i64)k == RFX_NULL_I64_D) continue;
        u64 s = rfx_hash_index_u64(RFX_U64_HASH_SEED, k) & (DSAMP_SLOTS - 1);
        for (int probe = 0; probe < 64; probe++) {
            const u64 old = atomicCAS((unsigned long long *)&table[s], (unsigned long long)RFX_NULL_I64_D, (unsigned long long)k);
            if ((i64)old == RFX_NULL_I64_D) break;
            if (old == k) { atomicAdd(dups, 1u); break; }
            s = (s + 1) & (DSAMP_SLOTS - 1);
        }
    }
}
int rfx_estimate_distinct(rfx_ctx *c, const u64 *d_key, i64 nrows, double *est) { // (also sizes the plane form of this path, rfx_group_plane.hip)
    const i64 nsamp = nrows < (1 << 15) ? nrows : (1 << 15);
    int rc = rfx_ws_reserve(c, (size_t)DSAMP_SLOTS * 8 + 512);
    if (rc != RFX_OK) return rc;
    u64 *table = (u64 *)((char *)c->d_ws + 512); // the first bytes of the workspace hold the caller's overflow flag
    unsigned *dups = (unsigned *)((char *)c->d_ws + 256);
    if ((rc = rfx_fill_u64(c, table, DSAMP_SLOTS, (u64)RFX_NULL_I64_D)) != RFX_OK) return rc;
    RFX_HIP_CHECK(hipMemsetAsync(dups, 0, 4, c->stream));
    hipLaunchKernelGGL(k_distinct_sample, dim3(32), dim3(RFX_BLOCK), 0, c->stream, d_key, nrows / nsamp, nsamp, table, dups);
    RFX_HIP_CHECK(hipGetLastError());
    unsigned *h = (unsigned *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, dups, 4, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    *est = (double)nsamp * (double)nsamp / (2.0 * ((double)h[0] + 0.5));
    // (the birthday figure cannot fall below nsamp / 2; when most samples WERE duplicates the sample has seen nearly every key, and what it
    // saw is the estimate: D (1 - e^(-s/D)) distinct values among s samples)
    if ((i64)h[0] * 2 > nsamp) *est = (double)(nsamp - (i64)h[0]) * 1.3;
    return RFX_OK;
}

int rfx_group_part_hash_accumulate(rfx_ctx *c, const Plan &P0, int key_idx, const HashArgs &H, int *d_overflow) {
    c->ext_i[2] = 0;
    if (P0.nrows >= (1LL << 32) || P0.nrows < (1 << 16) || (c->flags & RFX_TUNE_NO_PARTITION)) return RFX_ESTATE;
    {
        // 256 partitions x one CU's LDS hold a few thousand keys each: beyond ~4 M distinct keys nearly every record overflows its
        // partition's table into the device-wide one (1e8 keys: 820 ms against 30) -- those go to the device-wide table directly
        double est = 0;
        const int rc = rfx_estimate_distinct(c, (const u64 *)P0.cols[key_idx], P0.nrows, &est);
        if (rc != RFX_OK) return rc;
        c->ext_i[2] = (i64)est; // (the device-wide kernel's caller sizes its table by it)
        if (est > 4.0e6) return RFX_ESTATE;
    }
    Plan P = P0;
    if (P.nx > 0) { // expression aggregates: the records carry plain values
        if (P.ncols + P.nx > RFX_MAX_COLS) return RFX_ESTATE;
        const int rc = rfx_plan_materialise_exprs(c, &P);
        if (rc != RFX_OK) return rc;
    }
    PartArgs A;
    memset(&A, 0, sizeof(A));
    A.hashed = 1;
    A.nv = 1;
    A.vcol[0] = key_idx; // plane 0 of every record is the key itself
    int narr = 0;
    for (int a = 0; a < P.nagg; a++) {
        const PlanAgg ag = P.aggs[a];
        narr += 1 + (agg_has_cnt(ag.kind, ag.f64) ? 1 : 0);
        A.agg_plane[a] = -1;
        if (ag.kind == RFX_AGG_COUNT || ag.kind == RFX_AGG_FIRST || ag.col < 0) continue;
        int j = 1;
        for (; j < A.nv; j++)
            if (A.vcol[j] == ag.col) break;
        if (j == A.nv) {
            if (A.nv >= 3) return RFX_ESTATE; // records carry the key and at most 2 value planes
            A.vcol[A.nv++] = ag.col;
        }
        A.agg_plane[a] = j;
    }
    const int nwg = part_nwg(c);
    A.kmin = 0;
    A.range = 0;
    A.lb = 0;
    A.nparts = 256;
    A.key_idx = key_idx;
    A.narr = narr;
    A.split = 1;
    A.soa = A.nv == 2 && !(c->flags & RFX_TUNE_NO_SOA_WC);
    A.wc = (A.nv == 1 || A.soa) ? 8 : 4;
    const int rsu = A.soa ? (1 + A.nv) : ((A.nv == 1) ? 2 : 4);
    A.cap = ((P.nrows + 63) / 64) * 64 + (i64)nwg * A.nparts * A.wc;
    const size_t off_bytes = (size_t)nwg * A.nparts * 8;
    const size_t start_bytes = (size_t)(A.nparts + 2) * 8;
    const size_t rec_bytes = (size_t)rsu * A.cap * 8;
    const size_t need = ((off_bytes + 255) & ~(size_t)255) + ((start_bytes + 255) & ~(size_t)255) + rec_bytes;
    int rc = rfx_part_reserve(c, need);
    if (rc != RFX_OK) return rc;
    char *w = (char *)c->d_part;
    A.offsets = (u64 *)w;
    w += (off_bytes + 255) & ~(size_t)255;
    A.part_start = (u64 *)w;
    w += (start_bytes + 255) & ~(size_t)255;
    A.recs = (u64 *)w;
    c->pc_valid = 0;
    RFX_KERNEL_BEGIN(c);
    switch (P.ncols) {
        case 1: launch_part<1>(c, P, A, nwg); break;
        case 2: launch_part<2>(c, P, A, nwg); break;
        case 3: launch_part<3>(c, P, A, nwg); break;
        case 4: launch_part<4>(c, P, A, nwg); break;
        default: return RFX_ESTATE;
    }
    PartHashArgs X;
    X.H = H;
    X.narr = narr;
    X.overflow = d_overflow;
    // LDS entry = key 8 B + narr x 8 B + first 4 B; take what fits 150 KB (one 1024-thread workgroup per CU)
    const size_t entry = 8 + (size_t)narr * 8 + 4;
    X.lcap = (unsigned)(((size_t)150 * 1024) / entry) & ~63u;
    const size_t lds = (((size_t)X.lcap * (8 + (size_t)narr * 8)) + (size_t)X.lcap * 4 + 15) & ~(size_t)15;
    static unsigned long long attr_set[3] = {0};
#define RFX_PH(N)                                                                                                                              \
    case N:                                                                                                                                    \
        if (!((attr_set[N - 1] >> (c->device & 63)) & 1ull)) {                                                                                                                \
            RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_part_hash_aggregate<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            __atomic_fetch_or(&attr_set[N - 1], 1ull << (c->device & 63), __ATOMIC_RELAXED);                                                                                                            \
        }                                                                                                                                      \
        hipLaunchKernelGGL((k_part_hash_aggregate<N>), dim3(A.nparts), dim3(PH_THREADS), lds, c->stream, P, A, X);                             \
        break
    switch (A.nv) {
        RFX_PH(1);
        RFX_PH(2);
        default: RFX_PH(3);
    }
#undef RFX_PH
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
