// rfx_part_common.hpp -- internal: pieces shared by the radix-partitioned group-by (rfx_group_part.hip: exact-offset
// partitioning behind a histogram pass) and its one-pass form (rfx_group_chunk.hip: chunk-allocated partitions).
#pragma once
#include "rfx_group_common.hpp"

#define PART_TILE_ROWS 2048 /* rows per workgroup per tile: 256 lanes x 8 rows */
#define PART_MAX 1024       /* max partitions */
#define PART_LDS_BYTES (64 * 1024)      /* pass-2 tables of a partition: two 512-thread workgroups per CU */
#define PART_LDS_BIG_BYTES (144 * 1024) /* ... or one 1024-thread workgroup per CU owning (nearly) the whole LDS, when that keeps the partition count within
                                        * what the write-combining scatter handles (several aggregates over ~1e6 keys) */
#define PART_AGG_THREADS 512
#define WC_B 8 /* write-combining scatter: records per store group = 128 bytes */
#define WC_MAXP 256
#define WC_SENTINEL 0xFFFFFFFFFFFFFFFFULL

struct PartArgs {
    i64 kmin, range;
    int lb;      // log2(slots per partition)
    int nparts;  // partitions
    int key_idx;
    int nv;                    // value planes carried in the records
    int vcol[RFX_MAX_AGGS];    // plane j carries Plan::cols[vcol[j]]
    int agg_plane[RFX_MAX_AGGS]; // aggregate a reads plane agg_plane[a] (-1: none, COUNT / FIRST)
    int narr;                  // table arrays per slot (first + acc + cnt ...)
    int split;                 // workgroups per partition in pass 2
    int wc;                    // > 0: write-combining scatter, value = records per 128-byte store group (regions padded to it,
                               //      sentinel records possible); with 2-3 value planes records are 32-byte {hdr, v0, v1, v2}
    int soa;                   // 1: write-combined STRUCTURE-OF-ARRAYS records: plane 0 = headers, planes 1..nv = values, `cap` apart,
                               //    every (workgroup, partition) region padded to 8 records (k_part_scatter_soa)
    int hashed;                // 1: sparse keys -- partition = top 8 bits of hash_index_u64(key); records carry the KEY as plane 0
    int lowbit;                // 1: partition = key & 255 (known before the scope is), local slot = (key - kmin) >> 8
    u64 *offsets;              // [nwg][nparts] : counts, then exclusive offsets
    u64 *part_start;           // [nparts + 1]
    u64 *recs;                 // plane 0 = headers, planes 1..nv = values ; each plane `cap` entries
    i64 cap;
    u64 *first;
    u64 *acc[RFX_MAX_AGGS];
    u64 *cnt[RFX_MAX_AGGS];
};

// Which partition a selected row goes to, and whether it takes part at all.
__device__ __forceinline__ bool part_row_ok(const PartArgs &A, u64 slot) { return A.hashed || slot < (u64)A.range; }
__device__ __forceinline__ unsigned part_of(const PartArgs &A, u64 key, u64 slot) {
    if (A.hashed) return (unsigned)(rfx_hash_index_u64(RFX_U64_HASH_SEED, key) >> 56);
    return A.lowbit ? (unsigned)(key & 255ULL) : (unsigned)(slot >> A.lb);
}

// Shared tile front-end of pass 0 and pass 1: 8 rows per lane as four 16-byte loads per column.
template <int NC, int NP>
__device__ __forceinline__ unsigned part_load_eval(const Plan &P, const PredSet<NP> &S, i64 tile, u64 (&v)[NC][8]) {
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
    if (NP == 0) return valid;
    return eval_preds<NC, 8, NP>(S, v, valid);
}


// per-workgroup result of a scope pass fused with something else (index_scope_i64, core/index.c:376-435)
struct ScopePart {
    i64 mn, mx, sel, nulls;
};
