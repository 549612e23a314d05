// rfx_group_dict.hip -- sparse keys (K9) through a DICTIONARY pass: key -> dense id, then the dense plane-partitioned group-by on the ids.
//
// What the reference does here: index_group_i64_unscoped -> index_group_distribute (core/index.c:1959-1977, 1777-1911) -- one open-addressed
// table per CPU chunk (ht_oa_tab_next_with, core/hash.c:35-148), every row a find-or-insert, the per-chunk tables merged sequentially.  The
// group id of a key IS its table slot's payload: the hash table is a dictionary key -> group id, and AGGR_ITER then folds through the ids.
//
// Round 2's device form partitioned {row, key, value} records by hash (15 ms per 1e9 rows) and aggregated every partition in an LDS hash
// table (10 ms): 29 ms, 5.4 x the algorithmic traffic.  This file takes the reference's own decomposition instead:
//
//   k_dict_ids      ONE streaming pass over the key column: find-or-insert into a device-wide table of {key, id} entries (16 bytes, one
//                   load per probe; linear probing at load <= 0.4), ids handed out densely by one counter in order of insertion.  After the
//                   first few thousand rows every key is present and a row costs one random 16-byte read that hits the memory-side cache
//                   (the table is tens of MB); inserts are 1e6 CAS out of 1e9 rows.  Output: the id of every row as an i64 column.
//   dense group-by  rfx_hip_group_scope + rfx_hip_group_dense_accumulate on the id column over [0, ids): exactly the C3 problem -- the
//                   plane-partitioned kernels (rfx_group_plane.hip), 8-byte value + 4-byte meta planes, LDS tables per partition.
//   k_hash_merge    one insert per DISTINCT key into the caller's hashed table set (the contract of rfx_hip_group_hash_accumulate: tables
//                   keyed by slot, mergeable across GPUs), carrying first row / accumulators / counts over.
//
// Bytes per row: 8 read + 8 written (ids), then C3's 40: 56, against 86 measured for the partition-by-hash form; no value or row id is
// carried through a hash partition, and the only per-row random access is the dictionary probe.  The dictionary ignores `where:` (an
// unselected key gets an id whose group stays empty and is never merged).
// A null key (the table's empty marker, as in the reference) takes id 0 and ends up in the caller's dedicated slot `capacity`.
#include "rfx_group_common.hpp"
#include <stdlib.h>

#define DICT_EMPTY_ID 0xFFFFFFFFFFFFFFFFULL
#define DICT_MAX_PROBES 4096

struct DictArgs {
    const u64 *keys;
    i64 nrows;
    u64 *tab;      // [cap] entries {key, id}; empty key = NULL_I64, empty id = DICT_EMPTY_ID
    u64 mask;      // cap - 1
    int shift;     // 64 - log2(cap)
    unsigned idcap;
    u64 *idkeys;   // [idcap] id -> key (id 0: the null key)
    u64 *ids;      // [nrows] out
    unsigned *ctl; // [0] next id (starts at 1), [1] gave up: more distinct keys than idcap / a probe run without end
};

__device__ __forceinline__ u64 dict_hash(u64 k, int shift) {
    k *= 0x9E3779B97F4A7C15ULL;
    k ^= k >> 32;
    k *= 0xD6E8FEB86659FD93ULL;
    return k >> shift;
}

// a table entry through the caches (rfx_ld2 is the STREAMING load: non-temporal, every probe would come from HBM -- measured 28 ms per
// 1e9 probes of a 64 MB table that way)
__device__ __forceinline__ u64x2 dict_entry(const u64 *p) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    const v2 t = *(const v2 *)p;
    u64x2 r;
    r.x = t.x;
    r.y = t.y;
    return r;
}

__global__ __launch_bounds__(RFX_BLOCK) void k_dict_init(u64 *__restrict__ tab, i64 cap, u64 *__restrict__ idkeys, unsigned *__restrict__ ctl) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < cap; i += (i64)gridDim.x * RFX_BLOCK) {
        u64x2 e;
        e.x = (u64)RFX_NULL_I64_D;
        e.y = DICT_EMPTY_ID;
        *(u64x2 *)(tab + 2 * i) = e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        idkeys[0] = (u64)RFX_NULL_I64_D;
        ctl[0] = 1u;
        ctl[1] = 0u;
    }
}

#define DICT_DEAD_ID 0xFFFFFFFFFFFFFFFEULL /* the key's inserter found the id space exhausted: whoever reads this gives up as well */
#define DICT_CHUNK 32u                     /* ids a wave takes from the device-wide counter at a time */

// A wave owns 512 consecutive rows per step (lane l: rows 2l, 2l + 1 of each 128-row group): the eight first probes of a lane are issued
// together, rows whose key sits at its home slot (nearly all, at this load) are done after that one round trip.
//
// Ids come from ONE device-wide counter, and a single address takes some 1e7 returning atomics per second on this part (the same figure the
// load counter of k_group_hash met): one atomic per distinct key would cost 1e6 keys 100 ms.  So a wave draws DICT_CHUNK ids at a time into
// a wave-uniform pool {next, end} and hands them to its lanes by ballot rank; what a wave has left at the end is a hole in the id space
// (<= DICT_CHUNK per wave: cells that stay empty).  Nothing here waits for another lane: the lane whose CAS claims a slot stores the id
// in the same pass of the loop; a lane that finds the key without its id looks again in the next pass (device-scope load: another XCD's L2
// must not answer for it).
__global__ __launch_bounds__(RFX_BLOCK) void k_dict_ids(const DictArgs A) {
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nsteps = (A.nrows + 511) / 512;
    unsigned pool_next = 0, pool_end = 0; // wave-uniform
    bool dead = false;
    for (i64 q = wave_id; q < nsteps; q += nwaves) {
        const i64 base = q * 512 + lane * 2;
        const bool whole = (q + 1) * 512 <= A.nrows;
        u64 key[8], pos[8], out[8];
        unsigned um = 0; // rows still without an id
        if (whole) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u64x2 t = rfx_ld2(A.keys + base + j * 128);
                key[2 * j] = t.x;
                key[2 * j + 1] = t.y;
            }
            um = 0xffu;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const i64 row = base + (e >> 1) * 128 + (e & 1);
                const bool in = row < A.nrows;
                key[e] = in ? A.keys[row] : 0ULL;
                um |= (unsigned)in << e;
            }
        }
        u64x2 ent[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            pos[e] = dict_hash(key[e], A.shift);
            ent[e] = dict_entry(A.tab + 2 * pos[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            out[e] = 0;
            if ((i64)key[e] == RFX_NULL_I64_D) um &= ~(1u << e); // the null key: id 0
        }
        // Every pass looks at the loaded entries of ALL open rows, then issues the next loads of those still open together: one memory
        // round trip per probe step of the longest chain among the wave's 512 rows, not one per row and step.
        for (int it = 0; __any(um != 0u); it++) {
            if (it > DICT_MAX_PROBES) {
                dead = true;
                break;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const bool open = (um >> e) & 1u;
                u64 *slot = A.tab + 2 * pos[e];
                bool won = false;
                if (open) {
                    if (ent[e].x == key[e]) {
                        u64 id = ent[e].y;
                        if (id == DICT_EMPTY_ID) id = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (id == DICT_DEAD_ID) dead = true;
                        if (id != DICT_EMPTY_ID) {
                            out[e] = id;
                            um &= ~(1u << e);
                        }
                    } else if ((i64)ent[e].x == RFX_NULL_I64_D) {
                        const u64 old = atomicCAS((unsigned long long *)slot, (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key[e]);
                        won = (i64)old == RFX_NULL_I64_D;
                        if (!won && old != key[e]) pos[e] = (pos[e] + 1) & A.mask; // somebody else's key took the place
                        // (ours, inserted by somebody else a moment ago: the same place again, for the id)
                    } else {
                        pos[e] = (pos[e] + 1) & A.mask;
                    }
                }
                const u64 wb = __ballot(won);
                if (wb) { // wave-uniform: ids for the lanes that claimed a slot
                    const unsigned k = (unsigned)__popcll(wb), have = pool_end - pool_next;
                    unsigned fresh = 0;
                    if (k > have) {
                        if (lane == (int)__builtin_ctzll(wb)) fresh = atomicAdd(&A.ctl[0], (k - have) + DICT_CHUNK);
                        fresh = (unsigned)__shfl((int)fresh, (int)__builtin_ctzll(wb), 64);
                    }
                    if (won) {
                        const unsigned r = __builtin_amdgcn_mbcnt_hi((unsigned)(wb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wb, 0u));
                        const unsigned nid = r < have ? pool_next + r : fresh + (r - have);
                        if (nid >= A.idcap) {
                            __hip_atomic_store(slot + 1, DICT_DEAD_ID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            dead = true;
                        } else {
                            A.idkeys[nid] = key[e];
                            __hip_atomic_store(slot + 1, (u64)nid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            out[e] = (u64)nid;
                        }
                        um &= ~(1u << e);
                    }
                    if (k > have) {
                        pool_next = fresh + (k - have);
                        pool_end = fresh + (k - have) + DICT_CHUNK;
                    } else {
                        pool_next += k;
                    }
                }
            }
            if (__any(dead)) break;
#pragma unroll
            for (int e = 0; e < 8; e++)
                if ((um >> e) & 1u) ent[e] = dict_entry(A.tab + 2 * pos[e]);
        }
        if (__any(dead)) {
            if (lane == 0) atomicExch(&A.ctl[1], 1u);
            return;
        }
        if (whole) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u64x2 t;
                t.x = out[2 * j];
                t.y = out[2 * j + 1];
                *(u64x2 *)(A.ids + base + j * 128) = t;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const i64 row = base + (e >> 1) * 128 + (e & 1);
                if (row < A.nrows) A.ids[row] = out[e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct DictState { // ext_p[3]: the route's grow-only scratch block and its counters
    void *d;
    size_t bytes;
    i64 passes, fallbacks;
};
static DictState *dict_state(rfx_ctx *c) {
    if (!c->ext_p[3]) c->ext_p[3] = calloc(1, sizeof(DictState));
    return (DictState *)c->ext_p[3];
}
void rfx_dict_release(rfx_ctx *c) {
    DictState *st = (DictState *)c->ext_p[3];
    if (!st) return;
    if (st->d) (void)hipFree(st->d);
    free(st);
    c->ext_p[3] = NULL;
}
i64 rfx_dict_stat(rfx_ctx *c, int which) {
    DictState *st = (DictState *)c->ext_p[3];
    return st ? (which == 0 ? st->passes : st->fallbacks) : 0;
}
static int dict_reserve(rfx_ctx *c, DictState *st, size_t bytes) {
    if (st->bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (st->d) RFX_HIP_CHECK(hipFree(st->d));
    st->d = NULL;
    st->bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&st->d, bytes));
    st->bytes = bytes;
    return RFX_OK;
}
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

int rfx_estimate_distinct(rfx_ctx *c, const u64 *d_key, i64 nrows, double *est); // rfx_group_part.hip
int rfx_hash_merge_cells(rfx_ctx *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *into, const rfx_group_tables_t *from, const u64 *d_from_keys,
                         i64 from_null); // rfx_hash.hip

#define DICT_MIN_ROWS (1LL << 22)
#define DICT_MAX_IDS (3LL << 20) /* what 256 partitions of LDS tables hold with one 12-byte cell set */

// RFX_ESTATE: not applicable / gave up (the caller's tables are untouched: nothing is merged before the dense pass has finished).
// RFX_ELIMIT: the caller's table is too small for the distinct keys found (same contract as the other forms: grow and run again).
int rfx_group_dict_hash_accumulate(rfx_ctx *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs,
                                   int64_t nrows, int64_t row0, const rfx_hash_tables_t *t) {
    const char *off = getenv("RFX_NO_DICT"); // (A/B and the every-path tests; read per call)
    if (off || (c->flags & RFX_TUNE_NO_PARTITION) || nrows < ((c->flags & RFX_TUNE_CHUNK_SMALL) ? (1LL << 16) : DICT_MIN_ROWS) || nrows >= (1LL << 32) || t->nagg < 1) return RFX_ESTATE;
    DictState *st = dict_state(c);
    if (!st) return RFX_ESTATE;
    double est = 0;
    int rc = rfx_estimate_distinct(c, (const u64 *)d_key, nrows, &est);
    if (rc != RFX_OK) return rc;
    c->ext_i[2] = (i64)est;
    const int grid = c->num_cus * 4;
    // room for the estimate's error and for what the waves' id pools leave unused
    const double want = est * 1.5 + 65536.0 + (double)grid * (RFX_BLOCK / RFX_WAVE) * DICT_CHUNK;
    if (want > (double)DICT_MAX_IDS) return RFX_ESTATE;
    const unsigned idcap = (unsigned)want;
    i64 cap = 1 << 16;
    while ((double)cap < 2.5 * (double)idcap) cap <<= 1;
    int lg = 0;
    while ((1LL << lg) < cap) lg++;
    int narr = 0;
    rfx_hip_group_table_arrays(aggs, t->nagg, &narr);
    // [ctl 256][table cap x 16][idkeys idcap x 8][ids nrows x 8][dense tables narr x idcap x 8]
    const size_t o_tab = 256, o_idk = o_tab + al256((size_t)cap * 16), o_ids = o_idk + al256((size_t)idcap * 8), o_tbl = o_ids + al256((size_t)nrows * 8);
    const size_t need = o_tbl + (size_t)narr * al256((size_t)idcap * 8);
    if (dict_reserve(c, st, need) != RFX_OK) return RFX_ESTATE;
    char *w = (char *)st->d;
    DictArgs A;
    memset(&A, 0, sizeof(A));
    A.keys = (const u64 *)d_key;
    A.nrows = nrows;
    A.tab = (u64 *)(w + o_tab);
    A.mask = (u64)cap - 1;
    A.shift = 64 - lg;
    A.idcap = idcap;
    A.idkeys = (u64 *)(w + o_idk);
    A.ids = (u64 *)(w + o_ids);
    A.ctl = (unsigned *)w;
    st->passes++;
    hipLaunchKernelGGL(k_dict_init, dim3(c->num_cus * 4), dim3(RFX_BLOCK), 0, c->stream, A.tab, cap, A.idkeys, A.ctl);
    hipLaunchKernelGGL(k_dict_ids, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, A);
    RFX_HIP_CHECK(hipGetLastError());
    unsigned *h = (unsigned *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, A.ctl, 8, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (h[1]) { // more distinct keys than the sample promised
        st->fallbacks++;
        return RFX_ESTATE;
    }
    const i64 nids = (i64)h[0]; // ids 0 .. nids - 1 (0: the null key, used or not)
    if (getenv("RFX_DICT_DEBUG")) fprintf(stderr, "[dict] rows %lld est %.0f idcap %u cap %lld ids %lld caller capacity %lld\n", (long long)nrows, est, idcap, (long long)cap, (long long)nids, (long long)t->capacity);
    // (unfiltered: every id is a group.  Under a filter the dictionary holds keys no selected row has; the merge finds out.)
    if (npred == 0 && (nids - 1 - (i64)grid * (RFX_BLOCK / RFX_WAVE) * DICT_CHUNK) * 4 > t->capacity * 3) {
        rfx_set_error("group_hash_accumulate: hash table full (capacity must be >= 2x the number of distinct keys)");
        return RFX_ELIMIT;
    }
    // dense tables over the ids
    rfx_group_tables_t gt;
    memset(&gt, 0, sizeof(gt));
    gt.kmin = 0;
    gt.range = nids;
    gt.nagg = t->nagg;
    {
        char *p = w + o_tbl;
        const size_t arr = al256((size_t)idcap * 8);
        gt.d_first = (int64_t *)p;
        p += arr;
        for (int a = 0; a < t->nagg; a++) {
            gt.d_acc[a] = p;
            p += arr;
            if (t->d_cnt[a]) {
                gt.d_cnt[a] = (int64_t *)p;
                p += arr;
            }
        }
    }
    if ((rc = rfx_hip_group_tables_init(c, aggs, &gt)) != RFX_OK) return rc;
    int64_t mn = 0, mx = 0, seen = 0;
    // (the scope pass partitions the id column into planes on the way; its answers -- [0, ids) -- are known already)
    // (few ids: the LDS-table kernel of the dense path is one pass already and wants no scope pass)
    if (nids * 12 > 150 * 1024) {
        if ((rc = rfx_hip_group_scope(c, (const int64_t *)A.ids, preds, npred, logic, aggs, t->nagg, nrows, &mn, &mx, &seen)) != RFX_OK) return rc;
        if (seen == 0) return RFX_OK;
    }
    if ((rc = rfx_hip_group_dense_accumulate(c, (const int64_t *)A.ids, preds, npred, logic, aggs, nrows, row0, &gt)) != RFX_OK) return rc;
    // one insert per distinct key into the caller's table set
    return rfx_hash_merge_cells(c, aggs, t, &gt, A.idkeys, 0);
}
