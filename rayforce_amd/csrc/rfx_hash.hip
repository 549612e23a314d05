// rfx_hash.hip -- K9 sparse-key group-by (range > rows) and the reference's hash primitives.
//
// Reference: index_group_i64_unscoped -> index_group_distribute (core/index.c:1959-1977, 1777-1911) builds one
// open-addressed table per CPU chunk (ht_oa_create / ht_oa_tab_next_with, core/hash.c:35-148: prime size >= len/0.75,
// linear probing, empty slot = NULL_I64, fnv1a hash) and merges them sequentially; the group ORDER of that merge is
// implementation-defined.  Here: ONE device-wide open-addressed table (power-of-two capacity, linear
// probing, empty = NULL_I64 as in the reference, 64-bit CAS insert, hash = hash_index_u64 of core/hash.h:86-97), cells
// indexed by slot exactly like the dense tables, so first-row tracking, ranking and emit are shared with the dense
// path and the group order is FIRST OCCURRENCE (what the reference itself produces when it runs on one thread).
// A NULL_I64 key cannot be stored as a key (it is the empty marker); it gets the dedicated slot `capacity`
// (the reference's own table would open a fresh group for every such row -- see DESIGN.md "deliberate deviations").
#include "rfx_group_common.hpp"

// hash_fnv1a -- core/hash.c:530-542
__device__ __host__ __forceinline__ u64 rfx_hash_fnv1a(u64 key) {
    u64 h = 14695981039346656037ULL;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        h ^= (key >> (i * 8)) & 0xff;
        h *= 1099511628211ULL;
    }
    return h;
}


// A PACKED table (round 6, the row-hash route's device-wide table): entry e = the W cells [e * W, (e + 1) * W) of ONE block -- key, first row, accumulators,
// counts side by side, so that an insert touches one line where the array-per-field layout touches one per field (four for `count, sum`: 44 GB moved for 1e8
// inserts).  The table's array pointers are the block's first W cells' addresses and every index is e * W: find-or-insert answers that, the insert pass
// records it as the row's "slot", and k_slot_first / k_emit_rows index the arrays with it as they always did.
__device__ __forceinline__ i64 hash_slot_ins_packed(u64 *keys, i64 capacity, int W, u64 key, unsigned &inserted) {
    if ((i64)key == RFX_NULL_I64_D) return capacity * W;
    const u64 mask = (u64)capacity - 1;
    u64 s = rfx_hash_index_u64(RFX_U64_HASH_SEED, key) & mask;
    const i64 bound = capacity < RFX_HASH_MAX_PROBES ? capacity : RFX_HASH_MAX_PROBES;
    for (i64 probe = 0; probe < bound; probe++) {
        u64 *p = keys + s * (u64)W;
        const u64 k = *p;
        if (k == key) return (i64)(s * (u64)W);
        if ((i64)k == RFX_NULL_I64_D) {
            const u64 old = atomicCAS((unsigned long long *)p, (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key);
            if ((i64)old == RFX_NULL_I64_D) {
                inserted++;
                return (i64)(s * (u64)W);
            }
            if (old == key) return (i64)(s * (u64)W);
        }
        s = (s + 1) & mask;
    }
    return -1;
}

template <int NC, bool PACKED>
__global__ __launch_bounds__(RFX_BLOCK) void k_group_hash(const Plan P, const HashArgs H, int *__restrict__ overflow, i64 *__restrict__ row_slot, int stride) {
    constexpr int U = (NC <= 2) ? 2 : 1;
    constexpr int E = 2 * U;
    constexpr int TILE = RFX_BLOCK * E;
    constexpr int JSTRIDE = RFX_BLOCK * 2;
    const int tid = threadIdx.x;
    PredSet<RFX_MAX_PREDS> S;
    predset_load<RFX_MAX_PREDS>(P, S);
    const i64 ntiles = (P.nrows + TILE - 1) / TILE;
    unsigned pending = 0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (*(volatile int *)overflow) return; // another workgroup has found the table full: the caller will grow it and retry
        const i64 base = t * TILE + tid * 2;
        u64 v[NC][E];
        unsigned valid = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            i64 row = base + (i64)(e >> 1) * JSTRIDE + (e & 1);
            bool in = row < P.nrows;
            valid |= (unsigned)in << e;
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][e] = in ? P.cols[c][row] : 0ULL;
        }
        const unsigned m = eval_preds<NC, E, RFX_MAX_PREDS>(S, v, valid);
        if (__ballot(m != 0) == 0) { // wave-uniform: the slot count below is reduced across the wave
            if (row_slot) {
#pragma unroll
                for (int u = 0; u < U; u++)
                    if ((valid >> (2 * u)) & 1u) {
                        const i64 r = base + (i64)u * JSTRIDE;
                        if ((valid >> (2 * u + 1)) & 1u) *(longlong2 *)(row_slot + r) = make_longlong2(-1, -1);
                        else row_slot[r] = -1;
                    }
            }
            continue;
        }
        u64 key[E];
        sel_col<NC, E>(key, v, H.key_idx);
        i64 slot[E];
        unsigned fresh = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            slot[e] = -1;
            if (!((m >> e) & 1u)) continue;
            slot[e] = PACKED ? hash_slot_ins_packed(H.keys, H.capacity, stride, key[e], fresh) : hash_slot_ins(H.keys, H.capacity, key[e], fresh);
            if (slot[e] < 0) {
                atomicExch(overflow, 1);
                continue;
            }
            const u64 row = (u64)(P.row0 + base + (i64)(e >> 1) * JSTRIDE + (e & 1));
            if (row < H.first[slot[e]]) atomicMin((unsigned long long *)&H.first[slot[e]], (unsigned long long)row);
        }
        // every row's slot (-1: not selected), for the caller that wants it (the row-hash route: saves probing the table for all rows again)
        if (row_slot) {
#pragma unroll
            for (int u = 0; u < U; u++)
                if ((valid >> (2 * u)) & 1u) {
                    const i64 r = base + (i64)u * JSTRIDE;
                    if ((valid >> (2 * u + 1)) & 1u) *(longlong2 *)(row_slot + r) = make_longlong2(slot[2 * u], slot[2 * u + 1]);
                    else row_slot[r] = slot[2 * u];
                }
        }
        // load factor: one counter update per wave and tile that claimed slots; beyond 3/4 the launch is abandoned (grow and retry)
        for (int sft = 32; sft >= 1; sft >>= 1) fresh += __shfl_xor(fresh, sft, 64);
        pending += fresh; // wave-uniform; pushed to the one global counter in batches (a single address takes ~10 M atomics/s)
        if (pending >= 4096u) {
            if ((tid & 63) == 0) {
                const unsigned long long used = atomicAdd((unsigned long long *)(overflow + 2), (unsigned long long)pending) + pending;
                if (used * 4 > (unsigned long long)H.capacity * 3) atomicExch(overflow, 1);
            }
            pending = 0;
        }
        for (int a = 0; a < H.nagg; a++) {
            const PlanAgg ag = P.aggs[a];
            u64 x[E];
            if (ag.col >= RFX_XCOL) expr_input<NC, E>(x, v, P.xs[ag.col - RFX_XCOL]);
            else if (ag.col >= 0) sel_col<NC, E>(x, v, ag.col);
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (slot[e] < 0) continue;
                group_apply(&H.acc[a][slot[e]], H.cnt[a] ? &H.cnt[a][slot[e]] : (u64 *)0, ag.kind, ag.f64, x[e], ag.skipnull);
            }
        }
    }
    // what is left of the batch (a last tile without selected rows skips the loop body: the flush cannot live there)
    if (pending && (tid & 63) == 0) {
        const unsigned long long used = atomicAdd((unsigned long long *)(overflow + 2), (unsigned long long)pending) + pending;
        if (used * 4 > (unsigned long long)H.capacity * 3) atomicExch(overflow, 1);
    }
}

struct MergeArgs {
    i64 capacity;
    int nagg;
    int kinds[RFX_MAX_AGGS];
    int f64s[RFX_MAX_AGGS];
    u64 *keys;
    u64 *first;
    u64 *acc[RFX_MAX_AGGS];
    u64 *cnt[RFX_MAX_AGGS];
    const u64 *fkeys;
    const u64 *ffirst;
    const u64 *facc[RFX_MAX_AGGS];
    const u64 *fcnt[RFX_MAX_AGGS];
};

__global__ __launch_bounds__(RFX_BLOCK) void k_hash_merge(const MergeArgs M, int *__restrict__ overflow) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i <= M.capacity; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 f = M.ffirst[i];
        if (f == (u64)RFX_INF_I64_D) continue;
        const u64 key = (i == M.capacity) ? (u64)RFX_NULL_I64_D : M.fkeys[i];
        const i64 s = hash_slot(M.keys, M.capacity, key);
        if (s < 0) {
            atomicExch(overflow, 1);
            continue;
        }
        if (f < M.first[s]) atomicMin((unsigned long long *)&M.first[s], (unsigned long long)f);
        for (int a = 0; a < M.nagg; a++)
            group_merge_cell(&M.acc[a][s], M.cnt[a] ? &M.cnt[a][s] : (u64 *)0, M.kinds[a], M.f64s[a], M.facc[a][i], M.fcnt[a] ? M.fcnt[a][i] : 0ULL);
    }
}

int rfx_estimate_distinct(rfx_ctx *c, const u64 *d_key, i64 nrows, double *est);                                          // rfx_group_part.hip
int rfx_plane_hash_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const HashArgs &H, double est, int *d_overflow); // rfx_group_plane.hip

static int check_hash(const rfx_agg_t *aggs, const rfx_hash_tables_t *t) {
    RFX_REQUIRE(t && t->d_keys && t->d_first, RFX_EINVAL, "hash tables / d_keys / d_first is NULL");
    RFX_REQUIRE(t->capacity >= 2 && (t->capacity & (t->capacity - 1)) == 0, RFX_EINVAL, "capacity must be a power of two >= 2");
    RFX_REQUIRE(t->nagg >= 0 && t->nagg <= RFX_MAX_AGGS, RFX_ELIMIT, "too many aggregates");
    for (int a = 0; a < t->nagg; a++) {
        RFX_REQUIRE(t->d_acc[a] != NULL, RFX_EINVAL, "d_acc[a] is NULL");
        if (agg_has_cnt(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64)) RFX_REQUIRE(t->d_cnt[a] != NULL, RFX_EINVAL, "d_cnt[a] is NULL for SUM(i64)/AVG");
    }
    return RFX_OK;
}

// every array of a hash table set has capacity + 1 cells (the extra one is the NULL-key group)
extern "C" int rfx_hip_hash_tables_init(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *t) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    const i64 n = t->capacity + 1;
    if ((rc = rfx_fill_u64(c, t->d_keys, n, (u64)RFX_NULL_I64_D)) != RFX_OK) return rc;
    if ((rc = rfx_fill_u64(c, t->d_first, n, (u64)RFX_INF_I64_D)) != RFX_OK) return rc;
    for (int a = 0; a < t->nagg; a++) {
        if ((rc = rfx_fill_u64(c, t->d_acc[a], n, acc_identity(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64))) != RFX_OK) return rc;
        if (t->d_cnt[a] && (rc = rfx_fill_u64(c, t->d_cnt[a], n, 0)) != RFX_OK) return rc;
    }
    return RFX_OK;
}

static int read_overflow(rfx_ctx *c, int *d_flag, const char *what) {
    int *h = (int *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (h[0]) {
        rfx_set_error("%s: hash table full (capacity must be >= 2x the number of distinct keys)", what);
        return RFX_ELIMIT;
    }
    return RFX_OK;
}

template <int NC>
static void launch_hash(rfx_ctx *c, const Plan &P, const HashArgs &H, int grid, int *flag, i64 *row_slot, int stride = 1) {
    if (stride > 1) hipLaunchKernelGGL((k_group_hash<NC, true>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, H, flag, row_slot, stride);
    else hipLaunchKernelGGL((k_group_hash<NC, false>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, P, H, flag, row_slot, 1);
}
static void launch_hash_nc(rfx_ctx *c, const Plan &P, const HashArgs &H, int grid, int *flag, i64 *rs, int stride) {
    switch (P.ncols) {
        case 1: launch_hash<1>(c, P, H, grid, flag, rs, stride); break;
        case 2: launch_hash<2>(c, P, H, grid, flag, rs, stride); break;
        case 3: launch_hash<3>(c, P, H, grid, flag, rs, stride); break;
        case 4: launch_hash<4>(c, P, H, grid, flag, rs, stride); break;
        case 5: launch_hash<5>(c, P, H, grid, flag, rs, stride); break;
        case 6: launch_hash<6>(c, P, H, grid, flag, rs, stride); break;
        case 7: launch_hash<7>(c, P, H, grid, flag, rs, stride); break;
        default: launch_hash<8>(c, P, H, grid, flag, rs, stride); break;
    }
}

extern "C" int rfx_hip_group_hash_accumulate(rfx_ctx_t *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic,
                                             const rfx_agg_t *aggs, int64_t nrows, int64_t row0, const rfx_hash_tables_t *t) {
    return rfx_hip_group_hash_accumulate_slots(c, d_key, preds, npred, logic, aggs, nrows, row0, t, NULL, NULL);
}
// ... the same, and -- when the rows went straight into the device-wide table (a table too large for the partitioned forms: about as many groups as rows) --
// every row's slot in d_row_slots[nrows] (-1: not selected), *recorded = 1.  The partitioned forms aggregate in LDS tables and merge: no row learns its slot
// there (*recorded = 0: probe the table).
extern "C" int rfx_hip_group_hash_accumulate_slots(rfx_ctx_t *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int64_t nrows,
                                                   int64_t row0, const rfx_hash_tables_t *t, int64_t *d_row_slots, int *recorded) {
    if (recorded) *recorded = 0;
    RFX_REQUIRE(c && d_key, RFX_EINVAL, "NULL argument");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    if (nrows == 0) return RFX_OK;
    Plan P;
    int key_idx = 0;
    rc = rfx_plan_build(&P, preds, npred, logic, aggs, t->nagg, d_key, &key_idx, nrows, row0);
    if (rc == RFX_OK && rfx_plan_has_deep_expr(P) && P.ncols + P.nx > RFX_MAX_COLS) rc = RFX_ELIMIT; // no room to materialise the trees
    if (rc == RFX_ELIMIT && t->nagg > 1) { // too many columns / expressions for one launch: two passes, same keys and slots
        const int h = t->nagg / 2;
        rfx_hash_tables_t t1 = *t, t2 = *t;
        t1.nagg = h;
        t2.nagg = t->nagg - h;
        for (int a = 0; a < t2.nagg; a++) {
            t2.d_acc[a] = t->d_acc[h + a];
            t2.d_cnt[a] = t->d_cnt[h + a];
        }
        rc = rfx_hip_group_hash_accumulate_slots(c, d_key, preds, npred, logic, aggs, nrows, row0, &t1, d_row_slots, recorded); // (the same keys, the same slots: once)
        if (rc != RFX_OK) return rc;
        return rfx_hip_group_hash_accumulate(c, d_key, preds, npred, logic, aggs + h, nrows, row0, &t2);
    }
    if (rc != RFX_OK) return rc;
    if (rfx_plan_has_deep_expr(P)) { // expression trees: scratch columns first (the kernels here evaluate single operations only)
        RFX_REQUIRE(P.ncols + P.nx <= RFX_MAX_COLS, RFX_ELIMIT, "too many distinct columns once the expression trees are materialised");
        rc = rfx_plan_materialise_exprs(c, &P);
        if (rc != RFX_OK) return rc;
    }
    HashArgs H;
    memset(&H, 0, sizeof(H));
    H.capacity = t->capacity;
    H.key_idx = key_idx;
    H.nagg = t->nagg;
    H.keys = (u64 *)t->d_keys;
    H.first = (u64 *)t->d_first;
    for (int a = 0; a < t->nagg; a++) {
        H.acc[a] = (u64 *)t->d_acc[a];
        H.cnt[a] = (u64 *)t->d_cnt[a];
    }
    rc = rfx_ws_reserve(c, ((size_t)1 << 21) + 512); // the flag block + the distinct-count sample's table (rfx_group_part.hip): reserved once, the flag must not move
    if (rc != RFX_OK) return rc;
    int *flag = (int *)c->d_ws;
    RFX_HIP_CHECK(hipMemsetAsync(flag, 0, 16, c->stream)); // [0] full, [2..3] slots claimed by this launch
    // large inputs, round 3: hash-partitioned PLANES {key, value, meta} through barrier-free LDS rings, LDS hash tables per partition share
    // (rfx_group_plane.hip); sized by the sampled distinct-key estimate
    if (nrows >= (1 << 16) && nrows < (1LL << 32) && !(c->flags & RFX_TUNE_NO_PARTITION)) {
        double est = 0;
        rc = rfx_estimate_distinct(c, (const u64 *)P.cols[key_idx], nrows, &est);
        if (rc != RFX_OK) return rc;
        rc = rfx_plane_hash_accumulate(c, P, key_idx, H, est, flag);
        if (rc == RFX_OK) return read_overflow(c, flag, "group_hash_accumulate");
        if (rc != RFX_ESTATE) return rc;
    }
    // round 1's form: partition by hash (histogram, exact offsets), aggregate every partition in an LDS table, merge once (rfx_group_part.hip)
    rc = rfx_group_part_hash_accumulate(c, P, key_idx, H, flag);
    if (rc == RFX_OK) return read_overflow(c, flag, "group_hash_accumulate");
    if (rc != RFX_ESTATE) return rc;
    if (c->ext_i[2] > 0 && npred == 0 && t->capacity < 2 * nrows && (double)c->ext_i[2] > 0.6 * (double)t->capacity) {
        // (unfiltered inputs only, and never at the reference's own size of 2 x rows: there the estimate -- which errs high -- cannot turn into an error)
        // the sampled distinct-key estimate says this table is too small: say so before a launch finds out at 3/4 load (6 ms at 1e8 keys)
        rfx_set_error("group_hash_accumulate: about %lld distinct keys for %lld slots (sampled estimate): hash table too small", (long long)c->ext_i[2], (long long)t->capacity);
        return RFX_ELIMIT;
    }
    int grid = rfx_grid(c) * 4;
    i64 *rs = (d_row_slots && recorded) ? (i64 *)d_row_slots : NULL;
    launch_hash_nc(c, P, H, grid, flag, rs, 1);
    RFX_HIP_CHECK(hipGetLastError());
    rc = read_overflow(c, flag, "group_hash_accumulate");
    if (rc == RFX_OK && rs) *recorded = 1;
    return rc;
}

// ---- the packed form (see hash_slot_ins_packed): t's array pointers are the first entry's cells, `stride` the cells per entry ----
struct PackedPat {
    u64 v[2 + 2 * RFX_MAX_AGGS];
};
__global__ __launch_bounds__(RFX_BLOCK) void k_fill_packed(u64 *__restrict__ base, i64 cells, int W, const PackedPat pat) {
    const i64 step = (i64)gridDim.x * RFX_BLOCK;
    i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x;
    int r = (int)(i % W);
    const int dr = (int)(step % W); // (the cell's place in its entry, carried along: no 64-bit division per cell)
    for (; i < cells; i += step) {
        base[i] = pat.v[r];
        r += dr;
        r -= r >= W ? W : 0;
    }
}
static int packed_offsets_ok(const rfx_hash_tables_t *t, int stride) { // every array one of the entry's cells, no two the same
    unsigned seen = 0;
    const char *b = (const char *)t->d_keys;
    auto cell = [&](const void *p) -> int {
        const ptrdiff_t o = (const char *)p - b;
        if (o < 0 || (o & 7) || o / 8 >= stride || (seen & (1u << (o / 8)))) return -1;
        seen |= 1u << (o / 8);
        return (int)(o / 8);
    };
    if (cell(t->d_keys) != 0 || cell(t->d_first) < 0) return 0;
    for (int a = 0; a < t->nagg; a++)
        if (cell(t->d_acc[a]) < 0 || (t->d_cnt[a] && cell(t->d_cnt[a]) < 0)) return 0;
    return 1;
}
extern "C" int rfx_hip_hash_tables_init_packed(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int stride) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(stride >= 2 && stride <= 2 + 2 * RFX_MAX_AGGS && packed_offsets_ok(t, stride), RFX_EINVAL, "packed table: the arrays must be distinct cells of the first entry");
    PackedPat pat;
    for (int j = 0; j < stride; j++) pat.v[j] = 0; // (cells no array names: padding)
    const u64 *b = (const u64 *)t->d_keys;
    pat.v[0] = (u64)RFX_NULL_I64_D;
    pat.v[(const u64 *)t->d_first - b] = (u64)RFX_INF_I64_D;
    for (int a = 0; a < t->nagg; a++) {
        pat.v[(const u64 *)t->d_acc[a] - b] = acc_identity(aggs[a].kind, rfx_agg_input_type(&aggs[a]) == RFX_F64);
        if (t->d_cnt[a]) pat.v[(const u64 *)t->d_cnt[a] - b] = 0;
    }
    hipLaunchKernelGGL(k_fill_packed, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (u64 *)t->d_keys, (i64)(t->capacity + 1) * stride, stride, pat);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
// every row straight into the packed table, its scaled slot into d_row_slots[nrows] (-1: not selected).  RFX_ESTATE: a query this form does not carry
// (expression trees; more columns than one launch reads): the caller lays the table out field by field and takes rfx_hip_group_hash_accumulate_slots
extern "C" int rfx_hip_group_hash_accumulate_packed(rfx_ctx_t *c, const int64_t *d_key, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int64_t nrows,
                                                    int64_t row0, const rfx_hash_tables_t *t, int stride, int64_t *d_row_slots) {
    RFX_REQUIRE(c && d_key && d_row_slots, RFX_EINVAL, "NULL argument");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(stride >= 2 && stride <= 2 + 2 * RFX_MAX_AGGS && packed_offsets_ok(t, stride), RFX_EINVAL, "packed table: the arrays must be distinct cells of the first entry");
    if (nrows == 0) return RFX_OK;
    Plan P;
    int key_idx = 0;
    rc = rfx_plan_build(&P, preds, npred, logic, aggs, t->nagg, d_key, &key_idx, nrows, row0);
    if (rc == RFX_ELIMIT || (rc == RFX_OK && rfx_plan_has_deep_expr(P))) return RFX_ESTATE;
    if (rc != RFX_OK) return rc;
    HashArgs H;
    memset(&H, 0, sizeof(H));
    H.capacity = t->capacity;
    H.key_idx = key_idx;
    H.nagg = t->nagg;
    H.keys = (u64 *)t->d_keys;
    H.first = (u64 *)t->d_first;
    for (int a = 0; a < t->nagg; a++) {
        H.acc[a] = (u64 *)t->d_acc[a];
        H.cnt[a] = (u64 *)t->d_cnt[a];
    }
    rc = rfx_ws_reserve(c, ((size_t)1 << 21) + 512);
    if (rc != RFX_OK) return rc;
    int *flag = (int *)c->d_ws;
    RFX_HIP_CHECK(hipMemsetAsync(flag, 0, 16, c->stream));
    launch_hash_nc(c, P, H, rfx_grid(c) * 4, flag, (i64 *)d_row_slots, stride);
    RFX_HIP_CHECK(hipGetLastError());
    return read_overflow(c, flag, "group_hash_accumulate_packed");
}

// every row's group-first row from its slot (what the probe answers beside the slot): ids[i] = first[slot[i]], the null for a row without one
__global__ __launch_bounds__(RFX_BLOCK) void k_slot_first(const i64 *__restrict__ row_slot, const u64 *__restrict__ first, i64 n, i64 *__restrict__ ids) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 s = row_slot[i];
        ids[i] = s < 0 ? RFX_NULL_I64_D : (i64)first[s];
    }
}
extern "C" int rfx_hip_hash_slot_first(rfx_ctx_t *c, const rfx_hash_tables_t *t, const int64_t *d_row_slots, int64_t nrows, int64_t *d_ids) {
    RFX_REQUIRE(c && t && t->d_first, RFX_EINVAL, "NULL argument");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_row_slots && d_ids, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_slot_first, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_row_slots, (const u64 *)t->d_first, (i64)nrows, (i64 *)d_ids);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_hash_tables_merge(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *into,
                                         const rfx_hash_tables_t *from) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    int rc = check_hash(aggs, into);
    if (rc != RFX_OK) return rc;
    rc = check_hash(aggs, from);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(into->capacity == from->capacity && into->nagg == from->nagg, RFX_EINVAL, "tables differ in shape");
    MergeArgs M;
    memset(&M, 0, sizeof(M));
    M.capacity = into->capacity;
    M.nagg = into->nagg;
    M.keys = (u64 *)into->d_keys;
    M.first = (u64 *)into->d_first;
    M.fkeys = (const u64 *)from->d_keys;
    M.ffirst = (const u64 *)from->d_first;
    for (int a = 0; a < into->nagg; a++) {
        M.kinds[a] = aggs[a].kind;
        M.f64s[a] = rfx_agg_input_type(&aggs[a]) == RFX_F64;
        M.acc[a] = (u64 *)into->d_acc[a];
        M.cnt[a] = (u64 *)into->d_cnt[a];
        M.facc[a] = (const u64 *)from->d_acc[a];
        M.fcnt[a] = (const u64 *)from->d_cnt[a];
    }
    rc = rfx_ws_reserve(c, 256);
    if (rc != RFX_OK) return rc;
    int *flag = (int *)c->d_ws;
    RFX_HIP_CHECK(hipMemsetAsync(flag, 0, 4, c->stream));
    hipLaunchKernelGGL(k_hash_merge, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, M, flag);
    RFX_HIP_CHECK(hipGetLastError());
    return read_overflow(c, flag, "hash_tables_merge");
}

extern "C" int rfx_hip_hash_rank(rfx_ctx_t *c, const rfx_hash_tables_t *t, int64_t total_rows, int64_t *ngroups) {
    RFX_REQUIRE(c && t && ngroups, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(t->d_first && t->capacity > 0, RFX_EINVAL, "bad tables");
    return rfx_rank_slots(c, (const u64 *)t->d_first, t->capacity + 1, 0, total_rows, (i64 *)ngroups);
}

extern "C" int rfx_hip_hash_emit(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int64_t *d_keys,
                                 int64_t *d_first_ids, void *const *d_results) {
    return rfx_hip_hash_emit_sharded(c, aggs, t, 0, 0, d_keys, d_first_ids, d_results);
}
extern "C" int rfx_hip_hash_emit_sharded(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int64_t row0, int64_t local_rows,
                                         int64_t *d_keys, int64_t *d_first_ids, void *const *d_results) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(local_rows >= 0, RFX_EINVAL, "local_rows < 0");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(c->gid_cap >= (size_t)t->capacity + 1, RFX_ESTATE, "hash_emit without hash_rank");
    EmitArgs A;
    memset(&A, 0, sizeof(A));
    A.slots = t->capacity + 1;
    A.nagg = t->nagg;
    A.first = (const u64 *)t->d_first;
    A.keys = (const u64 *)t->d_keys;
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

// ---------------- emit by ROWS (round 6) ----------------
// A hashed group-by over MANY groups (the row-hash route's 1e8 groups in 2.7e8 slots) spent most of its time ranking and walking SLOTS: five passes over
// the 2.1 GB first-row array, a 2.1 GB slot -> id array, an inverse permutation written at random and a gather of every table array through it (52 GB read
// to write 3.2 GB: profiles/r05_pmc_detail.json).  With every row's group-first row at hand (the probe the tuple proof makes anyway) the groups ARE the rows
// that head their own group (probe[r] == r), and those rows in ascending order are the first-occurrence order: a byte mask, the `where` compaction
// (rfx_hip_where_begin / _emit) and ONE gather per table array at the representative rows' slots (k_join_probe_hash's slot_out).
__global__ __launch_bounds__(RFX_BLOCK) void k_rep_mask(const i64 *__restrict__ probe, i64 n, signed char *__restrict__ mask) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) mask[i] = probe[i] == i;
}
__global__ __launch_bounds__(RFX_BLOCK) void k_emit_rows(const EmitArgs A, const i64 *__restrict__ row_slot, const i64 *__restrict__ row_keys, i64 groups) {
    for (i64 g = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; g < groups; g += (i64)gridDim.x * RFX_BLOCK) {
        const i64 f = A.out_first[g]; // (the compaction wrote the representative rows there: the groups' first rows, ascending)
        const i64 i = row_slot[f];
        // the group's key = its first row's key: read from the grouped-on COLUMN at an ascending row (a stream) where the table's key array would be one more random line
        if (A.out_keys) A.out_keys[g] = row_keys ? row_keys[f] : (i64)A.keys[i];
        for (int a = 0; a < A.nagg; a++) {
            if (!A.out[a]) continue;
            if (A.kinds[a] == RFX_AGG_FIRST) {
                const i64 lr = f - A.row0;
                A.out[a][g] = (A.col[a] && lr >= 0 && (A.nloc == 0 || lr < A.nloc)) ? A.col[a][lr] : 0ULL;
            } else A.out[a][g] = group_final(A.kinds[a], A.f64s[a], A.acc[a][i], A.cnt[a] ? A.cnt[a][i] : 0ULL, A.skips[a]);
        }
    }
}
extern "C" int rfx_hip_hash_rows_begin(rfx_ctx_t *c, const int64_t *d_probe_first, int64_t nrows, int64_t *ngroups) {
    RFX_REQUIRE(c && ngroups, RFX_EINVAL, "NULL argument");
    *ngroups = 0;
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_probe_first, RFX_EINVAL, "NULL argument");
    void *mask = NULL;
    int rc = rfx_hip_malloc(c, &mask, (size_t)nrows + 64);
    if (rc != RFX_OK) return rc;
    hipLaunchKernelGGL(k_rep_mask, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_probe_first, (i64)nrows, (signed char *)mask);
    rc = hipGetLastError() == hipSuccess ? rfx_hip_where_begin(c, NULL, 0, RFX_AND, (const int8_t *)mask, nrows, ngroups) : RFX_EHIP; // (syncs: the mask has been read)
    rfx_hip_free(c, mask);
    return rc;
}
extern "C" int rfx_hip_hash_rows_emit(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, const int64_t *d_row_slots, const int64_t *d_row_keys, int64_t row0,
                                      int64_t local_rows, int64_t ngroups, int64_t *d_keys, int64_t *d_first_ids, void *const *d_results) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(local_rows >= 0 && ngroups >= 0, RFX_EINVAL, "bad argument");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    if (ngroups == 0) return RFX_OK;
    RFX_REQUIRE(d_row_slots && d_first_ids, RFX_EINVAL, "NULL argument");
    rc = rfx_hip_where_emit(c, 0, d_first_ids); // the representative rows, ascending: the groups' first rows in first-occurrence order
    if (rc != RFX_OK) return rc;
    EmitArgs A;
    memset(&A, 0, sizeof(A));
    A.slots = t->capacity + 1;
    A.nagg = t->nagg;
    A.first = (const u64 *)t->d_first;
    A.keys = (const u64 *)t->d_keys;
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
    hipLaunchKernelGGL(k_emit_rows, dim3(rfx_grid(c) * 4), dim3(RFX_BLOCK), 0, c->stream, A, (const i64 *)d_row_slots, (const i64 *)d_row_keys, (i64)ngroups);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_hash_rank_emit(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_hash_tables_t *t, int64_t total_rows, int64_t row0, int64_t local_rows,
                                      int nsl, int si, int64_t out_cap, int64_t *d_keys, int64_t *d_first_ids, void *const *d_results, int64_t *ngroups) {
    RFX_REQUIRE(c && t && ngroups, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(local_rows >= 0, RFX_EINVAL, "local_rows < 0");
    int rc = check_hash(aggs, t);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(t->capacity + 1 <= RFX_RANK_EMIT_MAX, RFX_EINVAL, "rank_emit: at most RFX_RANK_EMIT_MAX slots");
    EmitArgs A;
    memset(&A, 0, sizeof(A));
    A.slots = t->capacity + 1;
    A.nagg = t->nagg;
    A.first = (const u64 *)t->d_first;
    A.keys = (const u64 *)t->d_keys;
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

// ---------------- hash primitives (pinned against the compiled reference in tests/golden) ----------------
__global__ __launch_bounds__(RFX_BLOCK) void k_fnv1a(const u64 *__restrict__ in, i64 n, u64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) out[i] = rfx_hash_fnv1a(in[i]);
}
__global__ __launch_bounds__(RFX_BLOCK) void k_mix(const u64 *__restrict__ in, i64 n, u64 h, u64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) out[i] = rfx_hash_index_u64(h, in[i]);
}
extern "C" int rfx_hip_hash_fnv1a_i64(rfx_ctx_t *c, const int64_t *d_in, int64_t n, uint64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in && d_out, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_fnv1a, dim3(rfx_grid(c)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_in, (i64)n, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_hash_mix_u64(rfx_ctx_t *c, const uint64_t *d_in, int64_t n, uint64_t h, uint64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in && d_out, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_mix, dim3(rfx_grid(c)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_in, (i64)n, (u64)h, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
