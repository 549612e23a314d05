// rfx_join.hip -- equi-join index and column assembly, built on the group-by tables.
//
// Reference: lj / ij (ray_left_join / ray_inner_join, core/join.c:158-298) ask index_left_join_obj / index_inner_join_obj
// (core/index.c:2886-2990) for, per LEFT row, the FIRST right row with an equal key (ray_find for one key column; an
// open-addressing table over row hashes with a tuple comparison for several), or null.  Then every non-key column is
// assembled row by row: right[idx] where a match exists, else the left row's own value (select_column, core/join.c:38-66);
// the inner join keeps the matched pairs only (get_column, :68-81).
//
// "First right row per key" is exactly the first-occurrence table of the group-by (K7: d_first[slot] = min row id), so the
// BUILD side is rfx_hip_group_dense_accumulate / rfx_hip_group_hash_accumulate with zero aggregates; this file adds the PROBE
// (one streaming pass over the left keys, one random 8-byte read per row) and the null-aware gather.
//   k_join_probe_dense   idx = first[key - kmin] when the key is inside the right side's scope and the slot is occupied
//   k_join_probe_hash    read-only walk of the open-addressing table the hashed group-by filled (same hash, same probing)
//   k_gather_or          out[i] = ids[i] is null ? (left ? left[i] : fill) : right[ids[i]]
#include "rfx_group_common.hpp"

__global__ __launch_bounds__(RFX_BLOCK) void k_join_probe_dense(const u64 *__restrict__ keys, i64 n, u64 kmin, u64 range, const u64 *__restrict__ first,
                                                                i64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 slot = keys[i] - kmin;
        i64 r = RFX_NULL_I64_D;
        if (slot < range) {
            const u64 f = first[slot];
            if (f != (u64)RFX_INF_I64_D) r = (i64)f;
        }
        out[i] = r;
    }
}

// slot_out (optional): the table slot the row's key sits in (capacity = the null key's cell, -1 = not in the table): what lets a group-by emit its
// result by walking ROWS -- a group's representative row already knows its slot -- instead of ranking 2.7e8 slots (rfx_hip_hash_rows_emit)
__global__ __launch_bounds__(RFX_BLOCK) void k_join_probe_hash(const u64 *__restrict__ keys, i64 n, const u64 *__restrict__ tab, i64 capacity,
                                                               const u64 *__restrict__ first, i64 *__restrict__ out, i64 *__restrict__ slot_out) {
    const u64 mask = (u64)capacity - 1;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 key = keys[i];
        i64 r = RFX_NULL_I64_D, at = -1;
        if ((i64)key == RFX_NULL_I64_D) { // the null key has its own cell behind the table (hash_slot)
            const u64 f = first[capacity];
            if (f != (u64)RFX_INF_I64_D) r = (i64)f, at = capacity;
        } else {
            u64 s = rfx_hash_index_u64(RFX_U64_HASH_SEED, key) & mask;
            for (i64 probe = 0; probe < capacity; probe++) {
                const u64 k = tab[s];
                if (k == key) {
                    r = (i64)first[s];
                    at = (i64)s;
                    break;
                }
                if ((i64)k == RFX_NULL_I64_D) break; // an empty slot ends the probe sequence: the key is not on the right side
                s = (s + 1) & mask;
            }
        }
        out[i] = r;
        if (slot_out) slot_out[i] = at;
    }
}

// One hash = one tuple?  ids[r] = the first row of r's group (the probe above, keyed by the row hash): every key column must hold the same cell at r and at
// ids[r] (__index_list_cmp_row, core/index.c:2465-2790, done once per row instead of once per probe step).  ONE pass over the nk key columns -- a row that
// heads its own group (ids[r] == r: every row of the H2O Q7 shape) reads nothing twice -- and ONE counter back (round 5: a gather + a compare pass + a sync
// per key column).
struct TupleKeys {
    const u64 *k[RFX_MAX_KEYS];
};
__global__ __launch_bounds__(RFX_BLOCK) void k_tuple_check(const TupleKeys K, int nk, const i64 *__restrict__ ids, i64 n, unsigned long long *__restrict__ differ) {
    unsigned bad = 0;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 f = ids[i];
        if (f == i || f == RFX_NULL_I64_D) continue; // (null: a row the query's filter did not select -- its hash may not be in the table at all)
        if ((u64)f >= (u64)n) { // (cannot happen: first rows are rows of this table)
            bad++;
            continue;
        }
        for (int c = 0; c < nk; c++) bad += K.k[c][i] != K.k[c][f];
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0) && bad) atomicAdd(differ, (unsigned long long)bad);
}

__global__ __launch_bounds__(RFX_BLOCK) void k_gather_or(const u64 *__restrict__ right, const u64 *__restrict__ left, const i64 *__restrict__ ids, i64 n, u64 fill,
                                                         u64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 id = ids[i];
        out[i] = (id != RFX_NULL_I64_D) ? right[id] : (left ? left[i] : fill);
    }
}

// at_vec_i64_by_i64 / at_vec_f64_by_i64 (core/items.c:53-72): an index that is null, negative or >= len reads as the typed null
__global__ __launch_bounds__(RFX_BLOCK) void k_gather_checked(const u64 *__restrict__ col, i64 len, const i64 *__restrict__ ids, i64 n, u64 null_bits,
                                                              u64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 id = ids[i];
        out[i] = (id >= 0 && id < len) ? col[id] : null_bits;
    }
}

static int join_grid(rfx_ctx *c, i64 n) {
    const i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c) * 4;
    if (blocks < grid) grid = (int)blocks;
    return grid;
}

extern "C" int rfx_hip_join_probe_dense(rfx_ctx_t *c, const int64_t *d_left_keys, int64_t nleft, int64_t kmin, int64_t range, const int64_t *d_first,
                                        int64_t *d_ids) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nleft <= 0) return RFX_OK;
    RFX_REQUIRE(d_left_keys && d_first && d_ids, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(range > 0, RFX_EINVAL, "range must be > 0");
    hipLaunchKernelGGL(k_join_probe_dense, dim3(join_grid(c, nleft)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_left_keys, (i64)nleft, (u64)kmin, (u64)range,
                       (const u64 *)d_first, (i64 *)d_ids);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_join_probe_hash(rfx_ctx_t *c, const int64_t *d_left_keys, int64_t nleft, const rfx_hash_tables_t *t, int64_t *d_ids) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nleft <= 0) return RFX_OK;
    RFX_REQUIRE(d_left_keys && d_ids && t && t->d_keys && t->d_first, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(t->capacity >= 2 && (t->capacity & (t->capacity - 1)) == 0, RFX_EINVAL, "capacity must be a power of two >= 2");
    hipLaunchKernelGGL(k_join_probe_hash, dim3(join_grid(c, nleft)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_left_keys, (i64)nleft, (const u64 *)t->d_keys,
                       (i64)t->capacity, (const u64 *)t->d_first, (i64 *)d_ids, (i64 *)nullptr);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_join_probe_hash_slots(rfx_ctx_t *c, const int64_t *d_left_keys, int64_t nleft, const rfx_hash_tables_t *t, int64_t *d_ids, int64_t *d_slots) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nleft <= 0) return RFX_OK;
    RFX_REQUIRE(d_left_keys && d_ids && d_slots && t && t->d_keys && t->d_first, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(t->capacity >= 2 && (t->capacity & (t->capacity - 1)) == 0, RFX_EINVAL, "capacity must be a power of two >= 2");
    hipLaunchKernelGGL(k_join_probe_hash, dim3(join_grid(c, nleft)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_left_keys, (i64)nleft, (const u64 *)t->d_keys,
                       (i64)t->capacity, (const u64 *)t->d_first, (i64 *)d_ids, (i64 *)d_slots);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_tuple_check(rfx_ctx_t *c, const void *const *d_keys, int nk, const int64_t *d_first_of_row, int64_t nrows, int64_t *differ) {
    RFX_REQUIRE(c && differ, RFX_EINVAL, "NULL argument");
    *differ = 0;
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_keys && d_first_of_row && nk >= 1 && nk <= RFX_MAX_KEYS, RFX_EINVAL, "bad argument");
    TupleKeys K;
    for (int i = 0; i < RFX_MAX_KEYS; i++) K.k[i] = i < nk ? (const u64 *)d_keys[i] : nullptr;
    int rc = rfx_ws_reserve(c, 256);
    if (rc != RFX_OK) return rc;
    unsigned long long *cnt = (unsigned long long *)c->d_ws;
    RFX_HIP_CHECK(hipMemsetAsync(cnt, 0, 8, c->stream));
    hipLaunchKernelGGL(k_tuple_check, dim3(join_grid(c, nrows)), dim3(RFX_BLOCK), 0, c->stream, K, nk, (const i64 *)d_first_of_row, (i64)nrows, cnt);
    RFX_HIP_CHECK(hipGetLastError());
    unsigned long long *h = (unsigned long long *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    *differ = (int64_t)h[0];
    return RFX_OK;
}

extern "C" int rfx_hip_gather_or(rfx_ctx_t *c, const void *d_right, const void *d_left, const int64_t *d_ids, int64_t n, uint64_t fill_bits, void *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_right && d_ids && d_out, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_gather_or, dim3(join_grid(c, n)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_right, (const u64 *)d_left, (const i64 *)d_ids, (i64)n,
                       (u64)fill_bits, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_gather_checked(rfx_ctx_t *c, const void *d_col, int64_t col_len, int32_t col_type, const int64_t *d_ids, int64_t m, void *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (m <= 0) return RFX_OK;
    RFX_REQUIRE((d_col || col_len == 0) && d_ids && d_out && col_len >= 0, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(col_type == RFX_I64 || col_type == RFX_F64, RFX_EINVAL, "column type must be RFX_I64 or RFX_F64");
    hipLaunchKernelGGL(k_gather_checked, dim3(join_grid(c, m)), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_col, (i64)col_len, (const i64 *)d_ids, (i64)m,
                       col_type == RFX_F64 ? (u64)RFX_NAN_BITS : (u64)RFX_NULL_I64_D, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// slot -> group id of the dense tables `t` after rfx_hip_group_rank (the key table of the reference's INDEX_TYPE_SHIFT index,
// core/index.c:2037-2062): NULL_I64 for a slot no row maps to.
__global__ __launch_bounds__(RFX_BLOCK) void k_slot_ids(const i64 *__restrict__ gid, i64 n, i64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 g = gid[i];
        out[i] = g < 0 ? RFX_NULL_I64_D : g;
    }
}
extern "C" int rfx_hip_group_slot_ids(rfx_ctx_t *c, const rfx_group_tables_t *t, int64_t *d_out) {
    RFX_REQUIRE(c && t && d_out, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(t->range > 0 && c->gid_cap >= (size_t)t->range && c->d_gid, RFX_ESTATE, "group_slot_ids without group_rank");
    hipLaunchKernelGGL(k_slot_ids, dim3(join_grid(c, t->range)), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)c->d_gid, (i64)t->range, (i64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
