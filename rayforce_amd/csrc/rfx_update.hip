// rfx_update.hip -- the device half of `update ... where / by` (ray_update, core/update.c:936-1106).
//
// The reference evaluates `where:` to row ids (ray_where, update.c:1001), the mappings over the filtered / grouped table
// (MAPFILTER / MAPGROUP columns, update.c:1048-1080), and then writes: under a filter value i goes to row ids[i] (set_ids,
// __update_table's filter arm); under `by:` every group's aggregate goes to all of that group's (selected) rows
// (aggr_row + set_ids per group, update.c:781-850).  Here the row ids come from K3, element-wise mappings from rfx_hip_eval_expr,
// group aggregates from the K7/K10 tables, and these two kernels do the writes on a device copy of the column:
//   k_update_set     col[ids[i]] = vals ? vals[ids[i]] : atom          (vals: the mapping evaluated over ALL rows; element-wise, so
//                                                                        its value at row ids[i] is what the filtered evaluation yields)
//   k_update_group   col[row] = final aggregate of row's group          (slot = key[row] - kmin; no per-group id lists are built)
#include "rfx_group_common.hpp"

__global__ __launch_bounds__(RFX_BLOCK) void k_update_set(u64 *__restrict__ col, const i64 *__restrict__ ids, i64 m, const u64 *__restrict__ vals, u64 atom) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < m; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 r = ids ? ids[i] : i;
        col[r] = vals ? vals[r] : atom;
    }
}

struct UpdGroupArgs {
    i64 kmin, range, row0;
    int kind, f64, skip, _pad;
    const u64 *first, *acc, *cnt, *src; // src: the aggregate's column (FIRST)
};
__global__ __launch_bounds__(RFX_BLOCK) void k_update_group(u64 *__restrict__ col, const i64 *__restrict__ key, const i64 *__restrict__ ids, i64 m,
                                                            const UpdGroupArgs A) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < m; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 r = ids ? ids[i] : i;
        const u64 slot = (u64)key[r] - (u64)A.kmin;
        if (slot >= (u64)A.range) continue;
        u64 v;
        if (A.kind == RFX_AGG_FIRST) {
            const u64 f = A.first[slot];
            v = (A.src && f != (u64)RFX_INF_I64_D) ? A.src[(i64)f - A.row0] : 0ULL;
        } else v = group_final(A.kind, A.f64, A.acc[slot], A.cnt ? A.cnt[slot] : 0ULL, A.skip);
        col[r] = v;
    }
}

static int upd_grid(rfx_ctx *c, i64 n) {
    const i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c) * 4;
    if (blocks < grid) grid = (int)blocks;
    return grid < 1 ? 1 : grid;
}

extern "C" int rfx_hip_update_set(rfx_ctx_t *c, void *d_col, const int64_t *d_ids, int64_t m, const void *d_vals, uint64_t atom_bits) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (m <= 0) return RFX_OK;
    RFX_REQUIRE(d_col, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_update_set, dim3(upd_grid(c, m)), dim3(RFX_BLOCK), 0, c->stream, (u64 *)d_col, (const i64 *)d_ids, (i64)m, (const u64 *)d_vals, (u64)atom_bits);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

extern "C" int rfx_hip_update_group(rfx_ctx_t *c, void *d_col, const int64_t *d_key, const int64_t *d_ids, int64_t m, const rfx_agg_t *agg,
                                    const rfx_group_tables_t *t) {
    RFX_REQUIRE(c && agg && t, RFX_EINVAL, "NULL argument");
    if (m <= 0) return RFX_OK;
    RFX_REQUIRE(d_col && d_key && t->range > 0 && t->nagg == 1 && t->d_first, RFX_EINVAL, "bad argument");
    UpdGroupArgs A;
    memset(&A, 0, sizeof(A));
    A.kmin = t->kmin;
    A.range = t->range;
    A.kind = agg->kind;
    A.f64 = rfx_agg_input_type(agg) == RFX_F64;
    A.skip = agg->xop != RFX_X_NONE || agg->nxnodes > 0;
    A.first = (const u64 *)t->d_first;
    A.acc = (const u64 *)t->d_acc[0];
    A.cnt = (const u64 *)t->d_cnt[0];
    A.src = (const u64 *)agg->d_col;
    hipLaunchKernelGGL(k_update_group, dim3(upd_grid(c, m)), dim3(RFX_BLOCK), 0, c->stream, (u64 *)d_col, (const i64 *)d_key, (const i64 *)d_ids, (i64)m, A);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- d_dst[0..n) = value: the virtual column of a parted table (one value per partition, core/vary.c:354-361) ----
__global__ __launch_bounds__(RFX_BLOCK) void k_fill_i64(i64 *p, i64 n, i64 val) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) p[i] = val;
}
extern "C" int rfx_hip_fill_i64(rfx_ctx_t *ctx, int64_t *d_dst, int64_t n, int64_t value) {
    rfx_ctx *c = (rfx_ctx *)ctx;
    if (!c || n < 0 || (n > 0 && !d_dst)) return RFX_EINVAL;
    if (n == 0) return RFX_OK;
    i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c) * 4;
    if (blocks < grid) grid = (int)blocks;
    hipLaunchKernelGGL(k_fill_i64, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (i64 *)d_dst, (i64)n, (i64)value);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- over SHARDS (round 6): one row-local write per shard, no ids ----
//   k_update_select   out[i] = (mask01 ? mask01[i] != 0 : 1) ? (vals ? vals[i] : atom) : (old ? old[i] : null_bits)
// mask01: the where: tree as an i64 column of 0 / 1 (rfx_hip_widen_b8: what the planner reads as one comparison too); vals: the mapping evaluated over the
// shard's rows (element-wise), or -- under by: -- every row's group aggregate looked up through the merged groups' value table.
__global__ __launch_bounds__(RFX_BLOCK) void k_update_select(u64 *__restrict__ out, const u64 *__restrict__ old, u64 null_bits, const i64 *__restrict__ mask01,
                                                             const u64 *__restrict__ vals, u64 atom, i64 n) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const bool sel = mask01 ? mask01[i] != 0 : true;
        out[i] = sel ? (vals ? vals[i] : atom) : (old ? old[i] : null_bits);
    }
}
extern "C" int rfx_hip_update_select(rfx_ctx_t *c, void *d_out, const void *d_old, uint64_t null_bits, const int64_t *d_mask01, const void *d_vals, uint64_t atom_bits,
                                     int64_t n) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_out, RFX_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_update_select, dim3(upd_grid(c, n)), dim3(RFX_BLOCK), 0, c->stream, (u64 *)d_out, (const u64 *)d_old, (u64)null_bits, (const i64 *)d_mask01,
                       (const u64 *)d_vals, (u64)atom_bits, (i64)n);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
