// rfx_dist.hip -- the multi-GPU exchange of the select / where / by path, in the C ABI: RCCL over xGMI, one process per GPU.
//
// Reference analogue: the path already merges per-chunk partial states -- AGGR_COLLECT (core/aggr.c:163-181) adds the per-thread
// group arrays element-wise, unop_fold's second level folds per-thread scalar partials (core/math.c:2206-2228), the sparse-key
// path re-inserts per-chunk tables (core/index.c:1866-1906).  Row-range sharding over GPUs is the same decomposition one level
// up: GPU g owns rows [row0_g, row0_g + n_g) of every column, nothing moves on the data path, and ONE exchange merges the states:
//
//   scalar aggregates   rfx_dist_partials_allgather   one ncclAllGather of (nagg + 1) x 64 B; folded in rank order on the host
//   key scope           rfx_dist_scope                one ncclAllGather of 24 B per rank (min, max, rows seen), folded on the host
//   dense group-by      rfx_dist_group_tables_allreduce  every table array with its own reduce op -- first: MIN (global row ids);
//                       sums / counts: SUM; min / max: MIN / MAX on the order-preserving i64 image -- issued inside ONE
//                       ncclGroupStart / ncclGroupEnd (one fused exchange, no host synchronisation), arrays of one (type, op) class that
//                       sit next to each other in memory merged into one call: `select sum(v) by k` is two calls, 2 x 8 MB at 1e6 keys
//   where ids / hashed tables / FIRST values   rfx_dist_allgather, rfx_dist_allreduce_i64
//
// RCCL is bound lazily (dlopen at rfx_dist_init): librfx.so carries no link-time dependency on it, single-GPU users never load it.
// The communicator's unique id travels through whatever side channel the host has (the reference's own IPC; torch.distributed's
// store in the Python host): rank 0 calls rfx_dist_unique_id, everybody rfx_dist_init with the same 128 bytes.
#include <dlfcn.h>
#include "rfx_group_common.hpp"

// the slice of rccl.h this file needs (RCCL keeps NCCL's C ABI)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

struct RcclApi {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi g_nccl;

static int rccl_bind(void) {
    if (g_nccl.lib) return RFX_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = NULL;
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        rfx_set_error("rfx_dist: cannot load librccl.so (%s)", dlerror());
        return RFX_ENODEV;
    }
#define RFX_SYM(field, name)                                                   \
    *(void **)(&g_nccl.field) = dlsym(h, name);                                \
    if (!g_nccl.field) {                                                       \
        rfx_set_error("rfx_dist: librccl.so lacks %s", name);                  \
        return RFX_ENODEV;                                                     \
    }
    RFX_SYM(GetUniqueId, "ncclGetUniqueId");
    RFX_SYM(CommInitRank, "ncclCommInitRank");
    RFX_SYM(CommInitAll, "ncclCommInitAll");
    RFX_SYM(CommCount, "ncclCommCount");
    RFX_SYM(CommUserRank, "ncclCommUserRank");
    RFX_SYM(CommDestroy, "ncclCommDestroy");
    RFX_SYM(AllReduce, "ncclAllReduce");
    RFX_SYM(AllGather, "ncclAllGather");
    RFX_SYM(GroupStart, "ncclGroupStart");
    RFX_SYM(GroupEnd, "ncclGroupEnd");
    RFX_SYM(GetErrorString, "ncclGetErrorString");
#undef RFX_SYM
    g_nccl.lib = h;
    return RFX_OK;
}

#define RFX_NCCL_CHECK(expr)                                                                                  \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            rfx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_nccl.GetErrorString(_r));           \
            return RFX_EHIP;                                                                                  \
        }                                                                                                     \
    } while (0)

extern "C" int rfx_dist_unique_id(void *id128) {
    RFX_REQUIRE(id128, RFX_EINVAL, "NULL argument");
    int rc = rccl_bind();
    if (rc != RFX_OK) return rc;
    ncclUniqueId id;
    RFX_NCCL_CHECK(g_nccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return RFX_OK;
}

extern "C" int rfx_dist_init(rfx_ctx_t *c, int world, int rank, const void *id128) {
    RFX_REQUIRE(c && id128, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(world >= 1 && rank >= 0 && rank < world, RFX_EINVAL, "rank / world out of range");
    RFX_REQUIRE(c->comm == NULL, RFX_ESTATE, "the context already has a communicator");
    int rc = rccl_bind();
    if (rc != RFX_OK) return rc;
    RFX_HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t comm = NULL;
    RFX_NCCL_CHECK(g_nccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm;
    c->world = world;
    c->rank = rank;
    return RFX_OK;
}

extern "C" int rfx_dist_finalize(rfx_ctx_t *c) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!c->comm) return RFX_OK;
    (void)hipStreamSynchronize(c->stream);
    RFX_NCCL_CHECK(g_nccl.CommDestroy((ncclComm_t)c->comm));
    c->comm = NULL;
    c->world = 0;
    c->rank = 0;
    c->ext_p[3] = NULL;
    return RFX_OK;
}

extern "C" int rfx_dist_world(rfx_ctx_t *c, int *world, int *rank) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (world) *world = 1;
    if (rank) *rank = 0;
    if (c->comm) { // what the COMMUNICATOR says (ncclCommCount / ncclCommUserRank), not what the caller passed to rfx_dist_init
        int w = 0, r = 0;
        RFX_NCCL_CHECK(g_nccl.CommCount((ncclComm_t)c->comm, &w));
        RFX_NCCL_CHECK(g_nccl.CommUserRank((ncclComm_t)c->comm, &r));
        if (world) *world = w;
        if (rank) *rank = r;
    }
    return RFX_OK;
}

static int dist_scratch(rfx_ctx *c, size_t bytes) { // device scratch for the small gathers
    if (c->dist_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_dist) RFX_HIP_CHECK(hipFree(c->d_dist));
    c->d_dist = NULL;
    c->dist_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_dist, bytes));
    c->dist_bytes = bytes;
    return RFX_OK;
}

// [min, max, rows seen] of every rank -> the global scope (a rank that saw no row is neutral).  (syncs)
extern "C" int rfx_dist_scope(rfx_ctx_t *c, int64_t *kmin, int64_t *kmax, int64_t *seen) {
    RFX_REQUIRE(c && kmin && kmax && seen, RFX_EINVAL, "NULL argument");
    if (!c->comm) return RFX_OK; // (a one-rank communicator still runs the exchange: that is how its fixed cost is measured)
    const int W = c->world;
    int rc = dist_scratch(c, (size_t)(W + 1) * 32);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE((size_t)(W + 1) * 32 <= c->pin_bytes, RFX_ELIMIT, "pinned staging too small");
    i64 *h = (i64 *)c->h_pin;
    h[0] = *kmin;
    h[1] = *kmax;
    h[2] = *seen;
    h[3] = 0;
    char *d = (char *)c->d_dist;
    RFX_HIP_CHECK(hipMemcpyAsync(d, h, 32, hipMemcpyHostToDevice, c->stream));
    RFX_NCCL_CHECK(g_nccl.AllGather(d, d + 32, 4, ncclInt64, (ncclComm_t)c->comm, c->stream));
    RFX_HIP_CHECK(hipMemcpyAsync(h, d + 32, (size_t)W * 32, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D, tot = 0;
    bool any = false, null_min = false, all_null = true;
    for (int r = 0; r < W; r++) {
        const i64 rmn = h[4 * r], rmx = h[4 * r + 1], rs = h[4 * r + 2];
        if (rs <= 0) continue;
        tot += rs;
        any = true;
        // index_scope_i64 treats a null key as the value INT64_MIN: a rank reporting min == NULL saw one
        if (rmn == RFX_NULL_I64_D) null_min = true;
        else mn = rmn < mn ? rmn : mn;
        if (rmx != RFX_NULL_I64_D) {
            all_null = false;
            mx = rmx > mx ? rmx : mx;
        }
    }
    *seen = tot;
    if (any) {
        *kmin = null_min ? RFX_NULL_I64_D : mn;
        *kmax = all_null ? RFX_NULL_I64_D : mx;
    }
    return RFX_OK;
}

// one table array's (element type, reduce op)
static void array_class(int kind, int f64, int what /* 0 first, 1 acc, 2 cnt */, ncclDataType_t *dt, ncclRedOp_t *op) {
    *dt = ncclInt64;
    *op = ncclSum;
    if (what == 0) { *op = ncclMin; return; }
    if (what == 2) return;
    if (kind == RFX_AGG_MIN) { *op = ncclMin; return; } // order-preserving i64 image for f64 columns too
    if (kind == RFX_AGG_MAX) { *op = ncclMax; return; }
    if (kind == RFX_AGG_AVG || (kind == RFX_AGG_SUM && f64)) *dt = ncclFloat64;
}

struct ArrCall {
    void *p;
    size_t n;
    ncclDataType_t dt;
    ncclRedOp_t op;
};
// neighbours of one class become one call; the calls are ENQUEUED only (the caller brackets them with ncclGroupStart / End)
static int allreduce_arrays_enqueue(rfx_ctx *c, ArrCall *a, int n) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (m > 0 && a[m - 1].dt == a[i].dt && a[m - 1].op == a[i].op && (char *)a[m - 1].p + a[m - 1].n * 8 == (char *)a[i].p) a[m - 1].n += a[i].n;
        else a[m++] = a[i];
    }
    for (int i = 0; i < m; i++) {
        ncclResult_t r = g_nccl.AllReduce(a[i].p, a[i].p, a[i].n, a[i].dt, a[i].op, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess) {
            rfx_set_error("rfx_dist: ncclAllReduce -> %s", g_nccl.GetErrorString(r));
            return RFX_EHIP;
        }
    }
    c->dist_calls += m;
    return RFX_OK;
}
static int allreduce_arrays(rfx_ctx *c, ArrCall *a, int n) {
    RFX_NCCL_CHECK(g_nccl.GroupStart());
    const int rc = allreduce_arrays_enqueue(c, a, n);
    if (rc != RFX_OK) {
        (void)g_nccl.GroupEnd();
        return rc;
    }
    RFX_NCCL_CHECK(g_nccl.GroupEnd());
    return RFX_OK;
}

// Dense group tables: in-place all-reduce of every array, one fused exchange, asynchronous on the context's stream.
static int tables_arrays(const rfx_agg_t *aggs, const rfx_group_tables_t *t, ArrCall *a) {
    int n = 0;
    a[n].p = t->d_first;
    a[n].n = (size_t)t->range;
    array_class(-1, 0, 0, &a[n].dt, &a[n].op);
    n++;
    for (int i = 0; i < t->nagg; i++) {
        const int f64 = rfx_agg_input_type(&aggs[i]) == RFX_F64;
        if (t->d_acc[i]) {
            a[n].p = t->d_acc[i];
            a[n].n = (size_t)t->range;
            array_class(aggs[i].kind, f64, 1, &a[n].dt, &a[n].op);
            n++;
        }
        if (t->d_cnt[i]) {
            a[n].p = t->d_cnt[i];
            a[n].n = (size_t)t->range;
            array_class(aggs[i].kind, f64, 2, &a[n].dt, &a[n].op);
            n++;
        }
    }
    return n;
}
extern "C" int rfx_dist_group_tables_allreduce(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *t) {
    RFX_REQUIRE(c && t && (aggs || t->nagg == 0), RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(t->nagg >= 0 && t->nagg <= RFX_MAX_AGGS && t->range > 0 && t->d_first, RFX_EINVAL, "bad tables");
    if (!c->comm) return RFX_OK;
    ArrCall a[1 + 2 * RFX_MAX_AGGS];
    return allreduce_arrays(c, a, tables_arrays(aggs, t, a));
}

// ---- ONE process, several devices (the evaluator process that owns all 8 GPUs of a node: rfx_exec.c) ----
// rfx_dist_init_all: one communicator per context, all created by this process (ncclCommInitAll); the contexts must sit on DISTINCT
// devices (RCCL refuses two ranks on one device: shards that share a device are merged by a kernel instead, rfx_hip_group_tables_merge).
// Such communicators are marked process-local: what the host can fold itself (scopes, scalar partials, flags) never goes through RCCL.
extern "C" int rfx_dist_init_all(rfx_ctx_t *const *ctxs, int n) {
    RFX_REQUIRE(ctxs && n >= 1 && n <= 64, RFX_EINVAL, "bad argument");
    int devs[64];
    for (int i = 0; i < n; i++) {
        RFX_REQUIRE(ctxs[i] && ctxs[i]->comm == NULL, RFX_ESTATE, "a context is NULL or already has a communicator");
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++) RFX_REQUIRE(devs[j] != devs[i], RFX_EINVAL, "two contexts on one device: RCCL takes one rank per device");
    }
    int rc = rccl_bind();
    if (rc != RFX_OK) return rc;
    ncclComm_t comms[64];
    RFX_NCCL_CHECK(g_nccl.CommInitAll(comms, n, devs));
    for (int i = 0; i < n; i++) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->world = n;
        ctxs[i]->rank = i;
        ctxs[i]->ext_p[3] = (void *)(uintptr_t)1; // process-local communicator
    }
    RFX_HIP_CHECK(hipSetDevice(ctxs[0]->device));
    return RFX_OK;
}
extern "C" int rfx_dist_is_local(rfx_ctx_t *c) { return c && c->comm && c->ext_p[3] ? 1 : 0; }
extern "C" int rfx_dist_has_comm(rfx_ctx_t *c) { return c && c->comm ? 1 : 0; }
// the table sets of n contexts (one per device, same range and aggregates), all-reduced in place in ONE fused exchange issued by this thread
extern "C" int rfx_dist_group_tables_allreduce_all(rfx_ctx_t *const *ctxs, int n, const rfx_agg_t *aggs, const rfx_group_tables_t *const *ts) {
    RFX_REQUIRE(ctxs && ts && n >= 1, RFX_EINVAL, "bad argument");
    if (n == 1) return rfx_dist_group_tables_allreduce(ctxs[0], aggs, ts[0]);
    RFX_NCCL_CHECK(g_nccl.GroupStart());
    int rc = RFX_OK;
    for (int i = 0; i < n && rc == RFX_OK; i++) {
        if (!ctxs[i]->comm) { rfx_set_error("rfx_dist: a context has no communicator"); rc = RFX_ESTATE; break; }
        ArrCall a[1 + 2 * RFX_MAX_AGGS];
        rc = allreduce_arrays_enqueue(ctxs[i], a, tables_arrays(aggs, ts[i], a));
    }
    if (rc != RFX_OK) {
        (void)g_nccl.GroupEnd();
        return rc;
    }
    RFX_NCCL_CHECK(g_nccl.GroupEnd());
    return RFX_OK;
}
// in-place all-reduce (op 0 SUM / 1 MIN / 2 MAX) of one buffer of n 8-byte integers per context; all-gather of `bytes` bytes per context
extern "C" int rfx_dist_allreduce_i64_all(rfx_ctx_t *const *ctxs, int n, int64_t *const *d_bufs, int64_t cells, int op) {
    RFX_REQUIRE(ctxs && d_bufs && n >= 1 && op >= 0 && op <= 2, RFX_EINVAL, "bad argument");
    if (cells == 0) return RFX_OK;
    RFX_NCCL_CHECK(g_nccl.GroupStart());
    for (int i = 0; i < n; i++) {
        ncclResult_t r = ctxs[i]->comm ? g_nccl.AllReduce(d_bufs[i], d_bufs[i], (size_t)cells, ncclInt64, op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax), (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream)
                                       : (ncclResult_t)1;
        if (r != ncclSuccess) {
            (void)g_nccl.GroupEnd();
            rfx_set_error("rfx_dist: ncclAllReduce (all) failed");
            return RFX_EHIP;
        }
        ctxs[i]->dist_calls += 1;
    }
    RFX_NCCL_CHECK(g_nccl.GroupEnd());
    return RFX_OK;
}
extern "C" int rfx_dist_allgather_all(rfx_ctx_t *const *ctxs, int n, const void *const *d_ins, size_t bytes, void *const *d_outs) {
    RFX_REQUIRE(ctxs && d_ins && d_outs && n >= 1, RFX_EINVAL, "bad argument");
    RFX_NCCL_CHECK(g_nccl.GroupStart());
    for (int i = 0; i < n; i++) {
        ncclResult_t r = ctxs[i]->comm ? g_nccl.AllGather(d_ins[i], d_outs[i], bytes, ncclInt8, (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream) : (ncclResult_t)1;
        if (r != ncclSuccess) {
            (void)g_nccl.GroupEnd();
            rfx_set_error("rfx_dist: ncclAllGather (all) failed");
            return RFX_EHIP;
        }
        ctxs[i]->dist_calls += 1;
    }
    RFX_NCCL_CHECK(g_nccl.GroupEnd());
    return RFX_OK;
}

// ---- shards that SHARE a device: merged by a kernel (what AGGR_COLLECT does with the per-thread arrays, core/aggr.c:163-181) ----
struct MergeArgs {
    i64 range;
    int n;
    int op[1 + 2 * RFX_MAX_AGGS]; // 0 add i64, 1 add f64, 2 min i64, 3 max i64
    u64 *into[1 + 2 * RFX_MAX_AGGS];
    const u64 *from[1 + 2 * RFX_MAX_AGGS];
};
__global__ __launch_bounds__(RFX_BLOCK) void k_tables_merge(const MergeArgs A) {
    const i64 stride = (i64)gridDim.x * RFX_BLOCK;
    for (i64 i = (i64)blockIdx.x * RFX_BLOCK + threadIdx.x; i < A.range; i += stride) {
#pragma unroll 1
        for (int k = 0; k < A.n; k++) {
            const u64 a = A.into[k][i], b = A.from[k][i];
            u64 r;
            switch (A.op[k]) {
                case 0: r = a + b; break;
                case 1: r = rfx_as_u64(rfx_as_f64(a) + rfx_as_f64(b)); break;
                case 2: r = (u64)((i64)a < (i64)b ? (i64)a : (i64)b); break;
                default: r = (u64)((i64)a > (i64)b ? (i64)a : (i64)b); break;
            }
            A.into[k][i] = r;
        }
    }
}
// into[slot] (op)= from[slot] for every array of two dense table sets over the same scope, on the context's device and stream
extern "C" int rfx_hip_group_tables_merge(rfx_ctx_t *c, const rfx_agg_t *aggs, const rfx_group_tables_t *into, const rfx_group_tables_t *from) {
    RFX_REQUIRE(c && into && from && (aggs || into->nagg == 0), RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(into->range == from->range && into->kmin == from->kmin && into->nagg == from->nagg && into->nagg <= RFX_MAX_AGGS, RFX_EINVAL, "table sets differ");
    if (into->range <= 0) return RFX_OK;
    ArrCall a[1 + 2 * RFX_MAX_AGGS], b[1 + 2 * RFX_MAX_AGGS];
    const int n = tables_arrays(aggs, into, a);
    RFX_REQUIRE(tables_arrays(aggs, from, b) == n, RFX_EINVAL, "table sets differ");
    MergeArgs M;
    M.range = into->range;
    M.n = n;
    for (int k = 0; k < n; k++) {
        M.op[k] = a[k].op == ncclMin ? 2 : (a[k].op == ncclMax ? 3 : (a[k].dt == ncclFloat64 ? 1 : 0));
        M.into[k] = (u64 *)a[k].p;
        M.from[k] = (const u64 *)b[k].p;
    }
    const i64 blocks = (into->range + RFX_BLOCK - 1) / RFX_BLOCK;
    hipLaunchKernelGGL(k_tables_merge, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(RFX_BLOCK), 0, c->stream, M);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
// into[i] += from[i] (8-byte integers): FIRST values, of which exactly one shard contributed each
__global__ __launch_bounds__(RFX_BLOCK) void k_add_i64(u64 *__restrict__ into, const u64 *__restrict__ from, i64 n) {
    const i64 stride = (i64)gridDim.x * RFX_BLOCK;
    for (i64 i = (i64)blockIdx.x * RFX_BLOCK + threadIdx.x; i < n; i += stride) into[i] += from[i];
}
extern "C" int rfx_hip_add_i64(rfx_ctx_t *c, int64_t *d_into, const int64_t *d_from, int64_t n) {
    RFX_REQUIRE(c && (n == 0 || (d_into && d_from)), RFX_EINVAL, "NULL argument");
    if (n <= 0) return RFX_OK;
    const i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    hipLaunchKernelGGL(k_add_i64, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(RFX_BLOCK), 0, c->stream, (u64 *)d_into, (const u64 *)d_from, (i64)n);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
// device-to-device on the context's stream; the source may live on another device of this process (hipMemcpyPeer semantics through
// the unified address space).  Asynchronous.
extern "C" int rfx_hip_d2d(rfx_ctx_t *c, void *d_dst, const void *d_src, size_t bytes) {
    RFX_REQUIRE(c && (bytes == 0 || (d_dst && d_src)), RFX_EINVAL, "NULL argument");
    if (!bytes) return RFX_OK;
    RFX_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDefault, c->stream));
    return RFX_OK;
}
// the calling thread issues this context's work from now on (hipSetDevice): every thread of a host that drives several devices calls
// it once before its first call on the context
extern "C" int rfx_hip_ctx_bind_thread(rfx_ctx_t *c) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_HIP_CHECK(hipSetDevice(c->device));
    return RFX_OK;
}
extern "C" int rfx_hip_ctx_device(rfx_ctx_t *c) { return c ? c->device : -1; }
// `bytes` of HOST memory from every rank, rank order, through the context's communicator (the small exchanges the host folds itself:
// scopes, scalar partials, flags).  Without a communicator: out = in.  (syncs)
extern "C" int rfx_dist_allgather_host(rfx_ctx_t *c, const void *in, size_t bytes, void *out) {
    RFX_REQUIRE(c && in && out && bytes > 0, RFX_EINVAL, "bad argument");
    if (!c->comm) {
        memcpy(out, in, bytes);
        return RFX_OK;
    }
    const int W = c->world;
    const size_t pad = (bytes + 15) & ~(size_t)15;
    int rc = dist_scratch(c, pad * (size_t)(W + 1));
    if (rc != RFX_OK) return rc;
    char *d = (char *)c->d_dist;
    RFX_HIP_CHECK(hipMemcpyAsync(d, in, bytes, hipMemcpyHostToDevice, c->stream));
    RFX_NCCL_CHECK(g_nccl.AllGather(d, d + pad, pad, ncclInt8, (ncclComm_t)c->comm, c->stream));
    c->dist_calls += 1;
    for (int r = 0; r < W; r++) RFX_HIP_CHECK(hipMemcpyAsync((char *)out + (size_t)r * bytes, d + pad + (size_t)r * pad, bytes, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RFX_OK;
}

// Scalar partials: d_all[r * n .. (r + 1) * n) = rank r's d_local[0 .. n).  Asynchronous.  d_all may be NULL: context scratch is
// used and rfx_dist_partials_host copies it out.
extern "C" int rfx_dist_partials_allgather(rfx_ctx_t *c, const rfx_partial_t *d_local, int n, rfx_partial_t *d_all) {
    RFX_REQUIRE(c && d_local && d_all && n > 0, RFX_EINVAL, "bad argument");
    if (!c->comm) {
        RFX_HIP_CHECK(hipMemcpyAsync(d_all, d_local, (size_t)n * sizeof(rfx_partial_t), hipMemcpyDeviceToDevice, c->stream));
        return RFX_OK;
    }
    RFX_NCCL_CHECK(g_nccl.AllGather(d_local, d_all, (size_t)n * sizeof(rfx_partial_t), ncclInt8, (ncclComm_t)c->comm, c->stream));
    c->dist_calls += 1;
    return RFX_OK;
}

// Filter -> scalar aggregates over the SHARDED table: the local fused pass, one all-gather, the rank-ordered fold.  (syncs)
extern "C" int rfx_dist_filter_aggr_host(rfx_ctx_t *c, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg, int64_t nrows,
                                         int64_t row0, rfx_value_t *values, int64_t *selected) {
    RFX_REQUIRE(c && aggs && values && nagg >= 1 && nagg <= RFX_MAX_AGGS, RFX_EINVAL, "bad argument");
    const int W = c->comm ? c->world : 1;
    const size_t one = (size_t)(nagg + 1) * sizeof(rfx_partial_t);
    int rc = dist_scratch(c, one * (size_t)(W + 1));
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(one * (size_t)W <= c->pin_bytes, RFX_ELIMIT, "pinned staging too small");
    rfx_partial_t *d_loc = (rfx_partial_t *)c->d_dist, *d_all = d_loc + (nagg + 1);
    rc = rfx_hip_filter_aggr(c, preds, npred, logic, aggs, nagg, nrows, row0, d_loc);
    if (rc != RFX_OK) return rc;
    rc = rfx_dist_partials_allgather(c, d_loc, nagg + 1, d_all);
    if (rc != RFX_OK) return rc;
    rfx_partial_t *h = (rfx_partial_t *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, d_all, one * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int r = 1; r < W; r++) {
        for (int a = 0; a < nagg; a++) rfx_partial_merge(aggs[a].kind, rfx_agg_input_type(&aggs[a]), &h[a], &h[(size_t)r * (nagg + 1) + a]);
        rfx_partial_merge(RFX_AGG_COUNT, RFX_I64, &h[nagg], &h[(size_t)r * (nagg + 1) + nagg]);
    }
    for (int a = 0; a < nagg; a++) {
        rc = rfx_agg_finalize(aggs[a].kind, rfx_agg_input_type(&aggs[a]), &h[a], &values[a]);
        if (rc != RFX_OK) return rc;
    }
    if (selected) *selected = h[nagg].cnt;
    return RFX_OK;
}

// In-place all-reduce of n 8-byte integers: op 0 SUM (FIRST values: one owner per group), 1 MIN, 2 MAX (per-rank flags).  Asynchronous.
extern "C" int rfx_dist_allreduce_i64(rfx_ctx_t *c, int64_t *d_buf, int64_t n, int op) {
    RFX_REQUIRE(c && (d_buf || n == 0) && n >= 0 && op >= 0 && op <= 2, RFX_EINVAL, "bad argument");
    if (!c->comm || n == 0) return RFX_OK;
    RFX_NCCL_CHECK(g_nccl.AllReduce(d_buf, d_buf, (size_t)n, ncclInt64, op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax), (ncclComm_t)c->comm, c->stream));
    c->dist_calls += 1;
    return RFX_OK;
}

// d_out[r * bytes .. (r + 1) * bytes) = rank r's d_in[0 .. bytes)  (hashed table sets, padded id lists).  Asynchronous.
extern "C" int rfx_dist_allgather(rfx_ctx_t *c, const void *d_in, size_t bytes, void *d_out) {
    RFX_REQUIRE(c && d_in && d_out, RFX_EINVAL, "NULL argument");
    if (!c->comm) {
        if (d_in != d_out) RFX_HIP_CHECK(hipMemcpyAsync(d_out, d_in, bytes, hipMemcpyDeviceToDevice, c->stream));
        return RFX_OK;
    }
    RFX_NCCL_CHECK(g_nccl.AllGather(d_in, d_out, bytes, ncclInt8, (ncclComm_t)c->comm, c->stream));
    c->dist_calls += 1;
    return RFX_OK;
}

// collectives issued so far (tests: "a dense group-by is two calls")
extern "C" int64_t rfx_dist_calls(rfx_ctx_t *c) { return c ? c->dist_calls : 0; }
