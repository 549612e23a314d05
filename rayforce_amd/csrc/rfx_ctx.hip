// rfx_ctx.hip -- context, device memory, timers, synthetic-column generator, plan builder.
#include <pthread.h>
#include "rfx_common.hpp"
#include <stdarg.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void rfx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rfx_hip_last_error(void) { return g_err; }
extern "C" const char *rfx_hip_version(void) { return "rfx-hip 0.1 (gfx950)"; }

extern "C" int rfx_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int rfx_hip_ctx_create(int device, void *stream, rfx_ctx_t **out) {
    RFX_REQUIRE(out != NULL, RFX_EINVAL, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        rfx_set_error("no HIP device visible (hipGetDeviceCount failed or returned 0) -- the MI355X path has no CPU fallback");
        return RFX_ENODEV;
    }
    RFX_REQUIRE(device >= 0 && device < n, RFX_EINVAL, "device ordinal out of range");
    RFX_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    RFX_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    rfx_ctx *c = (rfx_ctx *)calloc(1, sizeof(rfx_ctx));
    RFX_REQUIRE(c != NULL, RFX_ENOMEM, "host calloc failed");
    c->device = device;
    c->ext_p[7] = calloc(1, sizeof(CtxExt));
    RFX_REQUIRE(c->ext_p[7] != NULL, RFX_ENOMEM, "host calloc failed");
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->blocks_per_cu = 2; // tools/probe_hw: 2 workgroups per CU streams fastest (7.0 TB/s)
    if (stream) {
        c->stream = (stream == RFX_STREAM_LEGACY) ? (hipStream_t)0 : (hipStream_t)stream;
        c->own_stream = false;
    } else {
        RFX_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    RFX_HIP_CHECK(hipEventCreate(&c->ev0));
    RFX_HIP_CHECK(hipEventCreate(&c->ev1));
    for (int i = 0; i < 8; i++) {
        RFX_HIP_CHECK(hipEventCreate(&c->evk[i][0]));
        RFX_HIP_CHECK(hipEventCreate(&c->evk[i][1]));
    }
    c->pin_bytes = 1 << 16;
    RFX_HIP_CHECK(hipHostMalloc(&c->h_pin, c->pin_bytes, hipHostMallocDefault));
    int rc = rfx_ws_reserve(c, 4u << 20);
    if (rc != RFX_OK) return rc;
    *out = c;
    return RFX_OK;
}

static void pool_release(rfx_ctx *c);
void rfx_plane_release(rfx_ctx *c); // rfx_group_plane.hip
void rfx_plane_invalidate(rfx_ctx *c);
extern "C" int rfx_hip_ctx_destroy(rfx_ctx_t *c) {
    if (!c) return RFX_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)rfx_dist_finalize(c);
    if (c->d_dist) (void)hipFree(c->d_dist);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_bitmap) (void)hipFree(c->d_bitmap);
    if (c->d_blksum) (void)hipFree(c->d_blksum);
    if (c->d_gid) (void)hipFree(c->d_gid);
    if (c->d_part) (void)hipFree(c->d_part);
    if (c->d_comp) (void)hipFree(c->d_comp);
    if (c->d_sel) (void)hipFree(c->d_sel);
    if (c->d_expr) (void)hipFree(c->d_expr);
    if (c->d_chunk) (void)hipFree(c->d_chunk);
    rfx_io_release(c);
    pool_release(c);
    if (c->ext_p[1]) (void)hipFree(c->ext_p[1]);
    free(c->ext_p[7]); // CtxExt
    free(c->ext_p[5]); // rfx_chunk_scope's sample memo
    free(c->ext_p[6]); // rfx_hip_where_estimate's
    rfx_plane_release(c);
    if (c->d_pc_counts) (void)hipFree(c->d_pc_counts);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < 8; i++) {
        (void)hipEventDestroy(c->evk[i][0]);
        (void)hipEventDestroy(c->evk[i][1]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    free(c);
    return RFX_OK;
}

extern "C" int64_t rfx_hip_ctx_stat(rfx_ctx_t *c, int which) {
    if (!c || which < 0 || which > 6) return -1;
    if (which == 6) return (int64_t)(uintptr_t)c->ext_p[4]; // RFX_STAT_WHERE_ONCE: k_where_once launches (rfx_where_once.hip)
    return which == RFX_STAT_MASK_PASSES ? c->ext_i[0] : c->ext_i[3 + which];
}

extern "C" int rfx_hip_ctx_sync(rfx_ctx_t *c) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RFX_OK;
}

extern "C" int rfx_hip_ctx_set_stream(rfx_ctx_t *c, void *stream) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(stream, RFX_EINVAL, "stream is NULL");
    if (c->own_stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    }
    c->stream = (stream == RFX_STREAM_LEGACY) ? (hipStream_t)0 : (hipStream_t)stream;
    return RFX_OK;
}

extern "C" int rfx_hip_ctx_tune(rfx_ctx_t *c, int blocks_per_cu, int flags) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (blocks_per_cu > 0) {
        RFX_REQUIRE(blocks_per_cu <= 64, RFX_EINVAL, "blocks_per_cu > 64");
        c->blocks_per_cu = blocks_per_cu;
    }
    c->flags = flags;
    return RFX_OK;
}

int rfx_ws_reserve(rfx_ctx *c, size_t bytes) {
    if (c->ws_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_ws) RFX_HIP_CHECK(hipFree(c->d_ws));
    c->d_ws = NULL;
    c->ws_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_ws, bytes));
    c->ws_bytes = bytes;
    return RFX_OK;
}

int rfx_bitmap_reserve(rfx_ctx *c, i64 nrows) {
    size_t words = (size_t)((nrows + 63) / 64) + 64;
    size_t blocks = (size_t)((nrows + 2047) / 2048) + 2;
    if (c->bitmap_cap < words) {
        RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_bitmap) RFX_HIP_CHECK(hipFree(c->d_bitmap));
        c->d_bitmap = NULL;
        c->bitmap_cap = 0;
        RFX_HIP_CHECK(hipMalloc((void **)&c->d_bitmap, words * 8));
        c->bitmap_cap = words;
    }
    if (c->blksum_cap < blocks) {
        RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_blksum) RFX_HIP_CHECK(hipFree(c->d_blksum));
        c->d_blksum = NULL;
        c->blksum_cap = 0;
        RFX_HIP_CHECK(hipMalloc((void **)&c->d_blksum, blocks * 8));
        c->blksum_cap = blocks;
    }
    return RFX_OK;
}

int rfx_gid_reserve(rfx_ctx *c, i64 slots) {
    if (c->gid_cap >= (size_t)slots) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_gid) RFX_HIP_CHECK(hipFree(c->d_gid));
    c->d_gid = NULL;
    c->gid_cap = 0;
    RFX_HIP_CHECK(hipMalloc((void **)&c->d_gid, (size_t)slots * 8));
    c->gid_cap = (size_t)slots;
    return RFX_OK;
}

int rfx_part_reserve(rfx_ctx *c, size_t bytes) {
    if (c->part_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_part) RFX_HIP_CHECK(hipFree(c->d_part));
    c->d_part = NULL;
    c->part_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_part, bytes));
    c->part_bytes = bytes;
    return RFX_OK;
}

int rfx_expr_reserve(rfx_ctx *c, size_t bytes) {
    if (c->expr_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_expr) RFX_HIP_CHECK(hipFree(c->d_expr));
    c->d_expr = NULL;
    c->expr_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_expr, bytes));
    c->expr_bytes = bytes;
    return RFX_OK;
}

int rfx_sel_reserve(rfx_ctx *c, size_t bytes) {
    if (c->sel_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_sel) RFX_HIP_CHECK(hipFree(c->d_sel));
    c->d_sel = NULL;
    c->sel_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_sel, bytes));
    c->sel_bytes = bytes;
    return RFX_OK;
}

int rfx_chunk_reserve(rfx_ctx *c, size_t bytes) {
    if (c->chunk_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_chunk) RFX_HIP_CHECK(hipFree(c->d_chunk));
    c->d_chunk = NULL;
    c->chunk_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_chunk, bytes));
    c->chunk_bytes = bytes;
    return RFX_OK;
}

int rfx_comp_reserve(rfx_ctx *c, size_t bytes) {
    if (c->comp_bytes >= bytes) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_comp) RFX_HIP_CHECK(hipFree(c->d_comp));
    c->d_comp = NULL;
    c->comp_bytes = 0;
    RFX_HIP_CHECK(hipMalloc(&c->d_comp, bytes));
    c->comp_bytes = bytes;
    return RFX_OK;
}

// ---- expressions as plain columns (for the kernels that do not fold them on the fly) ----
struct DeriveArgs {
    PlanExpr x;
    const u64 *cols[RFX_MAX_COLS];
    u64 *out;
};
__device__ __forceinline__ u64 derive_operand(int kind, u64 colval, u64 atom, int idx, const u64 (&res)[RFX_MAX_XNODES]) {
    if (kind == RFX_XK_COL) return colval;
    if (kind == RFX_XK_ATOM) return atom;
    return res[idx];
}
// Two rows per thread: every column operand of every operation is fetched with one 16-byte load (coalesced 1 KB per wave).
__global__ __launch_bounds__(RFX_BLOCK) void k_derive(const DeriveArgs A, i64 nrows) {
    const i64 npairs = nrows / 2;
    const int nops = A.x.nops;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < npairs; i += (i64)gridDim.x * RFX_BLOCK) {
        u64 r0[RFX_MAX_XNODES] = {0, 0, 0, 0}, r1[RFX_MAX_XNODES] = {0, 0, 0, 0};
        for (int k = 0; k < nops; k++) {
            const PlanXNode n = A.x.ops[k];
            u64x2 lc, rc;
            lc.x = lc.y = rc.x = rc.y = 0;
            if (n.l_kind == RFX_XK_COL) lc = rfx_ld2(A.cols[n.l_idx] + 2 * i);
            if (n.r_kind == RFX_XK_COL) rc = rfx_ld2(A.cols[n.r_idx] + 2 * i);
            r0[k] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, derive_operand(n.l_kind, lc.x, n.l_atom, n.l_idx, r0),
                                  derive_operand(n.r_kind, rc.x, n.r_atom, n.r_idx, r0));
            r1[k] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, derive_operand(n.l_kind, lc.y, n.l_atom, n.l_idx, r1),
                                  derive_operand(n.r_kind, rc.y, n.r_atom, n.r_idx, r1));
        }
        u64x2 o;
        o.x = r0[nops - 1];
        o.y = r1[nops - 1];
        *(u64x2 *)(A.out + 2 * i) = o;
    }
    if ((nrows & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const i64 i = nrows - 1;
        u64 res[RFX_MAX_XNODES] = {0, 0, 0, 0};
        for (int k = 0; k < nops; k++) {
            const PlanXNode n = A.x.ops[k];
            const u64 l = derive_operand(n.l_kind, n.l_kind == RFX_XK_COL ? A.cols[n.l_idx][i] : 0, n.l_atom, n.l_idx, res);
            const u64 r = derive_operand(n.r_kind, n.r_kind == RFX_XK_COL ? A.cols[n.r_idx][i] : 0, n.r_atom, n.r_idx, res);
            res[k] = rfx_expr_eval(n.op, n.o_f64, n.l_f64, n.r_f64, l, r);
        }
        A.out[i] = res[nops - 1];
    }
}

int rfx_plan_materialise_exprs(rfx_ctx *c, Plan *P) {
    if (P->nx == 0) return RFX_OK;
    RFX_REQUIRE(P->ncols + P->nx <= RFX_MAX_COLS, RFX_ELIMIT, "too many distinct columns once the expressions are materialised");
    const size_t col_bytes = (((size_t)P->nrows * 8) + 255) & ~(size_t)255;
    int rc = rfx_expr_reserve(c, col_bytes * (size_t)P->nx);
    if (rc != RFX_OK) return rc;
    int newcol[RFX_MAX_EXPRS];
    for (int i = 0; i < P->nx; i++) {
        DeriveArgs A;
        A.x = P->xs[i];
        for (int cc = 0; cc < RFX_MAX_COLS; cc++) A.cols[cc] = (cc < P->ncols) ? P->cols[cc] : NULL;
        A.out = (u64 *)((char *)c->d_expr + col_bytes * (size_t)i);
        hipLaunchKernelGGL(k_derive, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, A, P->nrows);
        RFX_HIP_CHECK(hipGetLastError());
        newcol[i] = P->ncols + i;
    }
    for (int i = 0; i < P->nx; i++) P->cols[P->ncols + i] = (const u64 *)((char *)c->d_expr + col_bytes * (size_t)i);
    P->ncols += P->nx;
    for (int a = 0; a < P->nagg; a++)
        if (P->aggs[a].col >= RFX_XCOL) P->aggs[a].col = newcol[P->aggs[a].col - RFX_XCOL];
    P->nx = 0;
    return RFX_OK;
}

// Element-wise evaluation into a caller's column: binop_map / ray_{add,sub,mul,div}_partial (core/math.c:251-1782,2280-2345) as one
// streaming pass -- every operand column read once with 16-byte loads, the result written once, no intermediate per operation.
extern "C" int rfx_hip_eval_expr(rfx_ctx_t *c, const rfx_agg_t *expr, int64_t nrows, void *d_out, int32_t *out_type) {
    RFX_REQUIRE(c && expr, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(expr->xop != RFX_X_NONE || expr->nxnodes > 0, RFX_EINVAL, "not an expression");
    rfx_agg_t a = *expr;
    a.kind = RFX_AGG_SUM; // any kind that accepts an expression: only the expression part of the plan is used
    Plan P;
    int rc = rfx_plan_build(&P, NULL, 0, RFX_AND, &a, 1, NULL, NULL, nrows, 0);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(P.nx == 1, RFX_EINVAL, "not an expression");
    if (out_type) *out_type = P.xs[0].out_f64 ? RFX_F64 : RFX_I64;
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_out && ((uintptr_t)d_out & 15) == 0, RFX_EINVAL, "output must be a 16-byte aligned device buffer");
    DeriveArgs A;
    A.x = P.xs[0];
    for (int cc = 0; cc < RFX_MAX_COLS; cc++) A.cols[cc] = (cc < P.ncols) ? P.cols[cc] : NULL;
    A.out = (u64 *)d_out;
    hipLaunchKernelGGL(k_derive, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, A, (i64)nrows);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- plain memory ----
// Small device blocks (<= 4 MB: group tables, result blocks, masks of small queries) are recycled inside the context: hipMalloc +
// hipFree cost tens of microseconds and hipFree waits for the device -- more than the whole rest of a 1e6-row query.  Blocks are
// power-of-two sized; everything that touches them runs on the context's one stream, so a recycled block is never read by work
// still in flight.  One process-wide lock guards the pools (contexts that share a device trim each other's idle blocks when memory runs out).
#define POOL_MAX_LOG 28 /* blocks up to 256 MB are recycled (group tables and result blocks of a 1e6-group query are 16 MB each: a hipMalloc / hipFree pair
                         * per query cost more than the query's kernels at 1e8 rows) */
#define POOL_MIN_LOG 8
#define POOL_KEEP 8
#define POOL_LIVE 1024
#define BIG_KEEP 12                          /* blocks above 256 MB kept for reuse ... */
#define BIG_KEEP_BYTES ((size_t)40 << 30)    /* ... up to this much in all (of 288 GB); rfx_hip_ctx_trim gives them back */
#define BIG_LIVE 64
struct SmallPool {
    void *live_p[POOL_LIVE];
    unsigned char live_c[POOL_LIVE];
    int nlive;
    void *freep[POOL_MAX_LOG + 1][POOL_KEEP];
    int nfree[POOL_MAX_LOG + 1];
    // Blocks above 256 MB (the tables, result columns and id vectors of queries over 1e8+ groups / ids): hipMalloc + hipFree of gigabytes
    // cost tens of milliseconds each (page-table work) -- the six-key row-hash query spent 0.6 s of its 0.65 s there.  Sized in 64 MB
    // steps, kept when freed, handed out again to a request they fit with at most a quarter to spare.
    void *big_live_p[BIG_LIVE];
    size_t big_live_b[BIG_LIVE];
    int nbig_live;
    void *big_free_p[BIG_KEEP];
    size_t big_free_b[BIG_KEEP];
    int nbig_free;
    size_t big_free_total;
};
// Several contexts may share a device (RFX_SHARDS=k, the Python host's Engine(shards=k)), each driven by its own host thread: ONE lock
// around every pool operation of the process, a registry of the live contexts, and the kept-for-reuse budget counted PER DEVICE -- so
// that a context out of memory can give back what its siblings keep idle instead of failing beside tens of idle gigabytes.
#define POOL_MAX_DEV 64
#define POOL_MAX_CTX 256
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static rfx_ctx *g_pool_ctx[POOL_MAX_CTX];
static int g_pool_nctx;
static size_t g_dev_kept[POOL_MAX_DEV]; // bytes of big blocks kept for reuse on a device, all contexts
static void pool_register(rfx_ctx *c) {
    for (int i = 0; i < g_pool_nctx; i++)
        if (g_pool_ctx[i] == c) return;
    if (g_pool_nctx < POOL_MAX_CTX) g_pool_ctx[g_pool_nctx++] = c;
}
// every block `c` keeps for reuse goes back to the device (the lock is held)
static void pool_drop_kept(rfx_ctx *c, int small_too) {
    SmallPool *sp = (SmallPool *)c->ext_p[0];
    if (!sp) return;
    if (small_too)
        for (int k = 0; k <= POOL_MAX_LOG; k++) {
            for (int i = 0; i < sp->nfree[k]; i++) (void)hipFree(sp->freep[k][i]);
            sp->nfree[k] = 0;
        }
    for (int i = 0; i < sp->nbig_free; i++) (void)hipFree(sp->big_free_p[i]);
    if (c->device >= 0 && c->device < POOL_MAX_DEV) g_dev_kept[c->device] -= sp->big_free_total < g_dev_kept[c->device] ? sp->big_free_total : g_dev_kept[c->device];
    sp->nbig_free = 0;
    sp->big_free_total = 0;
}
// out of memory on c's device: what c and every sibling context on that device keep idle goes back (hipFree waits for the device, so
// nothing in flight on a sibling's stream still reads a block that sat in its free list)
static void pool_trim_device(rfx_ctx *c) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    for (int i = 0; i < g_pool_nctx; i++)
        if (g_pool_ctx[i]->device == c->device) pool_drop_kept(g_pool_ctx[i], 1);
}
static void pool_release(rfx_ctx *c) {
    pthread_mutex_lock(&g_pool_mu);
    for (int i = 0; i < g_pool_nctx; i++)
        if (g_pool_ctx[i] == c) { g_pool_ctx[i] = g_pool_ctx[--g_pool_nctx]; break; }
    SmallPool *sp = (SmallPool *)c->ext_p[0];
    if (sp) {
        pool_drop_kept(c, 1);
        free(sp);
        c->ext_p[0] = NULL;
    }
    pthread_mutex_unlock(&g_pool_mu);
}
// a fresh block from the device.  Out of memory: what this context AND its siblings on the device keep idle goes back first (pool_trim_device:
// hipDeviceSynchronize + hipFree under the process-wide lock -- every allocation of the process waits meanwhile; an out-of-memory path, not a
// steady state).
static hipError_t pool_dev_malloc(rfx_ctx *c, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        pool_trim_device(c);
        e = hipMalloc(p, bytes);
    }
    return e;
}
static int pool_malloc_locked(rfx_ctx *c, void **d_ptr, size_t bytes) {
    SmallPool *sp = (SmallPool *)c->ext_p[0];
    if (!sp) {
        c->ext_p[0] = sp = (SmallPool *)calloc(1, sizeof(SmallPool));
        pool_register(c);
    }
    // An allocation leaves the calling thread on the context's device (as rfx_hip_ctx_bind_thread does): callers allocate right before they launch
    // on the context's stream.  hipSetDevice only when the thread is somewhere else -- a pooled hit on the right device touches no HIP state.
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != c->device) RFX_HIP_CHECK(hipSetDevice(c->device));
    }
    if (bytes <= ((size_t)1 << POOL_MAX_LOG)) {
        if (sp && sp->nlive < POOL_LIVE) {
            int k = POOL_MIN_LOG;
            while (((size_t)1 << k) < bytes) k++;
            void *p = NULL;
            if (sp->nfree[k] > 0) p = sp->freep[k][--sp->nfree[k]];
            else {
                RFX_HIP_CHECK(pool_dev_malloc(c, &p, (size_t)1 << k));
            }
            sp->live_p[sp->nlive] = p;
            sp->live_c[sp->nlive++] = (unsigned char)k;
            *d_ptr = p;
            return RFX_OK;
        }
        // (the table of live small blocks is full: an untracked block of the size asked for -- not one of the 64 big slots, not 64 MB)
        RFX_HIP_CHECK(pool_dev_malloc(c, d_ptr, bytes ? bytes : 8));
        return RFX_OK;
    }
    const size_t want = (bytes + (((size_t)64 << 20) - 1)) & ~(((size_t)64 << 20) - 1);
    if (sp && sp->nbig_live < BIG_LIVE) {
        int best = -1;
        for (int i = 0; i < sp->nbig_free; i++)
            if (sp->big_free_b[i] >= want && sp->big_free_b[i] <= want + want / 4 && (best < 0 || sp->big_free_b[i] < sp->big_free_b[best])) best = i;
        void *p = NULL;
        size_t got = want;
        if (best >= 0) {
            p = sp->big_free_p[best];
            got = sp->big_free_b[best];
            sp->big_free_total -= got;
            if (c->device >= 0 && c->device < POOL_MAX_DEV) g_dev_kept[c->device] -= got < g_dev_kept[c->device] ? got : g_dev_kept[c->device];
            sp->big_free_p[best] = sp->big_free_p[sp->nbig_free - 1];
            sp->big_free_b[best] = sp->big_free_b[--sp->nbig_free];
        } else {
            RFX_HIP_CHECK(pool_dev_malloc(c, &p, want));
        }
        sp->big_live_p[sp->nbig_live] = p;
        sp->big_live_b[sp->nbig_live++] = got;
        *d_ptr = p;
        return RFX_OK;
    }
    RFX_HIP_CHECK(pool_dev_malloc(c, d_ptr, bytes));
    return RFX_OK;
}
extern "C" int rfx_hip_malloc(rfx_ctx_t *c, void **d_ptr, size_t bytes) {
    RFX_REQUIRE(c && d_ptr, RFX_EINVAL, "NULL argument");
    pthread_mutex_lock(&g_pool_mu);
    const int rc = pool_malloc_locked(c, d_ptr, bytes);
    pthread_mutex_unlock(&g_pool_mu);
    return rc;
}
// give every block kept for reuse back to the device (a host about to need the memory for something else)
extern "C" int rfx_hip_ctx_trim(rfx_ctx_t *c) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!c->ext_p[0]) return RFX_OK;
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    pthread_mutex_lock(&g_pool_mu);
    pool_drop_kept(c, 1);
    pthread_mutex_unlock(&g_pool_mu);
    return RFX_OK;
}
static int pool_free_locked(rfx_ctx *c, void *d_ptr) {
    SmallPool *sp = (SmallPool *)c->ext_p[0];
    if (sp) {
        for (int i = sp->nlive - 1; i >= 0; i--)
            if (sp->live_p[i] == d_ptr) {
                const int k = sp->live_c[i];
                sp->live_p[i] = sp->live_p[sp->nlive - 1];
                sp->live_c[i] = sp->live_c[--sp->nlive];
                if (sp->nfree[k] < (k > 22 ? 2 : POOL_KEEP)) { // (two spares per size class above 4 MB)
                    sp->freep[k][sp->nfree[k]++] = d_ptr;
                    return RFX_OK;
                }
                break;
            }
        for (int i = sp->nbig_live - 1; i >= 0; i--)
            if (sp->big_live_p[i] == d_ptr) {
                const size_t b = sp->big_live_b[i];
                sp->big_live_p[i] = sp->big_live_p[sp->nbig_live - 1];
                sp->big_live_b[i] = sp->big_live_b[--sp->nbig_live];
                const int dv = (c->device >= 0 && c->device < POOL_MAX_DEV) ? c->device : 0;
                if (sp->nbig_free < BIG_KEEP && g_dev_kept[dv] + b <= BIG_KEEP_BYTES) { // (the budget is the DEVICE's, whatever the number of contexts on it)
                    sp->big_free_p[sp->nbig_free] = d_ptr;
                    sp->big_free_b[sp->nbig_free++] = b;
                    sp->big_free_total += b;
                    g_dev_kept[dv] += b;
                    return RFX_OK;
                }
                break;
            }
    }
    RFX_HIP_CHECK(hipFree(d_ptr));
    return RFX_OK;
}
extern "C" int rfx_hip_free(rfx_ctx_t *c, void *d_ptr) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!d_ptr) return RFX_OK;
    pthread_mutex_lock(&g_pool_mu);
    const int rc = pool_free_locked(c, d_ptr);
    pthread_mutex_unlock(&g_pool_mu);
    return rc;
}
extern "C" int rfx_hip_h2d(rfx_ctx_t *c, void *d_dst, const void *src, size_t bytes) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!bytes) return RFX_OK;
    // device data changed under whatever a scope pass left partitioned (the residency cache refreshes stale columns IN PLACE, at the same
    // device address): partitions are matched by pointers and sizes only, so none survives an upload
    c->ck_valid = 0;
    c->pc_valid = 0;
    rfx_plane_invalidate(c);
    if (c->ext_p[5]) memset(c->ext_p[5], 0, 256); // ... nor does a remembered sample (rfx_chunk_scope; it only sizes and routes, but why keep it)
    if (c->ext_p[6]) memset(c->ext_p[6], 0, 256); // ... or a remembered `where` estimate
    RFX_HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RFX_OK;
}
extern "C" int rfx_hip_d2h(rfx_ctx_t *c, void *dst, const void *d_src, size_t bytes) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!bytes) return RFX_OK;
    RFX_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RFX_OK;
}
extern "C" int rfx_hip_d2h_async(rfx_ctx_t *c, void *dst, const void *d_src, size_t bytes) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!bytes) return RFX_OK;
    RFX_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    return RFX_OK;
}
extern "C" int rfx_hip_memset(rfx_ctx_t *c, void *d_dst, int byte, size_t bytes) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!bytes) return RFX_OK;
    c->ck_valid = 0;
    c->pc_valid = 0;
    rfx_plane_invalidate(c);
    if (c->ext_p[5]) memset(c->ext_p[5], 0, 256);
    if (c->ext_p[6]) memset(c->ext_p[6], 0, 256);
    RFX_HIP_CHECK(hipMemsetAsync(d_dst, byte, bytes, c->stream));
    return RFX_OK;
}

// ---- timers ----
extern "C" int rfx_hip_timer_start(rfx_ctx_t *c) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    return RFX_OK;
}
extern "C" int rfx_hip_timer_stop(rfx_ctx_t *c, float *ms) {
    RFX_REQUIRE(c && ms, RFX_EINVAL, "NULL argument");
    RFX_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    RFX_HIP_CHECK(hipEventSynchronize(c->ev1));
    RFX_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return RFX_OK;
}

extern "C" int rfx_hip_ctx_profile(rfx_ctx_t *c, int enable) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    c->profile = enable ? 1 : 0;
    c->evk_valid = 0;
    c->evk_n = 0;
    return RFX_OK;
}
// Every bracketed kernel since the last call of this function (or since profiling was switched on), in launch order: their HIP-event durations
// on the context's stream.  At most the last eight are kept; *n = how many were written.  (syncs)
extern "C" int rfx_hip_profile_kernels(rfx_ctx_t *c, float *ms, int cap, int *n) {
    RFX_REQUIRE(c && ms && n && cap >= 0, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(c->profile, RFX_ESTATE, "profiling is off (call rfx_hip_ctx_profile(ctx, 1) first)");
    const int have = c->evk_n < 8 ? c->evk_n : 8, first = c->evk_n - have;
    int w = 0;
    for (int i = 0; i < have && w < cap; i++, w++) {
        hipEvent_t *p = c->evk[(first + i) & 7];
        RFX_HIP_CHECK(hipEventSynchronize(p[1]));
        RFX_HIP_CHECK(hipEventElapsedTime(&ms[w], p[0], p[1]));
    }
    *n = w;
    c->evk_n = 0;
    c->evk_valid = 0;
    return RFX_OK;
}
extern "C" int rfx_hip_last_kernel_ms(rfx_ctx_t *c, float *ms) {
    RFX_REQUIRE(c && ms, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(c->profile && c->evk_valid, RFX_ESTATE, "no profiled kernel recorded (call rfx_hip_ctx_profile(ctx, 1) first)");
    hipEvent_t *p = c->evk[(c->evk_n - 1) & 7];
    RFX_HIP_CHECK(hipEventSynchronize(p[1]));
    RFX_HIP_CHECK(hipEventElapsedTime(ms, p[0], p[1]));
    return RFX_OK;
}

// ---- generator ----
__global__ __launch_bounds__(RFX_BLOCK) void k_gen_i64(i64 *out, i64 n, u64 seed, i64 row0, u64 mod) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK)
        out[i] = (i64)(rfx_splitmix_mix(seed + (u64)(row0 + i + 1) * 0x9E3779B97F4A7C15ULL) % mod);
}
__global__ __launch_bounds__(RFX_BLOCK) void k_gen_f64(double *out, i64 n, u64 seed, i64 row0) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK)
        out[i] = (double)(rfx_splitmix_mix(seed + (u64)(row0 + i + 1) * 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53;
}
extern "C" int rfx_hip_gen_i64(rfx_ctx_t *c, int64_t *d_out, int64_t n, uint64_t seed, int64_t row0, uint64_t modulus) {
    RFX_REQUIRE(c && (d_out || n == 0), RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(modulus > 0, RFX_EINVAL, "modulus must be > 0");
    if (n <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_gen_i64, dim3(rfx_grid(c)), dim3(RFX_BLOCK), 0, c->stream, (i64 *)d_out, (i64)n, (u64)seed, (i64)row0, (u64)modulus);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_gen_f64(rfx_ctx_t *c, double *d_out, int64_t n, uint64_t seed, int64_t row0) {
    RFX_REQUIRE(c && (d_out || n == 0), RFX_EINVAL, "NULL argument");
    if (n <= 0) return RFX_OK;
    hipLaunchKernelGGL(k_gen_f64, dim3(rfx_grid(c)), dim3(RFX_BLOCK), 0, c->stream, d_out, (i64)n, (u64)seed, (i64)row0);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- plan builder ----
static int plan_col(Plan *P, const void *p) {
    for (int i = 0; i < P->ncols; i++)
        if ((const void *)P->cols[i] == p) return i;
    if (P->ncols >= RFX_MAX_COLS) return -1;
    P->cols[P->ncols] = (const u64 *)p;
    return P->ncols++;
}

int rfx_plan_add_col(Plan *P, const void *col) { return plan_col(P, col); }

static inline u64 host_f64_bits(double d) { u64 b; memcpy(&b, &d, 8); return b; }

int rfx_plan_build(Plan *P, const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg,
                   const void *extra_col, int *extra_idx, i64 nrows, i64 row0) {
    memset(P, 0, sizeof(*P));
    RFX_REQUIRE(npred >= 0 && npred <= RFX_MAX_PREDS, RFX_ELIMIT, "too many predicates");
    RFX_REQUIRE(nagg >= 0 && nagg <= RFX_MAX_AGGS, RFX_ELIMIT, "too many aggregates");
    RFX_REQUIRE(logic == RFX_AND || logic == RFX_OR, RFX_EINVAL, "logic must be RFX_AND or RFX_OR");
    RFX_REQUIRE(nrows >= 0, RFX_EINVAL, "nrows < 0");
    RFX_REQUIRE(npred == 0 || preds, RFX_EINVAL, "preds is NULL");
    RFX_REQUIRE(nagg == 0 || aggs, RFX_EINVAL, "aggs is NULL");
    P->npred = npred;
    P->nagg = nagg;
    P->logic = logic;
    P->nrows = nrows;
    P->row0 = row0;
    if (extra_col) {
        int ci = plan_col(P, extra_col);
        RFX_REQUIRE(ci >= 0, RFX_ELIMIT, "too many distinct columns");
        if (extra_idx) *extra_idx = ci;
    }
    for (int i = 0; i < npred; i++) {
        const rfx_pred_t *p = &preds[i];
        PlanPred *q = &P->preds[i];
        RFX_REQUIRE(p->d_col != NULL || nrows == 0, RFX_EINVAL, "predicate column is NULL");
        RFX_REQUIRE(p->col_type == RFX_I64 || p->col_type == RFX_F64, RFX_EINVAL, "predicate column type must be i64 or f64");
        RFX_REQUIRE(p->rhs_type == RFX_I64 || p->rhs_type == RFX_F64, RFX_EINVAL, "predicate rhs type must be i64 or f64");
        RFX_REQUIRE(p->op >= RFX_EQ && p->op <= RFX_GE, RFX_EINVAL, "bad comparison operator");
        q->col = plan_col(P, p->d_col);
        RFX_REQUIRE(q->col >= 0, RFX_ELIMIT, "too many distinct columns");
        q->op = p->op;
        if (p->more & RFX_PRED_TREE) { // a leaf of a nested tree: depth (0..3) and the parentheses that close after it
            const int d = p->more & 15, k = (p->more >> 4) & 15;
            RFX_REQUIRE((p->more & ~(RFX_PRED_TREE | 255)) == 0 && d <= 3 && k <= d, RFX_EINVAL, "predicate tree: depth 0..3, close <= depth");
            RFX_REQUIRE(npred >= 3, RFX_EINVAL, "predicate tree: a tree deeper than two levels has at least three comparisons");
            q->more = 0;
            q->tree = RFX_PRED_TREE | d | (k << 4);
        } else {
            RFX_REQUIRE((p->more == 0 || p->more == 1) && !(p->more && i == npred - 1), RFX_EINVAL, "predicate `more` must be 0 / 1, and 0 on the last one");
            q->more = p->more;
            q->tree = 0;
        }
        q->dom_f64 = (p->col_type == RFX_F64 || p->rhs_type == RFX_F64);
        q->lhs_cvt = q->dom_f64 && p->col_type == RFX_I64;
        if (p->d_rhs_col) {
            q->rhs_col = plan_col(P, p->d_rhs_col);
            RFX_REQUIRE(q->rhs_col >= 0, RFX_ELIMIT, "too many distinct columns");
            q->rhs_cvt = q->dom_f64 && p->rhs_type == RFX_I64;
            q->rhs_bits = 0;
        } else {
            q->rhs_col = -1;
            q->rhs_cvt = 0;
            if (!q->dom_f64) q->rhs_bits = (u64)p->rhs_i;
            else if (p->rhs_type == RFX_F64) q->rhs_bits = host_f64_bits(p->rhs_f);
            else q->rhs_bits = (p->rhs_i == RFX_NULL_I64_D) ? RFX_NAN_BITS : host_f64_bits((double)p->rhs_i); // i64_to_f64, core/ops.h:250
        }
    }
    { // either every comparison is a tree leaf or none is; the parentheses must nest (a leaf never sits above the level left open before it)
        int ntree = 0, cur = 0;
        for (int i = 0; i < npred; i++) {
            const int t = P->preds[i].tree;
            if (!t) continue;
            ntree++;
            const int d = t & 15, k = (t >> 4) & 15;
            RFX_REQUIRE(d >= cur, RFX_EINVAL, "predicate tree: a comparison above the level its predecessor left open");
            cur = d - k;
        }
        RFX_REQUIRE(ntree == 0 || (ntree == npred && cur == 0), RFX_EINVAL, "predicate tree: every comparison must carry the tree form, and every parenthesis must close");
    }
    for (int i = 0; i < RFX_MAX_AGGS; i++) P->aggs[i].kind = -1;
    for (int i = 0; i < nagg; i++) {
        const rfx_agg_t *a = &aggs[i];
        PlanAgg *q = &P->aggs[i];
        RFX_REQUIRE(a->kind >= RFX_AGG_SUM && a->kind <= RFX_AGG_FIRST, RFX_EINVAL, "bad aggregate kind");
        q->kind = a->kind;
        q->skipnull = 0;
        if (a->kind == RFX_AGG_COUNT) {
            // (count expr) is not the reference's count of the expression's rows when grouped (it answers the number of
            // groups, a quirk of its lazy-argument collection): not reproduced, refused instead
            RFX_REQUIRE(a->xop == RFX_X_NONE && a->nxnodes == 0, RFX_EINVAL, "count of an expression is not supported");
            q->col = -1;
            q->f64 = 0;
        } else {
            if (a->nxnodes > 0) { // expression tree: nodes in evaluation order
                RFX_REQUIRE(a->nxnodes <= RFX_MAX_XNODES && a->xnodes != NULL, RFX_ELIMIT, "1..RFX_MAX_XNODES expression nodes");
                RFX_REQUIRE(a->kind != RFX_AGG_FIRST, RFX_EINVAL, "first of an expression is not supported");
                PlanExpr x;
                memset(&x, 0, sizeof(x));
                x.nops = a->nxnodes;
                for (int i = 0; i < a->nxnodes; i++) {
                    const rfx_xnode_t *src = &a->xnodes[i];
                    PlanXNode &n = x.ops[i];
                    RFX_REQUIRE(src->op >= RFX_X_ADD && src->op <= RFX_X_MOD, RFX_EINVAL, "bad expression operator");
                    n.op = src->op;
                    const rfx_xoperand_t *opnd[2] = {&src->l, &src->r};
                    int kind[2], idx[2], f64[2];
                    u64 atoms[2];
                    for (int j = 0; j < 2; j++) {
                        const rfx_xoperand_t *o = opnd[j];
                        kind[j] = o->kind;
                        idx[j] = 0;
                        atoms[j] = 0;
                        if (o->kind == RFX_XK_COL) {
                            RFX_REQUIRE(o->d_col != NULL && (o->type == RFX_I64 || o->type == RFX_F64), RFX_EINVAL, "expression column operand");
                            idx[j] = plan_col(P, o->d_col);
                            RFX_REQUIRE(idx[j] >= 0, RFX_ELIMIT, "too many distinct columns");
                            f64[j] = o->type == RFX_F64;
                        } else if (o->kind == RFX_XK_ATOM) {
                            RFX_REQUIRE(o->type == RFX_I64 || o->type == RFX_F64, RFX_EINVAL, "expression atom type");
                            f64[j] = o->type == RFX_F64;
                            atoms[j] = f64[j] ? host_f64_bits(o->f) : (u64)o->i;
                        } else {
                            RFX_REQUIRE(o->kind == RFX_XK_NODE && o->node >= 0 && o->node < i && o->node < RFX_MAX_XNODES - 1, RFX_EINVAL,
                                        "expression node operand must reference an earlier node");
                            idx[j] = (int)o->node;
                            f64[j] = x.ops[o->node].o_f64;
                        }
                    }
                    n.l_kind = kind[0]; n.r_kind = kind[1];
                    n.l_idx = idx[0]; n.r_idx = idx[1];
                    n.l_f64 = f64[0]; n.r_f64 = f64[1];
                    n.l_atom = atoms[0]; n.r_atom = atoms[1];
                    n.o_f64 = RFX_XOP_RESULT_F64(n.op, n.l_f64, n.r_f64);
                }
                x.out_f64 = x.ops[x.nops - 1].o_f64;
                int xi = 0;
                for (; xi < P->nx; xi++)
                    if (memcmp(&P->xs[xi], &x, sizeof(x)) == 0) break;
                if (xi == P->nx) {
                    RFX_REQUIRE(P->nx < RFX_MAX_EXPRS, RFX_ELIMIT, "too many distinct expressions");
                    P->xs[P->nx++] = x;
                }
                q->col = RFX_XCOL + xi;
                q->f64 = x.out_f64;
                q->skipnull = 1;
                continue;
            }
            RFX_REQUIRE(a->d_col != NULL || nrows == 0, RFX_EINVAL, "aggregate column is NULL");
            RFX_REQUIRE(a->col_type == RFX_I64 || a->col_type == RFX_F64, RFX_EINVAL, "aggregate column type must be i64 or f64");
            const int ci = plan_col(P, a->d_col);
            RFX_REQUIRE(ci >= 0, RFX_ELIMIT, "too many distinct columns");
            if (a->xop == RFX_X_NONE) {
                q->col = ci;
                q->f64 = (a->col_type == RFX_F64);
            } else {
                RFX_REQUIRE(a->xop >= RFX_X_ADD && a->xop <= RFX_X_MOD, RFX_EINVAL, "bad expression operator");
                RFX_REQUIRE(a->xrhs_type == RFX_I64 || a->xrhs_type == RFX_F64, RFX_EINVAL, "expression operand type must be i64 or f64");
                RFX_REQUIRE(a->kind != RFX_AGG_FIRST, RFX_EINVAL, "first of an expression is not supported");
                PlanExpr x;
                memset(&x, 0, sizeof(x));
                x.nops = 1;
                PlanXNode &n = x.ops[0];
                n.op = a->xop;
                int oc = -1;
                if (a->d_xrhs_col) {
                    oc = plan_col(P, a->d_xrhs_col);
                    RFX_REQUIRE(oc >= 0, RFX_ELIMIT, "too many distinct columns");
                }
                const u64 atom = (a->xrhs_type == RFX_F64) ? host_f64_bits(a->xrhs_f) : (u64)a->xrhs_i;
                const bool swap = (a->xflags & RFX_XF_SWAP) != 0;
                const int ck = RFX_XK_COL, ok = (oc >= 0) ? RFX_XK_COL : RFX_XK_ATOM;
                n.l_kind = swap ? ok : ck;
                n.r_kind = swap ? ck : ok;
                n.l_idx = swap ? oc : ci;
                n.r_idx = swap ? ci : oc;
                n.l_f64 = swap ? (a->xrhs_type == RFX_F64) : (a->col_type == RFX_F64);
                n.r_f64 = swap ? (a->col_type == RFX_F64) : (a->xrhs_type == RFX_F64);
                n.l_atom = (swap && oc < 0) ? atom : 0;
                n.r_atom = (!swap && oc < 0) ? atom : 0;
                n.o_f64 = RFX_XOP_RESULT_F64(a->xop, n.l_f64, n.r_f64);
                x.out_f64 = n.o_f64;
                int xi = 0;
                for (; xi < P->nx; xi++)
                    if (memcmp(&P->xs[xi], &x, sizeof(x)) == 0) break;
                if (xi == P->nx) {
                    RFX_REQUIRE(P->nx < RFX_MAX_EXPRS, RFX_ELIMIT, "too many distinct expressions");
                    P->xs[P->nx++] = x;
                }
                q->col = RFX_XCOL + xi;
                q->f64 = x.out_f64;
                q->skipnull = 1;
            }
        }
    }
    return RFX_OK;
}
