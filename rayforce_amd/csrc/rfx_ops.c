/*
 * rfx_ops.c -- the drop-in operator layer (host code stays in C, as in the reference): obj_p-shaped entry points that
 * plan a RayforceDB select / where / by query onto the flat HIP ABI (rfx_hip.h).
 *
 *   rfx_select walks the select dictionary the way ray_select does (core/query.c:243-654): `from:` is evaluated through
 *   the host's eval, `where:` is an expression LIST whose head is a function object (the parser already substituted the
 *   built-in for the symbol, core/parse.c:771-772), `by:` is a column symbol, every other key is an output mapping
 *   `(aggr col)`.  Supported shapes run as <= 4 kernel launches on HBM-resident columns; anything else goes back to
 *   the host's own ray_select (plugin mode) -- never to a CPU re-implementation of ours.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <string.h>
#include <time.h>
#include "rfx_abi.h"
#include "rfx_hip.h"
#include "rfx_exec.h"
#include "rfx_ops.h"

typedef rfx_obj_p obj_p;

/* rfx_host.c */
obj_p rfx_host_null(void);
obj_p rfx_host_b8(int8_t v);
obj_p rfx_host_err(const char *msg);
obj_p rfx_host_eval(obj_p o);

/* ------------------------------------------------------------------------------------------------ host binding */
static struct {
    int bound; /* 0 = not yet, 1 = reference host, 2 = standalone */
    obj_p (*vector)(int8_t, int64_t);
    obj_p (*table)(obj_p, obj_p);
    obj_p (*i64)(int64_t);
    obj_p (*f64)(double);
    void (*drop)(obj_p);
    obj_p (*clone)(obj_p);
    obj_p (*eval)(obj_p);
    obj_p (*err)(const char *);
    int64_t (*intern)(const char *, int64_t);
    const char *(*symname)(int64_t);
    obj_p null_obj;
    /* the host's own built-ins, for recognising function objects inside parsed expressions and for delegation */
    void *f[32];
} H;
enum { F_SUM, F_AVG, F_MIN, F_MAX, F_COUNT, F_FIRST, F_EQ, F_NE, F_LT, F_GT, F_LE, F_GE, F_AND, F_OR, F_SELECT, F_ADD, F_SUB, F_MUL, F_FDIV, F_DIV, F_MOD, F_XBAR, F_LJ, F_IJ, F_UPDATE, F_TAKE, F_IN, F_WITHIN, F_NOT, F_N };
static const char *HOST_FN[F_N] = {"ray_sum", "ray_avg", "ray_min", "ray_max", "ray_count", "ray_first", "ray_eq",  "ray_ne",  "ray_lt",  "ray_gt",
                                   "ray_le",  "ray_ge",  "ray_and", "ray_or",  "ray_select", "ray_add",  "ray_sub", "ray_mul", "ray_fdiv", "ray_div", "ray_mod", "ray_xbar",
                                   "ray_left_join", "ray_inner_join", "ray_update", "ray_take", "ray_in", "ray_within", "ray_not"}; /* (in / within / not: recognised inside where: only) */
/* xbar is recognised inside `by:` only; the standalone object model still needs a distinct function object for it:
 * this stub is never called by this library. */
static obj_p x_stub_xbar(obj_p a, obj_p b) { (void)a; (void)b; return NULL; }
static void *OUR_FN[F_N];
static void *g_host_where, *g_host_at, *g_host_group; /* the host's built-ins behind rfx_where / rfx_at / rfx_group (NULL without a host) */
static char g_err[640];
static int g_last_gpu = 0;

const char *rfx_ops_last_error(void) { return g_err; }
int rfx_last_select_on_gpu(void) { return g_last_gpu; }

int rfx_host_bind(void) {
    if (H.bound) return H.bound == 1;
    OUR_FN[F_SUM] = (void *)rfx_sum; OUR_FN[F_AVG] = (void *)rfx_avg; OUR_FN[F_MIN] = (void *)rfx_min; OUR_FN[F_MAX] = (void *)rfx_max;
    OUR_FN[F_COUNT] = (void *)rfx_count; OUR_FN[F_FIRST] = (void *)rfx_first; OUR_FN[F_EQ] = (void *)rfx_eq; OUR_FN[F_NE] = (void *)rfx_ne;
    OUR_FN[F_LT] = (void *)rfx_lt; OUR_FN[F_GT] = (void *)rfx_gt; OUR_FN[F_LE] = (void *)rfx_le; OUR_FN[F_GE] = (void *)rfx_ge;
    OUR_FN[F_AND] = (void *)rfx_and; OUR_FN[F_OR] = (void *)rfx_or; OUR_FN[F_SELECT] = (void *)rfx_select;
    OUR_FN[F_ADD] = (void *)rfx_add; OUR_FN[F_SUB] = (void *)rfx_sub; OUR_FN[F_MUL] = (void *)rfx_mul; OUR_FN[F_FDIV] = (void *)rfx_div; OUR_FN[F_DIV] = (void *)rfx_floordiv; OUR_FN[F_MOD] = (void *)rfx_mod;
    OUR_FN[F_XBAR] = (void *)x_stub_xbar;
    OUR_FN[F_LJ] = (void *)rfx_left_join; OUR_FN[F_IJ] = (void *)rfx_inner_join; OUR_FN[F_UPDATE] = (void *)rfx_update;
    void *v = dlsym(RTLD_DEFAULT, "vector"), *t = dlsym(RTLD_DEFAULT, "table"), *e = dlsym(RTLD_DEFAULT, "eval");
    void *rs = dlsym(RTLD_DEFAULT, "ray_select"), *nu = dlsym(RTLD_DEFAULT, "__NULL_OBJ");
    if (v && t && e && rs && nu && !getenv("RFX_FORCE_STANDALONE")) {
        H.vector = (obj_p(*)(int8_t, int64_t))v;
        H.table = (obj_p(*)(obj_p, obj_p))t;
        H.eval = (obj_p(*)(obj_p))e;
        H.i64 = (obj_p(*)(int64_t))dlsym(RTLD_DEFAULT, "i64");
        H.f64 = (obj_p(*)(double))dlsym(RTLD_DEFAULT, "f64");
        H.drop = (void (*)(obj_p))dlsym(RTLD_DEFAULT, "drop_obj");
        H.clone = (obj_p(*)(obj_p))dlsym(RTLD_DEFAULT, "clone_obj");
        H.err = (obj_p(*)(const char *))dlsym(RTLD_DEFAULT, "ray_err");
        H.intern = (int64_t(*)(const char *, int64_t))dlsym(RTLD_DEFAULT, "symbols_intern");
        H.symname = (const char *(*)(int64_t))dlsym(RTLD_DEFAULT, "str_from_symbol");
        H.null_obj = (obj_p)nu;
        for (int i = 0; i < F_N; i++) H.f[i] = dlsym(RTLD_DEFAULT, HOST_FN[i]);
        g_host_where = dlsym(RTLD_DEFAULT, "ray_where"); /* (not in H.f: never recognised inside a query, only handed back to -- refused1x) */
        g_host_at = dlsym(RTLD_DEFAULT, "ray_at");
        g_host_group = dlsym(RTLD_DEFAULT, "ray_group");
        if (H.i64 && H.f64 && H.drop && H.clone && H.err && H.intern && H.symname) {
            H.bound = 1;
            return 1;
        }
    }
    H.vector = rfx_host_vector;
    H.table = rfx_host_table;
    H.i64 = rfx_host_i64;
    H.f64 = rfx_host_f64;
    H.drop = rfx_host_drop;
    H.clone = rfx_host_clone;
    H.eval = rfx_host_eval;
    H.err = rfx_host_err;
    H.intern = rfx_host_intern;
    H.symname = rfx_host_symbol_name;
    H.null_obj = rfx_host_null();
    memset(H.f, 0, sizeof(H.f));
    H.bound = 2;
    return 0;
}

obj_p rfx_host_fn(const char *name) {
    static const struct { const char *n; int f; int type; int attrs; } T[] = {
        {"sum", F_SUM, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"avg", F_AVG, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"min", F_MIN, RFX_TYPE_UNARY, RFX_FN_AGGR},
        {"max", F_MAX, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"count", F_COUNT, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"first", F_FIRST, RFX_TYPE_UNARY, RFX_FN_AGGR},
        {"==", F_EQ, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"!=", F_NE, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"<", F_LT, RFX_TYPE_BINARY, RFX_FN_ATOMIC},
        {">", F_GT, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"<=", F_LE, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {">=", F_GE, RFX_TYPE_BINARY, RFX_FN_ATOMIC},
        {"and", F_AND, RFX_TYPE_VARY, RFX_FN_SPECIAL_FORM}, {"or", F_OR, RFX_TYPE_VARY, RFX_FN_SPECIAL_FORM}, {"select", F_SELECT, RFX_TYPE_UNARY, 0},
        {"+", F_ADD, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"-", F_SUB, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"*", F_MUL, RFX_TYPE_BINARY, RFX_FN_ATOMIC},
        {"div", F_FDIV, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"/", F_DIV, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"%", F_MOD, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"xbar", F_XBAR, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"update", F_UPDATE, RFX_TYPE_UNARY, 0}};
    rfx_host_bind();
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); i++)
        if (strcmp(T[i].n, name) == 0) {
            obj_p o = rfx_host_i64((int64_t)(intptr_t)OUR_FN[T[i].f]);
            o->type = (int8_t)T[i].type; /* function objects carry the POSITIVE type code (core/env.c:66-74) */
            o->attrs = (uint8_t)T[i].attrs;
            return o;
        }
    return NULL;
}

static obj_p fail(const char *msg) {
    rfx_host_bind();
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return H.err(msg);
}
static obj_p fail_hip(const char *what) {
    char b[600];
    snprintf(b, sizeof(b), "%s: %s", what, rfx_hip_last_error());
    return fail(b);
}
static int g_refused_sharded;
static obj_p fail_ctx(void) {
    if (g_refused_sharded) return fail("this operator needs its columns whole on one device: with RFX_SHARDS / RFX_DEVICES the operator layer answers rfx_select (and pin / unpin / invalidate / stats) only");
    return fail_hip("no usable MI355X");
}
/* ... unless there is a host beside us: then the operator is simply the host's own again (the shards hold row ranges; RFX_SHARDS /
 * RFX_DEVICES is about rfx_select).  Defined below, once HOST_CALL is. */
static obj_p refused1(int f, obj_p x);
static obj_p refused2(int f, obj_p x, obj_p y);
static obj_p refusedn(int f, obj_p *x, int64_t n);

/* which built-in does this function object denote? -1 if none */
static int fn_id(obj_p o) {
    if (!o || (o->type != RFX_TYPE_UNARY && o->type != RFX_TYPE_BINARY && o->type != RFX_TYPE_VARY)) return -1;
    void *p = (void *)(intptr_t)o->i64;
    for (int i = 0; i < F_N; i++)
        if ((OUR_FN[i] && p == OUR_FN[i]) || (H.f[i] && p == H.f[i])) return i;
    return -1;
}

/* ------------------------------------------------------------------------------------------------ device + residency */
/* The operator layer owns ONE planner (rfx_exec.h) over one context per SHARD.  Default: one shard on device $RFX_DEVICE (0).
 * RFX_DEVICES="0,1,2,3" | "all": one shard per listed device -- the evaluator process that owns the node's GPUs: every column is split
 * row-range over them at upload (rfx_pin), rfx_select runs every shard's pass on its own host thread and merges the partial tables with
 * ONE fused RCCL exchange (core/query.c:607-654 -> aggr_map core/aggr.c:375 / AGGR_COLLECT :163-181, one level up).
 * RFX_SHARDS=k: k shards over the listed devices round robin (k > devices: several per device, merged by a kernel -- how the sharded
 * door is tested on a one-GPU box).  g_ctx is shard 0's context: the operators that are not sharded run there. */
static rfx_ctx_t *g_ctx;
static rfx_ctx_t *g_ctxs[RFX_MAX_SHARDS];
static rfx_exec_t *g_x;
static int g_nshards = 1;
static int g_device = -1;
static int g_cfg_devices[RFX_MAX_SHARDS], g_cfg_ndev, g_cfg_shards;
int rfx_ops_set_device(int device) {
    if (g_ctx) return RFX_ESTATE;
    g_device = device;
    return RFX_OK;
}
int rfx_ops_set_shards(const int *devices, int ndevices, int nshards) {
    if (g_ctx) return RFX_ESTATE;
    if (ndevices < 0 || ndevices > RFX_MAX_SHARDS || nshards < 0 || nshards > RFX_MAX_SHARDS || (ndevices && !devices)) return RFX_EINVAL;
    for (int i = 0; i < ndevices; i++) g_cfg_devices[i] = devices[i];
    g_cfg_ndev = ndevices;
    g_cfg_shards = nshards;
    return RFX_OK;
}
int rfx_ops_shards(void) { return g_nshards; }
rfx_exec_t *rfx_ops_exec(void) { return g_x; }
static int ensure_ctx(void) {
    if (g_ctx) return RFX_OK;
    int devs[RFX_MAX_SHARDS], ndev = g_cfg_ndev, nsh = g_cfg_shards;
    for (int i = 0; i < ndev; i++) devs[i] = g_cfg_devices[i];
    if (ndev == 0) {
        const char *e = getenv("RFX_DEVICES");
        if (e && strcmp(e, "all") == 0) {
            const int n = rfx_hip_device_count();
            for (int i = 0; i < n && i < RFX_MAX_SHARDS; i++) devs[ndev++] = i;
        } else if (e && *e) {
            for (const char *q = e; *q && ndev < RFX_MAX_SHARDS;) {
                devs[ndev++] = atoi(q);
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
        }
    }
    if (ndev == 0) {
        if (g_device < 0) {
            const char *e = getenv("RFX_DEVICE");
            g_device = e ? atoi(e) : 0;
        }
        devs[ndev++] = g_device;
    }
    if (nsh == 0) {
        const char *e = getenv("RFX_SHARDS");
        nsh = e ? atoi(e) : 0;
    }
    if (nsh < ndev) nsh = ndev;
    if (nsh > RFX_MAX_SHARDS) nsh = RFX_MAX_SHARDS;
    int rc = RFX_OK, made = 0;
    for (int s = 0; s < nsh && rc == RFX_OK; s++) {
        rc = rfx_hip_ctx_create(devs[s % ndev], NULL, &g_ctxs[s]);
        if (rc == RFX_OK) made++;
    }
    if (rc == RFX_OK) rc = rfx_exec_create(g_ctxs, nsh, &g_x);
    if (rc == RFX_OK) rc = rfx_exec_comm_init_all(g_x); /* (communicators among the devices when there are several) */
    if (rc != RFX_OK) {
        if (g_x) rfx_exec_destroy(g_x);
        g_x = NULL;
        for (int s = 0; s < made; s++) { rfx_hip_ctx_destroy(g_ctxs[s]); g_ctxs[s] = NULL; }
        return rc;
    }
    g_device = devs[0];
    g_nshards = nsh;
    g_ctx = g_ctxs[0];
    rfx_hip_ctx_bind_thread(g_ctx);
    if (getenv("RFX_TRACE")) fprintf(stderr, "[rfx] operator layer: %d shard(s) over %d device(s)\n", nsh, ndev);
    return RFX_OK;
}
static void op_begin(void);
static void op_end(void);
/* ONE PROCESS PER DEVICE (torch.distributed launches, bench.py --gpus N under a launcher): every process owns the rows [row0, row0 + n) of
 * every table; rank 0 draws the 128-byte id (rfx_dist_unique_id), the host ships it, every process calls rfx_ops_dist_init -- after that
 * rfx_select's planner exchanges scopes / partials / group tables with the other processes through the context's RCCL communicator and every
 * process returns the WHOLE answer.  (Projections stay local: a process returns its own rows.) */
int rfx_ops_dist_init(int world, int rank, const void *id128) {
    rfx_host_bind();
    op_begin();
    int rc = ensure_ctx();
    if (rc == RFX_OK && g_nshards > 1) rc = RFX_ESTATE; /* (shards inside a process and processes: one or the other) */
    if (rc == RFX_OK) rc = rfx_dist_init(g_ctx, world, rank, id128);
    op_end();
    return rc;
}
int rfx_ops_dist_finalize(void) {
    if (!g_ctx) return RFX_OK;
    op_begin();
    const int rc = rfx_dist_finalize(g_ctx);
    op_end();
    return rc;
}
/* for the operators that need a column WHOLE on one device (everything but rfx_select / rfx_pin / rfx_unpin / rfx_invalidate / rfx_stats) */
static int ensure_ctx1(void) {
    const int rc = ensure_ctx();
    g_refused_sharded = rc == RFX_OK && g_nshards > 1;
    return g_refused_sharded ? RFX_ELIMIT : rc;
}

/* Residency cache: host vector payload -> device copy, keyed by (payload address, length, type).
 *
 * A cached copy is used again only if it is PROVEN current:
 *   - default: the FULL payload is checksummed on every use (threaded multiply-xor over every 8-byte word, position dependent) and
 *     compared with the checksum taken at upload -- an in-place write of any single cell, a copy-on-write successor that the
 *     allocator put at the same address, a freed temporary whose address was recycled: all change the checksum and cost one
 *     re-upload, never a stale answer.  (Round 1 sampled 64 cells: one changed cell could escape; B8 masks collided almost
 *     always.)  Reading the host payload costs ~10 ms per GB on the box's cores -- still 15x cheaper than the PCIe upload it saves;
 *     where the kernel tracks soft-dirty pages (round 3, sd_* below) the checksum is taken once and later uses look at the
 *     payload's page-table bits instead: O(pages), 16 MB of pagemap per 8 GB column;
 *   - rfx_pin: the host promises to call rfx_invalidate / rfx_unpin before it writes into the vector (INTEGRATION.md shows the
 *     two places in the reference: `set` of a column and the rc == 1 in-place arithmetic, core/math.c:2248); a pinned entry is
 *     trusted without the checksum, which is what makes repeated queries over 8 GB columns free of host work.
 * Entries touched by the operator call in flight are never evicted (its descriptors hold their device pointers); temporaries
 * that only live for one call (masks handed to `where`, id vectors of `at` / MAPFILTER) are uploaded into per-call scratch and
 * not cached at all. */
typedef struct {
    const void *host;
    int64_t len;
    int type;
    uint64_t sum;
    void *dev;
    size_t bytes;
    int pinned;
    uint64_t tick, epoch;
    int tracked;       /* soft-dirty tracking: the payload's whole pages were clean-marked BEFORE `sum` was taken (see sd_*) */
    int stable;        /* ... uses in a row at which the checksum found the payload unchanged (tracking starts at SD_STABLE_USES) */
    int sd_never;      /* ... cannot be tracked (file-backed / shared pages): the checksum every time */
    uint64_t edge_sum; /* ... checksum of the payload bytes in its first and last, partial pages (they hold other objects too) */
    size_t dbytes;     /* bytes of the device copy (`bytes` are the host payload's: a 4-byte column is widened on the device) */
    int scope_ok;      /* [smin, smax] = index_scope_i64 of the WHOLE column (no filter), taken from this very copy: valid as long as the copy is */
    int64_t smin, smax;
    void *devs[RFX_MAX_SHARDS]; /* the copy, shard by shard (devs[0] == dev): rows rfx_exec_split(len, shards, s) of the column */
} resident_t;
static resident_t *g_res;
static int g_nres, g_capres;
static uint64_t g_tick, g_epoch = 1;
static size_t g_res_bytes;
static int64_t g_stat[10]; /* see rfx_stats */
static int64_t g_sd_hits;  /* uses of an unpinned cached column proven current by its pages' soft-dirty bits instead of the checksum */
enum { ST_SELECT_GPU, ST_SELECT_DELEGATED, ST_JOIN_GPU, ST_JOIN_DELEGATED, ST_UPLOADS, ST_CACHE_HITS, ST_CACHE_STALE, ST_OPS, ST_SCOPE_SAMPLED, ST_SCOPE_RETRIED };

typedef struct {
    const unsigned char *p;
    size_t bytes;
    uint64_t h;
} sum_job_t;
static uint64_t sum_range(const unsigned char *p, size_t bytes) {
    /* four independent multiply-xor lanes (the multiply's latency is the limit of a single chain), folded in a fixed order */
    const uint64_t K = 0x9E3779B97F4A7C15ULL;
    uint64_t h0 = 0x243F6A8885A308D3ULL, h1 = 0x13198A2E03707344ULL, h2 = 0xA4093822299F31D0ULL, h3 = 0x082EFA98EC4E6C89ULL;
    size_t nw = bytes / 8, i = 0;
    const uint64_t *w = (const uint64_t *)p; /* payloads are 8-byte aligned (obj + 16, 32-byte aligned blocks) */
    if (((uintptr_t)p & 7) == 0) {
        for (; i + 4 <= nw; i += 4) {
            h0 = (h0 ^ w[i]) * K;
            h1 = (h1 ^ w[i + 1]) * K;
            h2 = (h2 ^ w[i + 2]) * K;
            h3 = (h3 ^ w[i + 3]) * K;
        }
        for (; i < nw; i++) h0 = (h0 ^ w[i]) * K;
    } else i = 0, nw = 0;
    uint64_t h = ((h0 ^ (h1 >> 29)) * K) ^ ((h2 ^ (h3 >> 31)) * K);
    for (size_t b = nw * 8; b < bytes; b++) h = (h ^ p[b]) * K;
    return h ^ (h >> 32);
}
static void *sum_worker(void *arg) {
    sum_job_t *j = (sum_job_t *)arg;
    j->h = sum_range(j->p, j->bytes);
    return NULL;
}
static uint64_t payload_sum(const void *p, size_t bytes) {
    enum { MAXT = 32 };
    int nt = 1;
    if (bytes >= ((size_t)8 << 20)) {
        long cores = sysconf(_SC_NPROCESSORS_ONLN);
        nt = cores > MAXT ? MAXT : (cores < 1 ? 1 : (int)cores);
        if ((size_t)nt > bytes >> 22) nt = (int)(bytes >> 22); /* >= 4 MB per thread */
    }
    if (nt <= 1) return sum_range((const unsigned char *)p, bytes) ^ (uint64_t)bytes;
    sum_job_t job[MAXT];
    pthread_t th[MAXT];
    size_t per = ((bytes / (size_t)nt) + 63) & ~(size_t)63, off = 0;
    int started = 0;
    for (int i = 0; i < nt; i++) {
        job[i].p = (const unsigned char *)p + off;
        job[i].bytes = (i == nt - 1 || off + per > bytes) ? bytes - off : per;
        off += job[i].bytes;
        if (i < nt - 1 && pthread_create(&th[i], NULL, sum_worker, &job[i]) == 0) started |= 1 << i;
        else sum_worker(&job[i]);
    }
    uint64_t h = (uint64_t)bytes;
    for (int i = 0; i < nt; i++) {
        if (started & (1 << i)) pthread_join(th[i], NULL);
        h = (h ^ job[i].h) * 0x9E3779B97F4A7C15ULL; /* chunk order matters: a value moved between chunks changes the sum */
    }
    return h;
}

/* ---- O(pages) validation of unpinned columns: soft-dirty page tracking ----
 * The full-payload checksum costs 0.3 s per 8 GB column and query.  Where the kernel tracks soft-dirty pages (CONFIG_MEM_SOFT_DIRTY:
 * writing "4" to /proc/self/clear_refs write-protects every page of the process and clears bit 55 of its pagemap entry; the first
 * write to a page afterwards sets it again) a cached payload is proven current by reading 8 bytes of pagemap per 4 KB page of it --
 * 16 MB for an 8 GB column -- provided its pages were cleared BEFORE the checksum that vouches for the device copy was taken:
 *   use of a tracked entry:   no soft-dirty page among the payload's WHOLE pages and the checksum of its first / last partial page
 *                             (shared with other objects -- the vector's own header with its reference count sits there) unchanged
 *                             -> current.  Anything else -> the entry is no longer tracked, and is treated like a new one:
 *   (re)validation / upload:  clear_refs FIRST (once per operator call; every other tracked entry is scanned just before, because
 *                             the clear wipes their evidence too: a dirty one loses its tracking and meets its checksum at its next
 *                             use), THEN the checksum, THEN the compare / upload.  A host write that races with the call lands after
 *                             the clear and is seen at the next use.
 * File-backed and shared pages (pagemap bit 61: an mmapped column file other processes may write) are never tracked.  The kernel
 * is PROBED once (map two pages, clear, write one, look); without the feature -- the build container's kernel has none, the MI355X
 * boxes' has -- or with RFX_SOFT_DIRTY=0 nothing changes: the checksum on every use.  Cost to the HOST: after a clear the first
 * write to each of its pages takes a minor fault; clears happen only in calls that upload or re-validate a column, never in the steady
 * state of repeated queries over unchanged columns.
 * Measured (MI355X box, tools/unpinned.py: the c3w query over three unpinned 8 GB columns): 156.6 ms per query by checksums (52 ms a
 * column on 32 threads), 66.7 ms by page bits read on one thread; the clear itself 1.4 s once. */
#define SD_MIN_BYTES ((size_t)1 << 20)
#define SD_STABLE_USES 2 /* a column is tracked once this many uses in a row found it unchanged: clear_refs walks EVERY page of the process
                          * (measured: 1.4 s with 24 GB resident), which only pays for columns that are read far more often than written */
static int g_sd_state = -1; /* -1 not probed, 0 unavailable / off, 1 works */
static int g_sd_pagemap = -1;
static uintptr_t g_sd_page = 4096;
static uint64_t g_sd_clear_epoch;
static int sd_clear(void) {
    int fd = open("/proc/self/clear_refs", O_WRONLY);
    if (fd < 0) return -1;
    const ssize_t w = write(fd, "4", 1);
    close(fd);
    return w == 1 ? 0 : -1;
}
/* any soft-dirty page in [lo, hi) (page-aligned)?  1 yes, 0 none, -1 cannot tell (read failed / file-backed or shared pages) */
static int sd_scan_range(uintptr_t lo, uintptr_t hi) {
    static __thread uint64_t buf[4096];
    for (uintptr_t a = lo; a < hi;) {
        size_t n = (hi - a) / g_sd_page;
        if (n > 4096) n = 4096;
        const ssize_t got = pread(g_sd_pagemap, buf, n * 8, (off_t)((a / g_sd_page) * 8));
        if (got != (ssize_t)(n * 8)) return -1;
        for (size_t i = 0; i < n; i++) {
            if (buf[i] & (1ULL << 61)) return -1;
            if (buf[i] & (1ULL << 55)) return 1;
        }
        a += n * g_sd_page;
    }
    return 0;
}
typedef struct {
    uintptr_t lo, hi;
    int r;
} sd_job_t;
static void *sd_worker(void *arg) {
    sd_job_t *j = (sd_job_t *)arg;
    j->r = sd_scan_range(j->lo, j->hi);
    return NULL;
}
/* the same over a large range: the kernel walks the page tables for every entry read (~10 ns a page: 20 ms per 8 GB), so the range is
 * split over up to 16 readers */
static int sd_scan(uintptr_t lo, uintptr_t hi) {
    enum { MAXT = 16 };
    const size_t pages = (hi - lo) / g_sd_page;
    int nt = (int)(pages >> 16); /* >= 65 536 pages (256 MB) per reader */
    if (nt > MAXT) nt = MAXT;
    if (nt <= 1) return sd_scan_range(lo, hi);
    sd_job_t job[MAXT];
    pthread_t th[MAXT];
    const size_t per = (pages + (size_t)nt - 1) / (size_t)nt;
    int started = 0, r = 0;
    for (int i = 0; i < nt; i++) {
        job[i].lo = lo + (uintptr_t)i * per * g_sd_page;
        job[i].hi = (i == nt - 1 || job[i].lo + per * g_sd_page > hi) ? hi : job[i].lo + per * g_sd_page;
        if (job[i].lo > hi) job[i].lo = hi;
        if (i < nt - 1 && pthread_create(&th[i], NULL, sd_worker, &job[i]) == 0) started |= 1 << i;
        else sd_worker(&job[i]);
    }
    for (int i = 0; i < nt; i++) {
        if (started & (1 << i)) pthread_join(th[i], NULL);
        if (job[i].r < 0) r = -1;
        else if (job[i].r > 0 && r == 0) r = 1;
    }
    return r;
}
static void sd_probe(void) {
    g_sd_state = 0;
    const char *e = getenv("RFX_SOFT_DIRTY"); /* OPT-IN (RFX_SOFT_DIRTY=1): clear_refs write-protects every page of the HOST process -- 0.7-1.4 s with
                                               * 24 GB resident, then a minor fault on the host's next write to each page, and other users of soft-dirty
                                               * bits in the same process (CRIU-style checkpointing) lose theirs.  tools/unpinned.py measures both sides. */
    if (!e || atoi(e) == 0) return;
    const long pg = sysconf(_SC_PAGESIZE);
    if (pg < 4096) return;
    g_sd_page = (uintptr_t)pg;
    g_sd_pagemap = open("/proc/self/pagemap", O_RDONLY);
    if (g_sd_pagemap < 0) return;
    unsigned char *m = (unsigned char *)mmap(NULL, 2 * g_sd_page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return;
    m[0] = 1;
    m[g_sd_page] = 1;
    if (sd_clear() == 0 && sd_scan((uintptr_t)m, (uintptr_t)m + 2 * g_sd_page) == 0) {
        *(volatile unsigned char *)(m + g_sd_page) = 2;
        if (sd_scan((uintptr_t)m, (uintptr_t)m + g_sd_page) == 0 && sd_scan((uintptr_t)m + g_sd_page, (uintptr_t)m + 2 * g_sd_page) == 1) g_sd_state = 1;
    }
    munmap(m, 2 * g_sd_page);
    if (getenv("RFX_TRACE")) fprintf(stderr, "[rfx] soft-dirty page tracking: %s\n", g_sd_state ? "available (unpinned columns are validated by their pages)" : "not available (full checksum per use)");
}
/* the payload's whole pages */
static int sd_interior(const void *host, size_t bytes, uintptr_t *lo, uintptr_t *hi) {
    const uintptr_t a = (uintptr_t)host, b = a + bytes;
    *lo = (a + g_sd_page - 1) & ~(g_sd_page - 1);
    *hi = b & ~(g_sd_page - 1);
    return *hi > *lo;
}
static uint64_t sd_edge_sum(const void *host, size_t bytes) {
    uintptr_t lo, hi;
    if (!sd_interior(host, bytes, &lo, &hi)) return 0;
    const uintptr_t a = (uintptr_t)host, b = a + bytes;
    const uint64_t h = sum_range((const unsigned char *)a, lo - a), t = sum_range((const unsigned char *)hi, b - hi);
    return h ^ ((t << 21) | (t >> 43));
}
static int sd_usable(const void *host, size_t bytes) {
    if (g_sd_state < 0) sd_probe();
    uintptr_t lo, hi;
    return g_sd_state == 1 && bytes >= SD_MIN_BYTES && sd_interior(host, bytes, &lo, &hi);
}
/* clean-mark the process' pages, once per operator call.  The clear wipes every OTHER tracked entry's evidence too, and a scan taken before
 * it cannot vouch for them: a host thread writing between that scan and the clear would leave a stale copy that looks clean for ever.  So
 * the order is clear FIRST, then every tracked entry meets its CHECKSUM again (52 ms per 8 GB; clears are rare -- only calls that start
 * tracking a column make one): a write before the clear changes the checksum (the entry loses its tracking and is refreshed at its next
 * use), a write after it sets the page's bit again.  Soft-dirty validation still assumes what the checksum assumes -- host writes go
 * through the CPU's page tables (device DMA into registered host memory marks nothing) -- which is why it is OPT-IN. */
static int sd_call_clear(void) {
    if (g_sd_clear_epoch == g_epoch) return 0;
    if (sd_clear() != 0) { /* the kernel took the feature away (permissions?): back to checksums for good */
        g_sd_state = 0;
        for (int i = 0; i < g_nres; i++) g_res[i].tracked = 0;
        return -1;
    }
    g_sd_clear_epoch = g_epoch;
    for (int i = 0; i < g_nres; i++) {
        if (!g_res[i].tracked) continue;
        if (payload_sum(g_res[i].host, g_res[i].bytes) != g_res[i].sum) g_res[i].tracked = 0, g_res[i].stable = 0;
        else g_res[i].edge_sum = sd_edge_sum(g_res[i].host, g_res[i].bytes);
    }
    return 0;
}
static int sd_entry_clean(const resident_t *r) {
    uintptr_t lo, hi;
    if (g_sd_state != 1 || !sd_interior(r->host, r->bytes, &lo, &hi)) return 0;
    return sd_scan(lo, hi) == 0 && sd_edge_sum(r->host, r->bytes) == r->edge_sum;
}

static void res_free(int i) {
    for (int s = 0; s < g_nshards; s++)
        if (g_res[i].devs[s]) {
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
            rfx_hip_free(g_ctxs[s], g_res[i].devs[s]);
        }
    if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    g_res_bytes -= g_res[i].dbytes;
    g_res[i] = g_res[--g_nres];
}
/* the columns the operator call in flight has named, shard by shard: what the planner translates shard 0's addresses with */
static rfx_qcol_t g_qcols[64];
static int g_nqcols;
static int qcol_add(void *const *devs) {
    if (g_nshards == 1) return RFX_OK;
    for (int i = 0; i < g_nqcols; i++)
        if (g_qcols[i].d[0] == devs[0]) return RFX_OK;
    if (g_nqcols >= (int)(sizeof(g_qcols) / sizeof(g_qcols[0]))) return RFX_ELIMIT;
    for (int s = 0; s < RFX_MAX_SHARDS; s++) g_qcols[g_nqcols].d[s] = s < g_nshards ? devs[s] : NULL;
    g_nqcols++;
    return RFX_OK;
}
static void op_begin(void);
static void op_end(void);
void rfx_cache_clear(void) {
    op_begin();
    while (g_nres) res_free(g_nres - 1);
    for (int sh = 0; sh < g_nshards && g_ctx; sh++) { /* ... and the blocks the contexts keep for reuse go back to the device */
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[sh]);
        rfx_hip_ctx_trim(g_ctxs[sh]);
    }
    if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    op_end();
}
int64_t rfx_cache_bytes(void) { return (int64_t)g_res_bytes; }

static size_t cache_budget(void) {
    const char *e = getenv("RFX_CACHE_BYTES");
    return e ? (size_t)strtoull(e, NULL, 10) : (size_t)200 << 30; /* of the 288 GB of HBM3E */
}

/* per-call device scratch (temporaries of the operator call in flight): released by op_end() */
static void *g_optmp[64];
static int g_noptmp;
/* ONE lock around everything this file keeps between calls (residency cache, per-operator scratch lists, the device context): the reference
 * calls built-ins from its pool workers, each with its own VM (core/pool.c:168-219), so two rfx_* calls may arrive at once.  An operator
 * holds the lock from op_begin to op_end; a thread that re-enters (the host evaluating `from:` calls an rfx_* built-in) counts depth
 * instead of locking twice.  Calls BACK into the host that may fan out to its pool (a delegated select / update / join / fold) run with
 * the lock released -- after the operator has let go of its device scratch -- so that workers calling rfx_* are not shut out. */
static pthread_mutex_t g_op_lock = PTHREAD_MUTEX_INITIALIZER;
static __thread int t_op_depth;
static void op_begin(void) {
    if (t_op_depth++ == 0) {
        pthread_mutex_lock(&g_op_lock);
        g_epoch++; /* (a nested operator keeps the outer one's epoch: the outer call's columns stay protected from eviction) */
        g_nqcols = 0;
        if (g_ctx) rfx_hip_ctx_bind_thread(g_ctx); /* (the host calls built-ins from any of its threads) */
    }
    g_stat[ST_OPS]++;
}
static void op_scratch_release(void) {
    for (int i = 0; i < g_noptmp; i++) rfx_hip_free(g_ctx, g_optmp[i]);
    g_noptmp = 0;
}
static void op_end(void) {
    if (t_op_depth == 1) op_scratch_release(); /* (a nested operator leaves the outer one's scratch alone) */
    if (--t_op_depth == 0) pthread_mutex_unlock(&g_op_lock);
}
/* around a call into the host that may run for long / on other threads: returns the depth to hand back to host_call_end */
static int host_call_begin(void) {
    const int d = t_op_depth;
    if (d > 0) {
        if (d == 1) op_scratch_release();
        t_op_depth = 0;
        pthread_mutex_unlock(&g_op_lock);
    }
    return d;
}
static void host_call_end(int d) {
    if (d > 0) {
        pthread_mutex_lock(&g_op_lock);
        t_op_depth = d;
        if (d == 1) { /* other threads' operators ran meanwhile: the per-call column table may hold THEIR (freed) entries -- start over as op_begin does */
            g_epoch++;
            g_nqcols = 0;
            if (g_ctx) rfx_hip_ctx_bind_thread(g_ctx);
        }
    }
}
#define HOST_CALL(call) ({ const int _hd = host_call_begin(); obj_p _hr = (call); host_call_end(_hd); _hr; })
/* device copy of a vector that lives for this call only (never cached) */
static int transient(obj_p v, const void **dev) {
    const int esz = (v->type == RFX_TYPE_B8) ? 1 : 8;
    const size_t bytes = (size_t)v->len * esz;
    if (g_noptmp >= (int)(sizeof(g_optmp) / sizeof(g_optmp[0]))) return RFX_ELIMIT;
    void *d = NULL;
    int rc = rfx_hip_malloc(g_ctx, &d, bytes ? bytes : 8);
    if (rc != RFX_OK) return rc;
    g_optmp[g_noptmp++] = d;
    if (bytes) rc = rfx_hip_h2d_pipelined(g_ctx, d, RFX_AS_RAW(v), bytes);
    g_stat[ST_UPLOADS]++;
    *dev = d;
    return rc;
}

#define RFX_MAX_PROXY 128
/* the virtual column, or TYPE_PARTEDLIST + an element type (B8 .. ENUM), core/rayforce.h:67-82 */
#define IS_PARTED_TYPE(t) ((t) == RFX_TYPE_MAPCOMMON || ((t) >= RFX_TYPE_PARTEDLIST && (t) <= RFX_TYPE_PARTEDLIST + RFX_TYPE_ENUM))
/* ---- parted tables (get-parted, core/vary.c:185-392) ----
 * A parted table's columns are LISTs of one mmapped vector per partition (TYPE_PARTEDLIST + element type) plus ONE virtual
 * column (TYPE_MAPCOMMON: a value per partition and the partition's row count; `Date`).  On the device a parted column is what
 * the reference's PARTED_MAP loops over (core/aggr.c:183-260) laid end to end: one contiguous column, partition after partition.
 * For the duration of one operator call such a table is seen through a VIEW: a table-shaped object of ours whose columns are
 * proxy headers {element type, total rows} that only resident() knows how to upload (partition by partition into its slice; the
 * virtual column is expanded on the device).  Proxies never reach the host: results are built from device data. */
typedef struct {
    rfx_obj_t hdr; /* type = element type (I64 for the virtual column), len = total rows */
    obj_p src;     /* the parted LIST / the MAPCOMMON pair */
    int kind;      /* 1: parted data column, 2: virtual (MAPCOMMON) column */
    int8_t vtype;  /* kind 2: type of the per-partition values (DATE / I64) */
} proxy_t;
static proxy_t *g_px[RFX_MAX_PROXY];
static int g_npx;
static void *g_pxmem[3];
static proxy_t *proxy_of(obj_p o) {
    for (int i = 0; i < g_npx; i++)
        if ((obj_p)g_px[i] == o) return g_px[i];
    return NULL;
}
static int is_parted_table(obj_p tab) {
    obj_p cols = RFX_AS_LIST(tab)[1];
    for (int64_t i = 0; i < cols->len; i++) {
        const int t = RFX_AS_LIST(cols)[i]->type;
        if (IS_PARTED_TYPE(t)) return 1;
    }
    return 0;
}
static void parted_view_release(void) {
    for (int i = 0; i < g_npx; i++) free(g_px[i]);
    g_npx = 0;
    for (int i = 0; i < 3; i++) { free(g_pxmem[i]); g_pxmem[i] = NULL; }
}
/* rows of one partition's vector: an mmapped ENUM is its index vector, an in-memory one the pair (core/util.h:105) */
static obj_p enum_indices(obj_p e) { return e->mmod == RFX_MMOD_INTERNAL ? RFX_AS_LIST(e)[1] : e; }
static obj_p parted_view(obj_p tab) {
    obj_p names = RFX_AS_LIST(tab)[0], cols = RFX_AS_LIST(tab)[1];
    if (cols->len > RFX_MAX_PROXY) return NULL;
    rfx_obj_t *fc = (rfx_obj_t *)calloc(1, sizeof(rfx_obj_t) + (size_t)cols->len * sizeof(obj_p));
    rfx_obj_t *ft = (rfx_obj_t *)calloc(1, sizeof(rfx_obj_t) + 2 * sizeof(obj_p));
    if (!fc || !ft) { free(fc); free(ft); return NULL; }
    g_pxmem[0] = fc;
    g_pxmem[1] = ft;
    fc->type = RFX_TYPE_LIST;
    fc->len = cols->len;
    ft->type = RFX_TYPE_TABLE;
    ft->len = 2;
    RFX_AS_LIST(ft)[0] = names;
    RFX_AS_LIST(ft)[1] = fc;
    for (int64_t i = 0; i < cols->len; i++) {
        obj_p c = RFX_AS_LIST(cols)[i];
        if (!IS_PARTED_TYPE(c->type)) { RFX_AS_LIST(fc)[i] = c; continue; }
        proxy_t *px = (proxy_t *)calloc(1, sizeof(proxy_t));
        if (!px) return NULL;
        g_px[g_npx++] = px;
        px->src = c;
        int64_t total = 0;
        if (c->type == RFX_TYPE_MAPCOMMON) {
            obj_p vals = RFX_AS_LIST(c)[0], cnts = RFX_AS_LIST(c)[1];
            px->kind = 2;
            px->vtype = vals->type;
            px->hdr.type = (vals->type == RFX_TYPE_DATE || vals->type == RFX_TYPE_I64) ? RFX_TYPE_I64 : RFX_TYPE_LIST; /* LIST: not usable */
            for (int64_t j = 0; j < cnts->len; j++) total += RFX_AS_I64(cnts)[j];
        } else {
            px->kind = 1;
            px->hdr.type = (int8_t)(c->type - RFX_TYPE_PARTEDLIST); /* LIST (0) for a generic parted list: not usable */
            for (int64_t j = 0; j < c->len; j++) {
                obj_p part = RFX_AS_LIST(c)[j];
                if (part->type != (int8_t)(c->type - RFX_TYPE_PARTEDLIST)) px->hdr.type = RFX_TYPE_LIST; /* mixed partition types: not usable -- but the row
                                                                                                          * count stays the table's (column 0 gives nrows) */
                total += (part->type == RFX_TYPE_ENUM) ? enum_indices(part)->len : part->len;
            }
        }
        px->hdr.len = total;
        RFX_AS_LIST(fc)[i] = (obj_p)px;
    }
    return (obj_p)ft;
}
/* checksum over everything a proxy's device copy is made from */
static uint64_t proxy_sum(const proxy_t *px) {
    uint64_t h = 0x6A09E667F3BCC908ULL;
    if (px->kind == 2) {
        obj_p vals = RFX_AS_LIST(px->src)[0], cnts = RFX_AS_LIST(px->src)[1];
        h ^= payload_sum(RFX_AS_RAW(vals), (size_t)vals->len * (vals->type == RFX_TYPE_DATE ? 4 : 8));
        return (h * 0x9E3779B97F4A7C15ULL) ^ payload_sum(RFX_AS_RAW(cnts), (size_t)cnts->len * 8);
    }
    for (int64_t j = 0; j < px->src->len; j++) {
        obj_p part = RFX_AS_LIST(px->src)[j];
        if (part->type == RFX_TYPE_ENUM) part = enum_indices(part);
        h = ((h ^ payload_sum(RFX_AS_RAW(part), (size_t)part->len * 8)) * 0x9E3779B97F4A7C15ULL) ^ (uint64_t)part->len;
    }
    return h;
}
static int proxy_upload(const proxy_t *px, void *dev) {
    int64_t off = 0;
    if (px->kind == 2) {
        obj_p vals = RFX_AS_LIST(px->src)[0], cnts = RFX_AS_LIST(px->src)[1];
        for (int64_t j = 0; j < cnts->len; j++) {
            const int64_t n = RFX_AS_I64(cnts)[j];
            const int64_t v = vals->type == RFX_TYPE_DATE ? (int64_t)((const int32_t *)RFX_AS_RAW(vals))[j] : RFX_AS_I64(vals)[j];
            int rc = rfx_hip_fill_i64(g_ctx, (int64_t *)dev + off, n, v);
            if (rc != RFX_OK) return rc;
            off += n;
        }
        return RFX_OK;
    }
    for (int64_t j = 0; j < px->src->len; j++) { /* every partition's column file goes straight into its slice */
        obj_p part = RFX_AS_LIST(px->src)[j];
        if (part->type == RFX_TYPE_ENUM) part = enum_indices(part);
        if (part->len) {
            int rc = rfx_hip_h2d_pipelined(g_ctx, (int64_t *)dev + off, RFX_AS_RAW(part), (size_t)part->len * 8);
            if (rc != RFX_OK) return rc;
        }
        off += part->len;
    }
    return RFX_OK;
}

/* device pointer of a host vector's payload (uploading it if needed) */
/* 4-byte integer columns (I32 / DATE / TIME): comparable on the device through a widened copy (rfx_hip_widen_i32) */
#define IS_I32_FAMILY(t) ((t) == RFX_TYPE_I32 || (t) == RFX_TYPE_DATE || (t) == RFX_TYPE_TIME)
/* host payload -> device copy: 8-byte and 1-byte columns as they are, 4-byte integers widened to 8 bytes on the device */
static int payload_upload_one(rfx_ctx_t *c, int type, void *dev, const void *host, int64_t len) {
    if (!IS_I32_FAMILY(type)) return rfx_hip_h2d_pipelined(c, dev, host, (size_t)len * (type == RFX_TYPE_B8 ? 1 : 8));
    void *raw = NULL;
    int rc = rfx_hip_malloc(c, &raw, (size_t)(len ? len : 1) * 4);
    if (rc != RFX_OK) return rc;
    rc = rfx_hip_h2d_pipelined(c, raw, host, (size_t)len * 4);
    if (rc == RFX_OK) rc = rfx_hip_widen_i32(c, (const int32_t *)raw, len, (int64_t *)dev);
    if (rc == RFX_OK) rc = rfx_hip_ctx_sync(c); /* (the raw block goes back to the pool: the widening must have read it) */
    rfx_hip_free(c, raw);
    return rc;
}
/* the whole payload, every shard its row range (rfx_exec_split) */
static int payload_upload(int type, void *const *devs, const void *host, int64_t len) {
    const int esz = type == RFX_TYPE_B8 ? 1 : (IS_I32_FAMILY(type) ? 4 : 8);
    int rc = RFX_OK;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        int64_t r0, n;
        rfx_exec_split(len, g_nshards, s, &r0, &n);
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
        if (n > 0) rc = payload_upload_one(g_ctxs[s], type, devs[s], (const char *)host + (size_t)r0 * esz, n);
    }
    if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    return rc;
}
static int shards_alloc(void **devs, int64_t len, size_t desz) {
    int rc = RFX_OK;
    for (int s = 0; s < RFX_MAX_SHARDS; s++) devs[s] = NULL;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        int64_t n;
        rfx_exec_split(len, g_nshards, s, NULL, &n);
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
        rc = rfx_hip_malloc(g_ctxs[s], &devs[s], (size_t)(n ? n : 1) * desz);
    }
    if (rc != RFX_OK)
        for (int s = 0; s < g_nshards; s++)
            if (devs[s]) { rfx_hip_ctx_bind_thread(g_ctxs[s]); rfx_hip_free(g_ctxs[s], devs[s]); devs[s] = NULL; }
    if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    return rc;
}
/* A DEVICE column handle: a vector header of ours (mmod RFX_MMOD_DEVICE) whose payload is not the cells but their device address(es) --
 * what a host that already keeps its columns in HBM (bench.py, the Python test host, a C host with its own loader) hands to the
 * operators in place of a host vector.  Borrowed memory: never uploaded, cached, validated or freed here. */
#define RFX_MMOD_DEVICE 0xde
typedef struct {
    const void *d[RFX_MAX_SHARDS]; /* d[s] NULL beyond the first: one allocation, shard s = d[0] + its row range (shards on one device) */
} devcol_t;
rfx_obj_p rfx_host_device_vector(int8_t type, int64_t len, const void *const *d_ptrs, int nptrs) {
    if (len < 0 || !d_ptrs || nptrs < 1 || nptrs > RFX_MAX_SHARDS) return NULL;
    rfx_obj_p o = rfx_host_vector(RFX_TYPE_I64, (int64_t)(sizeof(devcol_t) / 8));
    if (!o) return NULL;
    devcol_t *dc = (devcol_t *)RFX_AS_RAW(o);
    memset(dc, 0, sizeof(*dc));
    for (int i = 0; i < nptrs; i++) dc->d[i] = d_ptrs[i];
    o->mmod = RFX_MMOD_DEVICE;
    o->type = type < 0 ? (int8_t)-type : type;
    o->len = len;
    return o;
}
static int resident(obj_p col, int pin, const void **dev) {
    if (col->mmod == RFX_MMOD_DEVICE) {
        const devcol_t *dc = (const devcol_t *)RFX_AS_RAW(col);
        const int esz = (col->type == RFX_TYPE_B8) ? 1 : 8;
        void *devs[RFX_MAX_SHARDS];
        for (int s = 0; s < g_nshards; s++) {
            int64_t r0;
            rfx_exec_split(col->len, g_nshards, s, &r0, NULL);
            devs[s] = (s == 0 || dc->d[s]) ? (void *)dc->d[s] : (void *)((const char *)dc->d[0] + (size_t)r0 * esz);
        }
        if (IS_I32_FAMILY(col->type)) return RFX_EINVAL; /* (device columns are 8-byte or B8 cells) */
        *dev = devs[0];
        return qcol_add(devs);
    }
    const proxy_t *px = g_npx ? proxy_of(col) : NULL;
    if (px && g_nshards > 1) return RFX_ELIMIT; /* (parted views run on one shard: the caller hands such tables to the host) */
    const int narrow = !px && IS_I32_FAMILY(col->type);
    const int esz = (col->type == RFX_TYPE_B8) ? 1 : (narrow ? 4 : 8);
    const void *host = px ? (const void *)px->src : RFX_AS_RAW(col); /* a parted column is known by its LIST object */
    const size_t bytes = (size_t)col->len * esz;          /* of the HOST payload: what is validated */
    const size_t dbytes = (size_t)col->len * (narrow ? 8 : esz); /* of the device copy: what the budget counts */
    const int ktype = px ? 64 + col->type : col->type;
    int have_sum = 0;
    uint64_t sum = 0;
    for (int i = 0; i < g_nres; i++)
        if (g_res[i].host == host && g_res[i].len == col->len && g_res[i].type == ktype) {
            int track = 0;
            if (!g_res[i].pinned) { /* unpinned: prove the copy current */
                if (!px && g_res[i].tracked && sd_entry_clean(&g_res[i])) { /* by its pages (soft-dirty bits): nothing wrote there */
                    g_sd_hits++;
                    g_res[i].tick = ++g_tick;
                    g_res[i].epoch = g_epoch;
                    g_res[i].pinned |= pin;
                    g_stat[ST_CACHE_HITS]++;
                    *dev = g_res[i].dev;
                    return qcol_add(g_res[i].devs);
                }
                g_res[i].tracked = 0;
                /* by its checksum -- taken AFTER the pages were clean-marked, so that it can vouch for them from now on */
                track = !px && !g_res[i].sd_never && g_res[i].stable >= SD_STABLE_USES && sd_usable(host, bytes) && sd_call_clear() == 0;
                sum = px ? proxy_sum(px) : payload_sum(host, bytes);
                have_sum = 1;
            }
            if (track) {
                uintptr_t lo, hi;
                sd_interior(host, bytes, &lo, &hi);
                if (sd_scan(lo, hi) < 0) g_res[i].sd_never = 1, track = 0; /* file-backed / shared pages: never by soft-dirty bits */
                else {
                    g_res[i].tracked = 1; /* (a write since the clear shows at the next use and costs one more checksum) */
                    g_res[i].edge_sum = sd_edge_sum(host, bytes);
                }
            }
            if (!g_res[i].pinned) g_res[i].stable = (g_res[i].sum == sum) ? g_res[i].stable + 1 : 0;
            if (g_res[i].pinned || g_res[i].sum == sum) {
                g_res[i].tick = ++g_tick;
                g_res[i].epoch = g_epoch;
                g_res[i].pinned |= pin;
                g_stat[ST_CACHE_HITS]++;
                *dev = g_res[i].dev;
                return qcol_add(g_res[i].devs);
            }
            /* stale: the payload changed under the same address -- refresh the device copy in place */
            g_stat[ST_CACHE_STALE]++;
            int rc = px ? proxy_upload(px, g_res[i].dev) : payload_upload(col->type, g_res[i].devs, host, col->len);
            if (rc != RFX_OK) { res_free(i); return rc; }
            g_stat[ST_UPLOADS]++;
            g_res[i].scope_ok = 0; /* (new cells: the scope remembered for the old ones is gone) */
            g_res[i].sum = sum;
            g_res[i].tick = ++g_tick;
            g_res[i].epoch = g_epoch;
            g_res[i].pinned |= pin;
            *dev = g_res[i].dev;
            return qcol_add(g_res[i].devs);
        }
    while (g_nres && g_res_bytes + dbytes > cache_budget()) {
        int victim = -1; /* least recently used, not pinned, not in use by the call in flight */
        for (int i = 0; i < g_nres; i++)
            if (!g_res[i].pinned && g_res[i].epoch != g_epoch && (victim < 0 || g_res[i].tick < g_res[victim].tick)) victim = i;
        if (victim < 0) break; /* everything left is pinned or in use: go over budget rather than free what the call reads */
        res_free(victim);
    }
    void *devs[RFX_MAX_SHARDS];
    int rc = shards_alloc(devs, col->len, narrow ? 8 : (size_t)esz);
    if (rc != RFX_OK) return rc;
    /* the checksum, THEN the copy: a host write racing with this call is either in both, or in the copy only and costs one refresh at
     * the next use -- never a device copy older than what vouches for it */
    if (!have_sum) sum = px ? proxy_sum(px) : payload_sum(host, bytes);
    rc = px ? proxy_upload(px, devs[0]) : payload_upload(col->type, devs, host, col->len); /* heap vector or mmapped column file alike: staged through pinned buffers */
    if (rc != RFX_OK) {
        for (int s = 0; s < g_nshards; s++) { if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]); rfx_hip_free(g_ctxs[s], devs[s]); }
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
        return rc;
    }
    g_stat[ST_UPLOADS]++;
    if (g_nres == g_capres) {
        g_capres = g_capres ? g_capres * 2 : 32;
        g_res = (resident_t *)realloc(g_res, sizeof(resident_t) * (size_t)g_capres);
    }
    resident_t e;
    memset(&e, 0, sizeof(e));
    e.host = host; e.len = col->len; e.type = ktype; e.sum = sum; e.dev = devs[0]; e.bytes = bytes; e.pinned = pin; e.tick = ++g_tick; e.epoch = g_epoch; e.dbytes = dbytes;
    for (int s = 0; s < g_nshards; s++) e.devs[s] = devs[s];
    g_res[g_nres++] = e; /* (page tracking starts once the column has proven stable) */
    g_res_bytes += dbytes;
    *dev = devs[0];
    return qcol_add(devs);
}
/* The key scope of a WHOLE resident column (index_scope_i64 without a filter, core/index.c:376-435), remembered with the copy it was taken
 * from.  A group-by over a few thousand slots is two host round trips -- the scope, the result -- and ~25 us each: the remembered scope
 * (a superset of any filtered selection's, which is all the tables' sizing needs) saves the first one for every later query over that key
 * column, whatever its filter.  Only entries proven current in THIS operator call are asked (epoch), a refreshed copy forgets its scope. */
static resident_t *resident_entry(const void *dev) {
    for (int i = 0; i < g_nres; i++)
        if (g_res[i].dev == dev && g_res[i].epoch == g_epoch) return &g_res[i];
    return NULL;
}
/* drop every cached copy that overlaps the vector's payload */
static void invalidate_payload(obj_p v) {
    if (v && IS_PARTED_TYPE(v->type)) { /* a parted column: cached under its LIST object */
        for (int i = 0; i < g_nres;) {
            if (g_res[i].host == (const void *)v) res_free(i);
            else i++;
        }
        return;
    }
    if (!v || v->type <= 0) return;
    const int esz = (v->type == RFX_TYPE_B8) ? 1 : (IS_I32_FAMILY(v->type) ? 4 : 8);
    const char *lo = (const char *)RFX_AS_RAW(v), *hi = lo + (size_t)v->len * esz;
    for (int i = 0; i < g_nres;) {
        const char *a = (const char *)g_res[i].host, *b = a + g_res[i].bytes;
        if (a < hi && lo < b) res_free(i);
        else i++;
    }
}

static int col_ctype(obj_p c) {
    switch (c->type) {
        case RFX_TYPE_I64: case RFX_TYPE_TIMESTAMP: case RFX_TYPE_SYMBOL: return RFX_I64; /* 8-byte integer payloads */
        case RFX_TYPE_F64: return RFX_F64;
        default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------------ table access */
static obj_p table_col(obj_p tab, int64_t sym) {
    obj_p names = RFX_AS_LIST(tab)[0], cols = RFX_AS_LIST(tab)[1];
    for (int64_t i = 0; i < names->len; i++)
        if (RFX_AS_I64(names)[i] == sym) return RFX_AS_LIST(cols)[i];
    return NULL;
}
static obj_p dict_get(obj_p d, const char *key) {
    int64_t id = H.intern(key, (int64_t)strlen(key));
    obj_p keys = RFX_AS_LIST(d)[0], vals = RFX_AS_LIST(d)[1];
    for (int64_t i = 0; i < keys->len; i++)
        if (RFX_AS_I64(keys)[i] == id) return RFX_AS_LIST(vals)[i];
    return NULL;
}

/* ------------------------------------------------------------------------------------------------ planning */
typedef struct {
    rfx_pred_t preds[RFX_MAX_PREDS];
    int npred, logic;
} wplan_t;

/* one comparison `(op colsym atom|colsym)` -> descriptor; 0 ok, -1 unsupported shape */
/* device scratch a query's PREDICATES allocate (operands that are expressions): released at the end of rfx_select */
static struct { void *d[RFX_MAX_SHARDS]; } g_qtmp[2 * RFX_MAX_PREDS * 4]; /* (per shard: every shard evaluates its own rows) */
static int g_nqtmp;
static void qtmp_release(void) {
    for (int i = 0; i < g_nqtmp; i++)
        for (int s = 0; s < g_nshards; s++) {
            if (!g_qtmp[i].d[s]) continue;
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
            rfx_hip_free(g_ctxs[s], g_qtmp[i].d[s]);
        }
    if (g_nqtmp && g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    g_nqtmp = 0;
}
static int build_xnodes(obj_p tab, obj_p e, rfx_xnode_t *nodes, int *nn, int *ncols, const char **why);
/* a comparison operand that is an element-wise expression (op x y): the reference evaluates it first (eval -> binop_map), so do
 * we -- one pass into a scratch column (rfx_hip_eval_expr), then the comparison reads it like any column */
static int expr_operand(obj_p tab, obj_p e, const void **d, int *ctype) {
    rfx_xnode_t nodes[RFX_MAX_XNODES];
    int nn = 0, ncols = 0;
    const char *why = NULL;
    int top = build_xnodes(tab, e, nodes, &nn, &ncols, &why);
    if (top == -2) return -2;
    if (top < 0 || ncols == 0 || g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) return -1;
    obj_p tcols = RFX_AS_LIST(tab)[1];
    const int64_t nrows = tcols->len ? RFX_AS_LIST(tcols)[0]->len : 0;
    rfx_agg_t a;
    memset(&a, 0, sizeof(a));
    a.kind = RFX_AGG_SUM;
    a.col_type = RFX_I64;
    a.nxnodes = nn;
    a.xnodes = nodes;
    int32_t ot = RFX_I64;
    /* every shard evaluates ITS rows of the operand columns on its own context (a shard holds its row range only: one evaluation over
     * the whole length would read past shard 0's piece); the scratch column then is a column of the query like any other (qcol_add) */
    memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
    void **devs = g_qtmp[g_nqtmp++].d;
    int rc = RFX_OK;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        rfx_xnode_t mine[RFX_MAX_XNODES];
        int64_t n = nrows;
        if (g_nshards > 1) {
            rfx_exec_split(nrows, g_nshards, s, NULL, &n);
            for (int j = 0; j < nn; j++) {
                mine[j] = nodes[j];
                rfx_xoperand_t *o[2] = {&mine[j].l, &mine[j].r};
                for (int k = 0; k < 2; k++) {
                    if (o[k]->kind != RFX_XK_COL) continue;
                    const void *there = NULL;
                    for (int i = 0; i < g_nqcols && !there; i++)
                        if (g_qcols[i].d[0] == o[k]->d_col) there = g_qcols[i].d[s];
                    if (!there) rc = RFX_EINVAL; /* (cannot happen: build_xnodes made every column resident, shard by shard) */
                    o[k]->d_col = there;
                }
            }
            a.xnodes = mine;
            rfx_hip_ctx_bind_thread(g_ctxs[s]);
        }
        if (rc == RFX_OK) rc = rfx_hip_malloc(g_ctxs[s], &devs[s], (size_t)(n ? n : 1) * 8);
        if (rc == RFX_OK) rc = rfx_hip_eval_expr(g_ctxs[s], &a, n, devs[s], &ot);
    }
    if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    if (rc != RFX_OK || qcol_add(devs) != RFX_OK) return -2;
    *d = devs[0];
    *ctype = ot;
    return 0;
}
/* the SYMBOL vector an ENUM column indexes: the global its key names (in-memory pair: the key symbol; mmapped: the key's characters sit
 * one page before the indices, core/util.h:103-105, core/binary.c:135-137).  NULL when it does not resolve; the caller drops it. */
static obj_p enum_domain(obj_p e) {
    int64_t key_id;
    if (e->mmod == RFX_MMOD_INTERNAL) key_id = RFX_AS_LIST(e)[0]->i64;
    else {
        const char *ks = (const char *)e - 4096 + sizeof(rfx_obj_t);
        key_id = H.intern(ks, (int64_t)strnlen(ks, 4096 - sizeof(rfx_obj_t)));
    }
    obj_p ka = H.i64(key_id);
    ka->type = -RFX_TYPE_SYMBOL;
    obj_p dom = H.eval(ka);
    H.drop(ka);
    if (dom && dom->type != RFX_TYPE_SYMBOL) {
        H.drop(dom);
        dom = NULL;
    }
    return dom;
}
#define RFX_ATTR_QUOTED 8 /* ATTR_QUOTED, core/ops.h:55: a symbol atom that stands for itself ('x), not for a column */
static int g_where_virtual, g_where_data; /* comparisons of the where: in flight that read the virtual column / data columns of a parted table */
static int plan_cmp(obj_p tab, obj_p e, rfx_pred_t *p) {
    if (e->type == RFX_TYPE_LIST && e->len == 2 && fn_id(RFX_AS_LIST(e)[0]) == F_NOT) {
        /* (not (cmp x y)) = the complementary comparison: the reference's order is total (nulls and NaN sort lowest, core/ops.h:97), so
         * exactly one of < == > holds for every pair of cells and the complement of a set of them is the rest */
        static const int COMPLEMENT[6] = {RFX_NE, RFX_EQ, RFX_GE, RFX_LE, RFX_GT, RFX_LT}; /* of EQ NE LT GT LE GE */
        const int rc = plan_cmp(tab, RFX_AS_LIST(e)[1], p);
        if (rc == 0) p->op = COMPLEMENT[p->op];
        return rc;
    }
    if (e->type != RFX_TYPE_LIST || e->len != 3) return -1;
    int f = fn_id(RFX_AS_LIST(e)[0]);
    if (f < F_EQ || f > F_GE) return -1;
    obj_p l = RFX_AS_LIST(e)[1], r = RFX_AS_LIST(e)[2];
    memset(p, 0, sizeof(*p));
    p->op = f - F_EQ; /* F_EQ..F_GE are in RFX_EQ..RFX_GE order */
    const void *d;
    int64_t llen = -1;
    int lvirt = 0, ldate = 0;
    if (l->type == RFX_TYPE_LIST) {
        int ct = RFX_I64, rc0 = expr_operand(tab, l, &d, &ct);
        if (rc0) return rc0;
        p->col_type = ct;
    } else {
        if (l->type != -RFX_TYPE_SYMBOL) return -1;
        obj_p lc = table_col(tab, l->i64);
        if (lc && lc->type == RFX_TYPE_ENUM) {
            /* (== enum-column 'sym): the reference compares the domain's symbol at every index with the atom (MTYPE2(TYPE_ENUM, -TYPE_SYMBOL),
             * core/cmp.c:260-281); the symbol's place in the domain is found once on the host and the INDEX column is compared on the
             * device -- a symbol the domain does not hold selects nothing (index -1).  Only == : the other operators are the host's. */
            if (f != F_EQ || r->type != -RFX_TYPE_SYMBOL || !(r->attrs & RFX_ATTR_QUOTED)) return -1;
            if (lc->mmod != RFX_MMOD_INTERNAL) return -1; /* an mmapped enum (splayed table): the reference's own `where:` answers `type` there -- the host's to say */
            obj_p dom = enum_domain(lc);
            if (!dom) return -1;
            int64_t at = -1;
            for (int64_t i = 0; i < dom->len && at < 0; i++)
                if (RFX_AS_I64(dom)[i] == r->i64) at = i;
            H.drop(dom);
            if (resident(enum_indices(lc), 0, &d) != RFX_OK) return -2;
            g_where_data++;
            p->d_col = d;
            p->col_type = RFX_I64;
            p->rhs_type = RFX_I64;
            p->rhs_i = at;
            return 0;
        }
        if (lc && IS_I32_FAMILY(lc->type) && !(g_npx && proxy_of(lc))) { /* (a parted table's 4-byte columns are the host's: proxies upload 8-byte partitions only) */
            /* a 4-byte integer column (I32 / DATE / TIME) in a comparison: its widened device copy against an atom or a column of the
             * types the reference's i32 arms take (core/cmp.c:148-166: the same 4-byte type; for I32 also I64 / F64, promoted as
             * i32_to_i64 / i32_to_f64 do -- which is what the widened column compares as) */
            if (resident(lc, 0, &d) != RFX_OK) return -2;
            g_where_data++;
            p->d_col = d;
            p->col_type = RFX_I64;
            const int8_t lt = lc->type;
            if (r->type == -lt) { p->rhs_type = RFX_I64; p->rhs_i = r->i32 == INT32_MIN ? RFX_NULL_I64 : (int64_t)r->i32; return 0; }
            if (lt == RFX_TYPE_I32 && r->type == -RFX_TYPE_I64) { p->rhs_type = RFX_I64; p->rhs_i = r->i64; return 0; }
            if (lt == RFX_TYPE_I32 && r->type == -RFX_TYPE_F64) { p->rhs_type = RFX_F64; p->rhs_f = r->f64; return 0; }
            if (r->type == -RFX_TYPE_SYMBOL && !(r->attrs & RFX_ATTR_QUOTED)) {
                obj_p rc = table_col(tab, r->i64);
                if (!rc || rc->len != lc->len) return -1;
                if (!(rc->type == lt || (lt == RFX_TYPE_I32 && (rc->type == RFX_TYPE_I64 || rc->type == RFX_TYPE_F64)))) return -1;
                if (resident(rc, 0, &d) != RFX_OK) return -2;
                p->d_rhs_col = d;
                p->rhs_type = rc->type == RFX_TYPE_F64 ? RFX_F64 : RFX_I64;
                return 0;
            }
            return -1;
        }
        if (!lc || !col_ctype(lc)) return -1;
        p->col_type = col_ctype(lc);
        if (resident(lc, 0, &d) != RFX_OK) return -2;
        llen = lc->len;
        const proxy_t *px = g_npx ? proxy_of(lc) : NULL;
        lvirt = px && px->kind == 2;
        ldate = lvirt && px->vtype == RFX_TYPE_DATE;
    }
    if (lvirt) g_where_virtual++;
    else g_where_data++;
    p->d_col = d;
    if (r->type == -RFX_TYPE_I64) { p->rhs_type = RFX_I64; p->rhs_i = r->i64; }
    else if (r->type == -RFX_TYPE_TIMESTAMP && l->type == -RFX_TYPE_SYMBOL && table_col(tab, l->i64) && table_col(tab, l->i64)->type == RFX_TYPE_TIMESTAMP) {
        p->rhs_type = RFX_I64; /* a TIMESTAMP column against a timestamp atom: nanoseconds as i64 on both sides (core/cmp.c) */
        p->rhs_i = r->i64;
    }
    else if (r->type == -RFX_TYPE_DATE && ldate) { p->rhs_type = RFX_I64; p->rhs_i = (int64_t)r->i32; } /* (== Date 2024.01.03): partition pruning, core/cmp.c:341-358 */
    else if (r->type == -RFX_TYPE_F64) { p->rhs_type = RFX_F64; p->rhs_f = r->f64; }
    else if (r->type == -RFX_TYPE_SYMBOL && (r->attrs & RFX_ATTR_QUOTED)) {
        /* a quoted symbol is a value, never a column name -- even when the table has a column of that name (eval_sym, core/eval.c:829):
         * a SYMBOL column compares its interned ids with it (== and != ; the ordering of symbols is the host's business) */
        obj_p lc = (l->type == -RFX_TYPE_SYMBOL) ? table_col(tab, l->i64) : NULL;
        if (!lc || lc->type != RFX_TYPE_SYMBOL || (f != F_EQ && f != F_NE)) return -1;
        p->rhs_type = RFX_I64;
        p->rhs_i = r->i64;
    } else if (r->type == -RFX_TYPE_SYMBOL) {
        obj_p rc = table_col(tab, r->i64);
        if (!rc || !col_ctype(rc) || (llen >= 0 && rc->len != llen)) return -1;
        g_where_data++;
        if (resident(rc, 0, &d) != RFX_OK) return -2;
        p->d_rhs_col = d;
        p->rhs_type = col_ctype(rc);
    } else if (r->type == RFX_TYPE_LIST) {
        int ct = RFX_I64, rc0 = expr_operand(tab, r, &d, &ct);
        if (rc0) return rc0;
        p->d_rhs_col = d;
        p->rhs_type = ct;
    } else return -1;
    return 0;
}
/* (within col [lo hi]) = lo <= col <= hi (ray_within, core/items.c:848-872: an I64 column against a two-element I64 vector, raw integer
 * order) and (in col [v1 .. vn]) = col == v1 or ... (ray_in, core/items.c:736+ -> index_in_i64_i64: raw equality; I64 / TIMESTAMP / SYMBOL
 * columns against a vector of their own type) as comparisons of the fused pass: appends them to out[0 .. room) and says through *glogic
 * how they combine among themselves.  Returns how many (>= 1), -1 when `e` is not such a form (or too long), -2 on an upload error. */
static int plan_set_cmp(obj_p tab, obj_p e, rfx_pred_t *out, int room, int *glogic) {
    if (!e || e->type != RFX_TYPE_LIST || e->len != 3) return -1;
    const int f = fn_id(RFX_AS_LIST(e)[0]);
    if (f != F_IN && f != F_WITHIN) return -1;
    obj_p l = RFX_AS_LIST(e)[1], r = RFX_AS_LIST(e)[2];
    if (l->type != -RFX_TYPE_SYMBOL || (l->attrs & RFX_ATTR_QUOTED) || r->type <= 0) return -1;
    obj_p lc = table_col(tab, l->i64);
    if (!lc || g_npx) return -1; /* (parted tables: the reference prunes partitions through these forms -- not taken apart here) */
    int n;
    if (f == F_WITHIN) {
        if (lc->type != RFX_TYPE_I64 || r->type != RFX_TYPE_I64 || r->len != 2) return -1;
        n = 2;
        *glogic = RFX_AND;
    } else {
        if (!(lc->type == RFX_TYPE_I64 || lc->type == RFX_TYPE_TIMESTAMP || lc->type == RFX_TYPE_SYMBOL) || r->type != lc->type || r->len < 1 || r->len > RFX_MAX_PREDS) return -1;
        n = (int)r->len;
        *glogic = RFX_OR;
    }
    if (n > room) return -1;
    const void *d;
    if (resident(lc, 0, &d) != RFX_OK) return -2;
    g_where_data++;
    for (int i = 0; i < n; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        out[i].d_col = d;
        out[i].col_type = RFX_I64;
        out[i].rhs_type = RFX_I64;
        out[i].rhs_i = RFX_AS_I64(r)[i];
        out[i].op = f == F_WITHIN ? (i == 0 ? RFX_GE : RFX_LE) : RFX_EQ;
    }
    return n;
}
/* where: a comparison, or ANY tree of and / or over comparisons (logic_map nests freely, core/logic.c:89-260) -- its leaves in order, each
 * with the depth of parentheses it sits in and the parentheses that close after it.  Level 0 combines with the root's operator, every
 * deeper level with the opposite of the level above: the same operator nested in itself is associative and stays on its level.  Up to
 * RFX_MAX_PREDS comparisons and four levels run in ONE fused pass (rfx_pred_t: the two-level `more` form where it suffices -- the
 * kernels' short path -- else the RFX_PRED_TREE form); anything beyond: -1 (the mask path answers it). */
typedef struct {
    int dep[RFX_MAX_PREDS], clo[RFX_MAX_PREDS];
} wtree_t;
static int plan_node(obj_p tab, obj_p e, int level_op, int depth, wplan_t *wp, wtree_t *wt) {
    if (!e || e->type != RFX_TYPE_LIST || e->len < 1) return -1;
    const int f = fn_id(RFX_AS_LIST(e)[0]);
    if (f == F_IN || f == F_WITHIN) { /* a group of comparisons: on this level when it combines like it, else a parenthesis of its own */
        int gl = RFX_AND;
        const int n = plan_set_cmp(tab, e, &wp->preds[wp->npred], RFX_MAX_PREDS - wp->npred, &gl);
        if (n < 0) return n;
        const int own = n > 1 && gl != (level_op == F_AND ? RFX_AND : RFX_OR);
        for (int i = 0; i < n; i++) {
            wt->dep[wp->npred + i] = depth + own;
            wt->clo[wp->npred + i] = 0;
        }
        if (own) wt->clo[wp->npred + n - 1] = 1;
        wp->npred += n;
        return 0;
    }
    if (f != F_AND && f != F_OR) {
        if (wp->npred >= RFX_MAX_PREDS) return -1;
        const int rc = plan_cmp(tab, e, &wp->preds[wp->npred]);
        if (rc) return rc;
        wt->dep[wp->npred] = depth;
        wt->clo[wp->npred] = 0;
        wp->npred++;
        return 0;
    }
    if (e->len < 2) return -1;
    const int own = f != level_op; /* the opposite operator: a parenthesis one level down, closed after its last leaf */
    const int first = wp->npred;
    for (int64_t i = 1; i < e->len; i++) {
        const int rc = plan_node(tab, RFX_AS_LIST(e)[i], f, depth + own, wp, wt);
        if (rc) return rc;
    }
    if (own && wp->npred > first) wt->clo[wp->npred - 1]++;
    return 0;
}
static int plan_where(obj_p tab, obj_p w, wplan_t *wp) {
    wp->npred = 0;
    wp->logic = RFX_AND;
    if (!w) return 0;
    if (w->type != RFX_TYPE_LIST || w->len < 1) return -1;
    const int f = fn_id(RFX_AS_LIST(w)[0]);
    wtree_t wt;
    if (f == F_AND || f == F_OR) {
        if (w->len < 2) return -1;
        wp->logic = (f == F_AND) ? RFX_AND : RFX_OR;
    } else if (f == F_IN || f == F_WITHIN) {
        const int n = plan_set_cmp(tab, w, wp->preds, RFX_MAX_PREDS, &wp->logic);
        if (n < 0) return n;
        wp->npred = n;
        return 0;
    }
    const int rc = plan_node(tab, w, (f == F_AND || f == F_OR) ? f : F_AND, 0, wp, &wt);
    if (rc) return rc;
    int maxd = 0;
    for (int i = 0; i < wp->npred; i++) {
        if (wt.dep[i] > maxd) maxd = wt.dep[i];
        if (wt.clo[i] > wt.dep[i]) return -1; /* (cannot happen: a parenthesis closes on the level it opened) */
    }
    if (maxd > 3 || (maxd > 1 && wp->npred < 3)) return -1; /* deeper than four levels (or a degenerate nest of one-armed parentheses): through masks */
    if (maxd <= 1) { /* flat, or parentheses of the opposite operator over comparisons: the two-level form */
        for (int i = 0; i < wp->npred; i++) wp->preds[i].more = (wt.dep[i] == 1 && wt.clo[i] == 0) ? 1 : 0;
        return 0;
    }
    for (int i = 0; i < wp->npred; i++) wp->preds[i].more = RFX_PRED_LEAF(wt.dep[i], wt.clo[i]);
    return 0;
}


/* ---- nested boolean trees: evaluated the way the reference does (mask per comparison, and/or in place, where), but on
 * the GPU: core/cmp.c -> K2 rfx_hip_cmp_mask, core/logic.c -> rfx_hip_mask_logic, core/ops.c:254 -> K3 ---- */
static int mask_of_expr(obj_p tab, obj_p e, int64_t nrows, int8_t **out) {
    *out = NULL;
    if (!e || e->type != RFX_TYPE_LIST || e->len < 2) return -1;
    int f = fn_id(RFX_AS_LIST(e)[0]);
    void *m = NULL;
    if (f >= F_EQ && f <= F_GE) {
        rfx_pred_t p;
        int rc = plan_cmp(tab, e, &p);
        if (rc) return rc;
        if (rfx_hip_malloc(g_ctx, &m, (size_t)nrows + 16) != RFX_OK) return -2;
        if (rfx_hip_cmp_mask(g_ctx, &p, nrows, (int8_t *)m) != RFX_OK) { rfx_hip_free(g_ctx, m); return -2; }
        *out = (int8_t *)m;
        return 0;
    }
    if (f != F_AND && f != F_OR) return -1;
    int8_t *acc = NULL;
    for (int64_t i = 1; i < e->len; i++) {
        int8_t *sub = NULL;
        int rc = mask_of_expr(tab, RFX_AS_LIST(e)[i], nrows, &sub);
        if (rc) { if (acc) rfx_hip_free(g_ctx, acc); return rc; }
        if (!acc) acc = sub;
        else {
            rc = rfx_hip_mask_logic(g_ctx, f == F_AND ? RFX_AND : RFX_OR, acc, sub, 0, nrows);
            rfx_hip_free(g_ctx, sub);
            if (rc != RFX_OK) { rfx_hip_free(g_ctx, acc); return -2; }
        }
    }
    *out = acc;
    return 0;
}

/* selection of `where` as ascending device row ids (flat predicates fused, nested trees through masks) */
static int where_ids(obj_p tab, obj_p where, const wplan_t *wp, int flat, int64_t nrows, int64_t **d_ids, int64_t *count) {
    *d_ids = NULL;
    *count = 0;
    int8_t *mask = NULL;
    int rc;
    if (flat) {
        /* one pass over the predicate columns (rfx_where_once.hip): buffer by sampled estimate, exact count back, a second run if the
         * sample underestimated a clustered selection */
        int64_t cap = 0;
        void *d = NULL;
        if (rfx_hip_where_estimate(g_ctx, wp->preds, wp->npred, wp->logic, nrows, &cap) != RFX_OK) return -2;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (cap > 0 && rfx_hip_malloc(g_ctx, &d, (size_t)cap * 8) != RFX_OK) return -2;
            const int wrc = rfx_hip_where_once(g_ctx, wp->preds, wp->npred, wp->logic, nrows, 0, (int64_t *)d, cap, count);
            if (wrc == RFX_OK) {
                if (*count > 0) *d_ids = (int64_t *)d;
                else if (d) rfx_hip_free(g_ctx, d);
                return 0;
            }
            if (d) rfx_hip_free(g_ctx, d);
            d = NULL;
            if (wrc != RFX_ELIMIT || *count <= cap) break;
            cap = *count;
        }
        *count = 0;
        return -2;
    } else {
        rc = mask_of_expr(tab, where, nrows, &mask);
        if (rc == 0) rc = rfx_hip_where_begin(g_ctx, NULL, 0, RFX_AND, mask, nrows, count) == RFX_OK ? 0 : -2;
    }
    if (rc == 0 && *count > 0) {
        void *d = NULL;
        if (rfx_hip_malloc(g_ctx, &d, (size_t)*count * 8) != RFX_OK || rfx_hip_where_emit(g_ctx, 0, (int64_t *)d) != RFX_OK) {
            if (d) rfx_hip_free(g_ctx, d);
            rc = -2;
        } else *d_ids = (int64_t *)d;
    }
    if (mask) rfx_hip_free(g_ctx, mask);
    return rc;
}

static obj_p value_atom(const rfx_value_t *v) { return v->type == RFX_F64 ? H.f64(v->f) : H.i64(v->i); }
static obj_p one_row(const rfx_value_t *v) {
    obj_p c = H.vector(v->type == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64, 1);
    RFX_AS_I64(c)[0] = v->i;
    return c;
}

static obj_p refused1(int f, obj_p x) {
    if (g_refused_sharded && H.bound == 1 && f >= 0 && f < F_N && H.f[f]) return HOST_CALL(((rfx_unary_f)H.f[f])(x));
    return fail_ctx();
}
static obj_p refused2(int f, obj_p x, obj_p y) {
    if (g_refused_sharded && H.bound == 1 && f >= 0 && f < F_N && H.f[f]) return HOST_CALL(((rfx_binary_f)H.f[f])(x, y));
    return fail_ctx();
}
static obj_p refusedn(int f, obj_p *x, int64_t n) {
    if (g_refused_sharded && H.bound == 1 && f >= 0 && f < F_N && H.f[f]) return HOST_CALL(((rfx_vary_f)H.f[f])(x, n));
    return fail_ctx();
}
static obj_p delegate_select(obj_p dict, const char *why) {
    g_last_gpu = 0;
    snprintf(g_err, sizeof(g_err), "rfx_select: handed to the host (%s)", why); /* rfx_ops_last_error(): why the last query was delegated */
    if (getenv("RFX_TRACE")) fprintf(stderr, "[rfx] select delegated: %s\n", why);
    if (H.bound == 1 && H.f[F_SELECT]) return HOST_CALL(((rfx_unary_f)H.f[F_SELECT])(dict));
    char b[300];
    snprintf(b, sizeof(b), "rfx_select: query shape not covered by the MI355X path (%s) and no host ray_select to delegate to", why);
    return fail(b);
}

/* (op x y) with x / y a column symbol, an i64 / f64 atom or another such list -> nodes in evaluation order (rfx_xnode_t).
 * Returns the index of the node holding the value, -1 with *why set when the shape is not covered, -2 on an upload error. */
static int build_xnodes(obj_p tab, obj_p e, rfx_xnode_t *nodes, int *nn, int *ncols, const char **why) {
    if (e->type != RFX_TYPE_LIST || e->len != 3) { *why = "expression is not (op x y)"; return -1; }
    int xf = fn_id(RFX_AS_LIST(e)[0]);
    if (xf < F_ADD || xf > F_MOD) { *why = "expression operator is not + - * div / %"; return -1; }
    rfx_xnode_t node;
    memset(&node, 0, sizeof(node));
    node.op = RFX_X_ADD + (xf - F_ADD);
    rfx_xoperand_t *ops[2] = {&node.l, &node.r};
    for (int j = 0; j < 2; j++) {
        obj_p x = RFX_AS_LIST(e)[1 + j];
        if (x->type == -RFX_TYPE_SYMBOL) {
            obj_p c = table_col(tab, x->i64);
            if (!c || !(c->type == RFX_TYPE_I64 || c->type == RFX_TYPE_F64)) { *why = "expression operand column type"; return -1; }
            const void *d;
            if (resident(c, 0, &d) != RFX_OK) return -2;
            ops[j]->kind = RFX_XK_COL;
            ops[j]->type = col_ctype(c);
            ops[j]->d_col = d;
            (*ncols)++;
        } else if (x->type == -RFX_TYPE_I64) {
            ops[j]->kind = RFX_XK_ATOM;
            ops[j]->type = RFX_I64;
            ops[j]->i = x->i64;
        } else if (x->type == -RFX_TYPE_F64) {
            ops[j]->kind = RFX_XK_ATOM;
            ops[j]->type = RFX_F64;
            ops[j]->f = x->f64;
        } else if (x->type == RFX_TYPE_LIST) {
            int sub = build_xnodes(tab, x, nodes, nn, ncols, why);
            if (sub < 0) return sub;
            ops[j]->kind = RFX_XK_NODE;
            ops[j]->node = sub;
        } else { *why = "expression operand is neither a column, an i64/f64 atom nor an expression"; return -1; }
    }
    if (*nn >= RFX_MAX_XNODES) { *why = "expression deeper than RFX_MAX_XNODES operations"; return -1; }
    nodes[*nn] = node;
    return (*nn)++;
}

/* ------------------------------------------------------------------------------------------------ select */
/* RFX_TRACE=2: where a select's wall time goes (microseconds between marks), one line per query on stderr */
static double g_tm[12];
static int g_ntm;
static void tm_mark(void) {
    if (g_ntm < 12) {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        g_tm[g_ntm++] = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    }
}
static void tm_print(void) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("RFX_TRACE"); on = e && atoi(e) >= 2; }
    if (on && g_ntm > 1) {
        fprintf(stderr, "[rfx] select us:");
        for (int i = 1; i < g_ntm; i++) fprintf(stderr, " %.0f", g_tm[i] - g_tm[i - 1]);
        fprintf(stderr, "  (plan | scope | tables+pass+rank | emit | fetch | build+free) total %.0f\n", g_tm[g_ntm - 1] - g_tm[0]);
    }
}
/* ---- rfx_select, piece by piece.  Every piece answers SEL_GO (carry on), SEL_OUT (*why says which shape the host must answer) or
 * SEL_DONE (an error / a finished result: the caller stops). ---- */
enum { SEL_GO = 0, SEL_OUT = 1, SEL_DONE = 2 };
/* the output mappings {name: (aggr column | expression)} of a select dict (everything but from: where: by: take:) as aggregate
 * descriptors over resident device columns */
typedef struct {
    rfx_agg_t aggs[RFX_MAX_AGGS];
    rfx_xnode_t xnodes[RFX_MAX_AGGS][RFX_MAX_XNODES];
    int64_t names[RFX_MAX_AGGS];
    int outtype[RFX_MAX_AGGS];
    int nagg;
} sel_maps_t;
/* result cells of an aggregate over a widened 4-byte column, back in the column's own width: the i64 null and the i64 identities of an
 * all-null group (core/aggr.c:1246) become the 4-byte ones */
static void sel_narrow_i32(obj_p col, const int64_t *cells, int64_t n, int kind) {
    int32_t *o = (int32_t *)RFX_AS_RAW(col);
    if (kind == RFX_AGG_SUM) { /* FOLD_ADDI32's result IS the low half of the 64-bit sum, whatever that sum is (no null / identity to translate) */
        for (int64_t i = 0; i < n; i++) o[i] = (int32_t)(uint32_t)(uint64_t)cells[i];
        return;
    }
    for (int64_t i = 0; i < n; i++) o[i] = cells[i] == RFX_NULL_I64 ? INT32_MIN : (cells[i] == INT64_MAX ? INT32_MAX : (int32_t)cells[i]);
}
static int sel_mappings(obj_p tab, obj_p dkeys, obj_p dvals, int grouped, sel_maps_t *M, const char **why) {
    const int64_t s_from = H.intern("from", 4), s_where = H.intern("where", 5), s_by = H.intern("by", 2), s_take = H.intern("take", 4);
    M->nagg = 0;
    for (int64_t i = 0; i < dkeys->len; i++) {
        int64_t k = RFX_AS_I64(dkeys)[i];
        if (k == s_from || k == s_where || k == s_by || k == s_take) continue;
        obj_p e = RFX_AS_LIST(dvals)[i];
        const int n = M->nagg;
        if (n >= RFX_MAX_AGGS || e->type != RFX_TYPE_LIST || e->len != 2) { *why = "mapping shape"; return SEL_OUT; }
        int f = fn_id(RFX_AS_LIST(e)[0]);
        obj_p a = RFX_AS_LIST(e)[1];
        static const int KIND[] = {RFX_AGG_SUM, RFX_AGG_AVG, RFX_AGG_MIN, RFX_AGG_MAX, RFX_AGG_COUNT, RFX_AGG_FIRST};
        if (f < F_SUM || f > F_FIRST) { *why = "mapping is not (aggr ...)"; return SEL_OUT; }
        memset(&M->aggs[n], 0, sizeof(M->aggs[n]));
        M->aggs[n].kind = KIND[f - F_SUM];
        if (a->type == RFX_TYPE_LIST && a->len == 3) {
            /* (aggr expr), expr = (op x y) over columns, atoms and nested expressions: folded on the device.  (count expr)
             * answers the number of groups in the reference and (first expr) under by: is a `length` error there: the host's */
            if (f == F_COUNT || f == F_FIRST) { *why = "count / first of an expression"; return SEL_OUT; }
            int nn = 0, ncols = 0;
            int top = build_xnodes(tab, a, M->xnodes[n], &nn, &ncols, why);
            if (top == -2) return SEL_DONE;
            if (top < 0) return SEL_OUT;
            if (ncols == 0) { *why = "expression without a column"; return SEL_OUT; }
            M->aggs[n].nxnodes = nn;
            M->aggs[n].xnodes = M->xnodes[n];
            M->aggs[n].col_type = RFX_I64;
            M->outtype[n] = (f == F_AVG || rfx_agg_input_type(&M->aggs[n]) == RFX_F64) ? RFX_TYPE_F64 : RFX_TYPE_I64;
            M->names[M->nagg++] = k;
            continue;
        }
        if (a->type != -RFX_TYPE_SYMBOL) { *why = "mapping is not (aggr column)"; return SEL_OUT; }
        obj_p c = table_col(tab, a->i64);
        /* a 4-byte integer column (I32 / DATE / TIME): min / max / first / count / sum fold its widened device copy and the result cells
         * are narrowed back (sel_narrow_i32); avg and the sum of dates are the host's */
        const int narrow = c && IS_I32_FAMILY(c->type) && !(g_npx && proxy_of(c)) && !(grouped && c->type == RFX_TYPE_I32) && /* (any grouped aggregate over an I32 column is a `type` error in the reference: its to say) */
                           ((f == F_MIN || f == F_MAX || f == F_FIRST || f == F_COUNT) ||
                            /* sums of I32 / TIME columns wrap in 32 bits there (FOLD_ADDI32 / ADDI32, core/math.c:1864-1871, core/aggr.c:1095-1100):
                             * the low 32 bits of the 64-bit sum of the widened column are that sum */
                            (f == F_SUM && !grouped && (c->type == RFX_TYPE_I32 || c->type == RFX_TYPE_TIME))); /* (grouped: a `type` error there) */
        if (!c || (!narrow && (!col_ctype(c) || c->type == RFX_TYPE_SYMBOL))) { *why = "aggregate column type"; return SEL_OUT; }
        const void *d;
        if (resident(c, 0, &d) != RFX_OK) return SEL_DONE;
        M->aggs[n].d_col = d;
        M->aggs[n].col_type = narrow ? RFX_I64 : col_ctype(c);
        M->outtype[n] = (f == F_AVG) ? RFX_TYPE_F64 : (f == F_COUNT) ? RFX_TYPE_I64 : c->type;
        M->names[M->nagg++] = k;
    }
    return SEL_GO;
}

/* the key column(s) of a group-by result, read back in group order from what the planner emitted: one key as its cells (the virtual Date
 * column of a parted table narrowed to 4-byte days, ENUM indices decoded through the enum's domain -- aggr_first, core/aggr.c:515-546);
 * several keys as the planner's key columns (decoded from the composite key = key_i[first row], core/query.c:110-135, or gathered at the
 * groups' first rows on the row-hash path).  *ok carries the device-call status on; SEL_OUT: an enum whose domain does not resolve. */
typedef struct {
    int nkeys;
    int8_t key_out_type;
    obj_p kenum;
    obj_p *kcs;
} sel_keys_t;
/* (two steps around ONE read-back of every result column -- rfx_exec_groups_fetch_all, each slice over its own device's link:
 * sel_key_columns_plan makes the vectors and names (device column, host destination) pairs, sel_key_columns_finish narrows / decodes) */
typedef struct {
    int n;
    const void *src[RFX_MAX_KEYS + RFX_MAX_AGGS];
    void *dst[RFX_MAX_KEYS + RFX_MAX_AGGS];
    void *tmp[RFX_MAX_KEYS + RFX_MAX_AGGS]; /* 8-byte staging of a column whose vector is 4 bytes wide (freed by the caller) */
    int ntmp;
} sel_fetch_t;
static int sel_fetch_add(sel_fetch_t *F, const void *src, void *dst) {
    if (F->n >= (int)(sizeof(F->src) / sizeof(F->src[0]))) return 0;
    F->src[F->n] = src;
    F->dst[F->n++] = dst;
    return 1;
}
static void *sel_fetch_tmp(sel_fetch_t *F, int64_t groups) {
    void *t = malloc((size_t)(groups ? groups : 1) * 8);
    if (t) F->tmp[F->ntmp++] = t;
    return t;
}
static int sel_key_columns_plan(const sel_keys_t *K, const rfx_groups_t *R, obj_p *okcols, sel_fetch_t *F, int64_t **k8) {
    const int64_t groups = R->groups;
    int ok = 1;
    *k8 = NULL;
    if (K->nkeys == 1 && K->key_out_type == RFX_TYPE_DATE) { /* the virtual Date column: 4-byte days */
        okcols[0] = H.vector(RFX_TYPE_DATE, groups);
        *k8 = (int64_t *)sel_fetch_tmp(F, groups);
        ok = *k8 && sel_fetch_add(F, R->d_keys, *k8);
    } else if (K->nkeys == 1) {
        okcols[0] = H.vector(K->key_out_type, groups);
        ok = sel_fetch_add(F, R->d_keys, RFX_AS_RAW(okcols[0]));
    } else {
        for (int i = 0; i < K->nkeys && ok; i++) {
            okcols[i] = H.vector(K->kcs[i]->type, groups);
            ok = sel_fetch_add(F, R->d_keycols[i], RFX_AS_RAW(okcols[i]));
        }
    }
    return ok;
}
static int sel_key_columns_finish(const sel_keys_t *K, const rfx_groups_t *R, obj_p *okcols, const int64_t *k8) {
    const int64_t groups = R->groups;
    if (K->nkeys == 1 && K->key_out_type == RFX_TYPE_DATE) {
        for (int64_t g = 0; g < groups; g++) ((int32_t *)RFX_AS_RAW(okcols[0]))[g] = (int32_t)k8[g];
    } else if (K->nkeys == 1 && K->kenum) { /* indices -> symbols of the enum's domain (the global its key names) */
        obj_p dom = enum_domain(K->kenum);
        int good = dom != NULL;
        int64_t *kk = RFX_AS_I64(okcols[0]);
        for (int64_t g = 0; g < groups && good; g++) {
            if (kk[g] < 0 || kk[g] >= dom->len) good = 0;
            else kk[g] = RFX_AS_I64(dom)[kk[g]];
        }
        if (dom) H.drop(dom);
        if (!good) {
            H.drop(okcols[0]);
            okcols[0] = NULL;
            return SEL_OUT;
        }
    }
    return SEL_GO;
}

/* by: a column symbol, or a dict {name: column | (xbar column positive-width) ...} (get_gkeys / get_gvals, core/query.c:165-240): the key
 * columns as the table holds them, the names they take in the result, and the bucket width of the bucketed ones */
static int sel_by_shape(obj_p tab, obj_p by, obj_p *kcs, int64_t *knames, int64_t *kxbar, int *nkeys, const char **why) {
    *nkeys = 0;
    if (by->type == -RFX_TYPE_SYMBOL) {
        knames[0] = by->i64;
        kxbar[0] = 0;
        kcs[(*nkeys)++] = table_col(tab, by->i64);
        return SEL_GO;
    }
    if (!(by->type == RFX_TYPE_DICT && RFX_AS_LIST(by)[0]->type == RFX_TYPE_SYMBOL)) { *why = "by: is neither a column nor a dict of columns"; return SEL_OUT; }
    obj_p bk = RFX_AS_LIST(by)[0], bv = RFX_AS_LIST(by)[1];
    if (bk->len < 1 || bk->len > RFX_MAX_KEYS || bv->len != bk->len) { *why = "by: dict shape"; return SEL_OUT; }
    for (int64_t i = 0; i < bk->len; i++) {
        int64_t sym;
        obj_p bx = (bv->type == RFX_TYPE_LIST) ? RFX_AS_LIST(bv)[i] : NULL;
        kxbar[*nkeys] = 0;
        if (bv->type == RFX_TYPE_SYMBOL) sym = RFX_AS_I64(bv)[i];
        else if (bx && bx->type == -RFX_TYPE_SYMBOL) sym = bx->i64;
        else if (bx && bx->type == RFX_TYPE_LIST && bx->len == 3 && fn_id(RFX_AS_LIST(bx)[0]) == F_XBAR && RFX_AS_LIST(bx)[1]->type == -RFX_TYPE_SYMBOL &&
                 RFX_AS_LIST(bx)[2]->type == -RFX_TYPE_I64 && RFX_AS_LIST(bx)[2]->i64 > 0) {
            sym = RFX_AS_LIST(bx)[1]->i64; /* (xbar column width): bucketed key, evaluated on the device by the caller */
            kxbar[*nkeys] = RFX_AS_LIST(bx)[2]->i64;
        } else { *why = "by: key is an expression other than (xbar column positive-width)"; return SEL_OUT; }
        knames[*nkeys] = RFX_AS_I64(bk)[i];
        kcs[(*nkeys)++] = table_col(tab, sym);
    }
    return SEL_GO;
}

/* select without aggregates: filter_collect of every column (core/filter.c:51-165) -- where -> ids (every shard its own, rfx_exec_where) ->
 * every column gathered at them where its rows live, straight into the result vectors */
static int sel_projection(obj_p tab, const rfx_query_t *Q, int parted, obj_p *res, const char **why) {
    obj_p tcols = RFX_AS_LIST(tab)[1];
    if (parted) { *why = "parted table: projection"; return SEL_OUT; } /* the reference keeps such a result lazy (filter maps over the partitions) */
    if (!Q->npred && !Q->d_mask) { *res = H.clone(tab); g_last_gpu = 1; return SEL_DONE; }
    for (int64_t i = 0; i < tcols->len; i++)
        if (!col_ctype(RFX_AS_LIST(tcols)[i])) { *why = "projection of a non-8-byte column"; return SEL_OUT; }
    rfx_ids_t ids;
    if (rfx_exec_where(g_x, Q, &ids) != RFX_OK) { *res = fail(rfx_exec_last_error(g_x)); return SEL_DONE; }
    const int64_t nsel = ids.total;
    obj_p rv = H.vector(RFX_TYPE_LIST, tcols->len);
    int ok = 1;
    for (int64_t i = 0; i < tcols->len && ok; i++) {
        obj_p c = RFX_AS_LIST(tcols)[i];
        obj_p o = H.vector(c->type, nsel);
        RFX_AS_LIST(rv)[i] = o;
        const void *dc;
        if (nsel == 0) continue;
        ok = resident(c, 0, &dc) == RFX_OK;
        int64_t at = 0;
        for (int sh = 0; sh < ids.nshards && ok; sh++) {
            if (!ids.count[sh]) continue;
            int64_t r0;
            rfx_exec_split(Q->nrows, ids.nshards, sh, &r0, NULL);
            const void *dcs = dc; /* this shard's slice, addressed by the GLOBAL ids it emitted */
            for (int k = 0; k < g_nqcols && sh > 0; k++)
                if (g_qcols[k].d[0] == dc) dcs = g_qcols[k].d[sh];
            void *dg = NULL;
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[sh]);
            ok = rfx_hip_malloc(g_ctxs[sh], &dg, (size_t)ids.count[sh] * 8) == RFX_OK &&
                 rfx_hip_gather(g_ctxs[sh], (const char *)dcs - (size_t)r0 * 8, ids.d_ids[sh], ids.count[sh], dg) == RFX_OK &&
                 rfx_hip_d2h(g_ctxs[sh], (char *)RFX_AS_RAW(o) + (size_t)at * 8, dg, (size_t)ids.count[sh] * 8) == RFX_OK;
            if (dg) rfx_hip_free(g_ctxs[sh], dg);
            at += ids.count[sh];
        }
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    }
    rfx_exec_ids_free(g_x, &ids);
    if (!ok) { H.drop(rv); *res = fail_hip("projection"); return SEL_DONE; }
    *res = H.table(H.clone(RFX_AS_LIST(tab)[0]), rv);
    g_last_gpu = 1;
    return SEL_DONE;
}

/* rfx_select = PLAN (the dict's clauses as descriptors over resident columns: sel_mappings, plan_where, sel_by_shape -- and what the
 * reference answers differently is handed back before anything runs) -> RUN (the planner: rfx_exec_group_by / rfx_exec_filter_aggr /
 * rfx_exec_where over the operator layer's shards) -> BUILD (the result table from the planner's device columns). */
static obj_p sel_build_groups(const rfx_groups_t *R, const sel_maps_t *M, const sel_keys_t *K, const int64_t *knames, const char **why) {
    const int nagg = M->nagg, nkeys = K->nkeys;
    obj_p ocols[RFX_MAX_AGGS] = {0}, okcols[RFX_MAX_KEYS] = {0};
    int ok = 1, enum_out = 0;
    if (R->groups > 0) {
        /* every result vector first, then ONE read-back of all of them (every slice of a sliced result by the shard that holds it, over that
         * device's own link: the table construction of core/query.c:559-605 with N writers), then the 4-byte narrowing / enum decoding */
        sel_fetch_t F;
        int64_t *k8 = NULL, *c8[RFX_MAX_AGGS] = {0};
        memset(&F, 0, sizeof(F));
        ok = sel_key_columns_plan(K, R, okcols, &F, &k8);
        for (int a = 0; a < nagg && ok; a++) {
            ocols[a] = H.vector((int8_t)M->outtype[a], R->groups);
            if (IS_I32_FAMILY(M->outtype[a])) {
                c8[a] = (int64_t *)sel_fetch_tmp(&F, R->groups);
                ok = c8[a] && sel_fetch_add(&F, R->d_results[a], c8[a]);
            } else ok = sel_fetch_add(&F, R->d_results[a], RFX_AS_RAW(ocols[a]));
        }
        if (ok) ok = rfx_exec_groups_fetch_all(g_x, R, F.n, F.src, F.dst) == RFX_OK;
        if (ok) {
            enum_out = sel_key_columns_finish(K, R, okcols, k8) == SEL_OUT;
            for (int a = 0; a < nagg && !enum_out; a++)
                if (c8[a]) sel_narrow_i32(ocols[a], c8[a], R->groups, M->aggs[a].kind);
        }
        for (int i = 0; i < F.ntmp; i++) free(F.tmp[i]);
    }
    if (!ok || enum_out) {
        for (int i = 0; i < nkeys; i++) if (okcols[i]) H.drop(okcols[i]);
        for (int a = 0; a < nagg; a++) if (ocols[a]) H.drop(ocols[a]);
        if (enum_out) {
            *why = "by: enum column whose domain cannot be resolved";
            return NULL;
        }
        return fail_hip("group-by result");
    }
    obj_p rk = H.vector(RFX_TYPE_SYMBOL, nagg + nkeys), rv = H.vector(RFX_TYPE_LIST, nagg + nkeys);
    for (int i = 0; i < nkeys; i++) {
        RFX_AS_I64(rk)[i] = knames[i];
        RFX_AS_LIST(rv)[i] = okcols[i] ? okcols[i] : H.vector(nkeys == 1 ? K->key_out_type : K->kcs[i]->type, 0);
    }
    for (int a = 0; a < nagg; a++) {
        RFX_AS_I64(rk)[a + nkeys] = M->names[a];
        RFX_AS_LIST(rv)[a + nkeys] = ocols[a] ? ocols[a] : H.vector((int8_t)M->outtype[a], 0);
    }
    return H.table(rk, rv);
}
static obj_p sel_build_scalar(const rfx_value_t *vals, const sel_maps_t *M) {
    const int nagg = M->nagg;
    obj_p rk = H.vector(RFX_TYPE_SYMBOL, nagg), rv = H.vector(RFX_TYPE_LIST, nagg);
    for (int a = 0; a < nagg; a++) {
        RFX_AS_I64(rk)[a] = M->names[a];
        if (IS_I32_FAMILY(M->outtype[a])) {
            RFX_AS_LIST(rv)[a] = H.vector((int8_t)M->outtype[a], 1);
            sel_narrow_i32(RFX_AS_LIST(rv)[a], &vals[a].i, 1, M->aggs[a].kind);
        } else {
            RFX_AS_LIST(rv)[a] = one_row(&vals[a]);
            if (M->outtype[a] == RFX_TYPE_TIMESTAMP && vals[a].type != RFX_F64) RFX_AS_LIST(rv)[a]->type = RFX_TYPE_TIMESTAMP; /* min / max / first of a TIMESTAMP column */
        }
    }
    return H.table(rk, rv);
}

static obj_p select_impl(obj_p dict) {
    rfx_host_bind();
    g_ntm = 0;
    tm_mark();
    if (!dict || dict->type != RFX_TYPE_DICT || RFX_AS_LIST(dict)[0]->type != RFX_TYPE_SYMBOL) return fail("select: expected a dict");
    obj_p from = dict_get(dict, "from");
    if (!from) return fail("'select' expects 'from' param"); /* core/query.c:281 */
    /* take: is applied to the finished result table (ray_take(res, take), core/query.c:294-303,596-599): by the host's own ray_take */
    obj_p take = dict_get(dict, "take");
    if (take && !(H.bound == 1 && H.f[F_TAKE])) return delegate_select(dict, "take: without the host's ray_take");
    obj_p host_tab = HOST_CALL(H.eval(from)); /* (the host may fan this out to pool workers that call rfx_* built-ins: not under our lock; no device state is held yet) */
    if (!host_tab || host_tab->type == RFX_TYPE_ERR) return host_tab;
    obj_p tab = host_tab; /* the table the plan reads: host_tab itself, or the view of a parted table */
    int parted = 0;
    obj_p res = NULL;
    const char *why = NULL;
    void *tmp[2 * RFX_MAX_KEYS + 6]; /* device scratch of this query on shard 0 (a mask): freed at `done` */
    int ntmp = 0;
    obj_p where = dict_get(dict, "where"), by = dict_get(dict, "by");
    obj_p dkeys = RFX_AS_LIST(dict)[0], dvals = RFX_AS_LIST(dict)[1];
    if (tab->type != RFX_TYPE_TABLE) { why = "from: is not a table"; goto out; }
    if (ensure_ctx() != RFX_OK) { res = fail_hip("no usable MI355X"); goto done; }
    if (is_parted_table(host_tab)) {
        if (g_nshards > 1) { why = "parted table: the sharded operator layer takes in-memory tables"; goto out; }
        tab = parted_view(host_tab);
        if (!tab) { parted_view_release(); tab = host_tab; why = "parted table: view"; goto out; }
        parted = 1;
    }
    {
        /* ---------------------------------------------------------------- PLAN */
        obj_p tcols = RFX_AS_LIST(tab)[1];
        const int64_t nrows = tcols->len ? RFX_AS_LIST(tcols)[0]->len : 0;
        wplan_t wp;
        int flat = 1;
        g_where_virtual = g_where_data = 0;
        int rc = plan_where(tab, where, &wp);
        if (rc == -2) { res = fail_hip("column upload"); goto done; }
        if (rc) { /* more comparisons / levels than the fused form carries: its selection comes as a mask */
            flat = 0;
            wp.npred = 0;
            wp.logic = RFX_AND;
        }
        if (parted) {
            /* What the reference answers correctly over a parted table, and so what is answered here: aggregates, over everything or
             * grouped by the virtual column, filtered by the virtual column (partition pruning) or -- ungrouped -- by data columns.
             * A filter mixing both kinds, and a data-column filter under by:, come out wrong there (DESIGN.md "reference defects"):
             * left to the host so that this entry point never answers differently. */
            if (!flat) { why = "parted table: where: is not a flat and / or of comparisons"; goto out; }
            for (int i = 0; i < wp.npred; i++)
                if (wp.preds[i].more) { why = "parted table: where: is not a flat and / or of comparisons"; goto out; }
            if (g_where_virtual && g_where_data) { why = "parted table: where: mixes the virtual column with data columns"; goto out; }
            if (by && g_where_data) { why = "parted table: by: under a data-column filter"; goto out; }
        }
        sel_maps_t M;
        {
            const int mrc = sel_mappings(tab, dkeys, dvals, by != NULL, &M, &why);
            if (mrc == SEL_DONE) { res = fail_hip("column upload"); goto done; }
            if (mrc == SEL_OUT) goto out;
        }
        /* by: a column symbol, or a dict {name: column ...} (get_gkeys / get_gvals, core/query.c:165-240) */
        obj_p kcs[RFX_MAX_KEYS] = {0};
        const void *dks[RFX_MAX_KEYS] = {0};
        int64_t knames[RFX_MAX_KEYS], kxbar[RFX_MAX_KEYS] = {0};
        int nkeys = 0;
        int8_t key_out_type = RFX_TYPE_I64; /* one key: type of the result's key column */
        obj_p kenum = NULL;                 /* one key, an ENUM column */
        if (by) {
            if (sel_by_shape(tab, by, kcs, knames, kxbar, &nkeys, &why) == SEL_OUT) goto out;
            for (int i = 0; i < nkeys; i++) {
                if (parted) { /* only the virtual column groups a parted table in the reference (INDEX_TYPE_PARTEDCOMMON, core/index.c:2199-2222) */
                    const proxy_t *px = kcs[i] ? proxy_of(kcs[i]) : NULL;
                    if (nkeys != 1 || !px || px->kind != 2 || kcs[i]->type != RFX_TYPE_I64) { why = "parted table: by: is not the virtual column"; goto out; }
                    key_out_type = px->vtype;
                } else if (kcs[i] && kcs[i]->type == RFX_TYPE_ENUM && nkeys == 1 && !kxbar[i]) {
                    /* an enumerated symbol column groups on its indices (index_group_i64(ENUM_VAL(val)), core/index.c:2190-2191); the
                     * result's key column is decoded through the enum's domain (aggr_first, core/aggr.c:515-546) */
                    kenum = kcs[i];
                    if (resident(enum_indices(kcs[i]), 0, &dks[i]) != RFX_OK) { res = fail_hip("column upload"); goto done; }
                    key_out_type = RFX_TYPE_SYMBOL;
                    continue;
                }
                /* index_group's 8-byte integer arms: I64 / SYMBOL / TIMESTAMP group on the raw i64 (core/index.c:2183-2186); an F64 key
                 * column groups on its BIT PATTERN through the open-addressing path (index_group_f64 = index_group_i64_unscoped,
                 * core/index.c:2108,1959-1977): the same device column read as i64 -- a range of bit patterns is never dense, so the
                 * hashed tables take it here too; -0.0 has the bits of NULL_I64, the reference's empty-slot marker: handed back like
                 * any null key.  One key column only (several keys with an f64 among them are the host's). */
                const int f64key = kcs[i] && kcs[i]->type == RFX_TYPE_F64 && nkeys == 1 && !kxbar[i] && !parted;
                if (!kcs[i] || !(kcs[i]->type == RFX_TYPE_I64 || kcs[i]->type == RFX_TYPE_SYMBOL || kcs[i]->type == RFX_TYPE_TIMESTAMP || f64key)) {
                    why = "by: key is not an 8-byte integer column";
                    goto out;
                }
                if (kxbar[i] > 0 && kcs[i]->type == RFX_TYPE_SYMBOL) { why = "xbar over a symbol column"; goto out; }
                if (nkeys == 1 && !parted) key_out_type = kcs[i]->type;
                if (resident(kcs[i], 0, &dks[i]) != RFX_OK) { res = fail_hip("column upload"); goto done; }
            }
            /* where: + several keys: the reference's own result is defective (its composite index drops the filter, so key
             * columns and aggregates are taken from the wrong rows -- DESIGN.md "reference defects"); leave that to the host
             * so that this entry point never answers differently from ray_select. */
            if (nkeys > 1 && where) { why = "where: with several by: columns"; goto out; }
        }
        rfx_query_t Q;
        memset(&Q, 0, sizeof(Q));
        Q.preds = wp.preds;
        Q.npred = wp.npred;
        Q.logic = wp.logic;
        Q.aggs = M.aggs;
        Q.nagg = M.nagg;
        Q.nkeys = nkeys;
        Q.d_keys = dks;
        Q.kxbar = kxbar;
        Q.nrows = nrows;
        if (!flat) { /* the tree as ONE B8 mask on the device (core/cmp.c -> K2, core/logic.c in place), handed to the planner beside the query */
            if (g_nshards > 1) { why = "sharded table: where: tree beyond the fused form"; goto out; }
            int8_t *m = NULL;
            const int mrc = mask_of_expr(tab, where, nrows, &m);
            if (mrc == -1) { why = "where: shape"; goto out; }
            if (mrc) { res = fail_hip("where"); goto done; }
            tmp[ntmp++] = m;
            Q.d_mask = m;
        }
        Q.cols = g_nshards > 1 ? g_qcols : NULL;
        Q.ncols = g_nqcols;
        tm_mark();
        /* ---------------------------------------------------------------- RUN + BUILD */
        if (!by && M.nagg == 0) { /* projection */
            if (sel_projection(tab, &Q, parted, &res, &why) == SEL_OUT) goto out;
            goto done;
        }
        if (by) {
            /* small inputs over a plain resident key column: its whole-column scope, remembered with the device copy (or taken now, without
             * the filter: the same pass) -- a superset of any selection's, which is all the tables' sizing needs; the planner takes it when it
             * is LDS-sized and saves the scope round trip */
            int64_t kscope[2];
            Q.flags = RFX_Q_REFUSE_NULL_KEY | /* the reference opens one group per null-key row (core/index.c:1808-1816): its own select answers those */
                      RFX_Q_SLICED;           /* the result is read through rfx_exec_groups_fetch_all only: every device may keep and read back its own slice */
            resident_t *ke = (g_nshards == 1 && nkeys == 1 && !parted && flat && nrows > 0 && nrows < ((int64_t)1 << 24) && !kxbar[0]) ? resident_entry(dks[0]) : NULL;
            if (ke) {
                if (!ke->scope_ok) {
                    int64_t c0 = 0;
                    if (rfx_hip_scope_i64(g_ctx, (const int64_t *)dks[0], NULL, 0, RFX_AND, nrows, &ke->smin, &ke->smax, &c0) != RFX_OK) { res = fail_hip("scope"); goto done; }
                    ke->scope_ok = 1;
                }
                kscope[0] = ke->smin;
                kscope[1] = ke->smax;
                Q.key_scope = kscope;
            }
            rfx_groups_t R;
            const int grc = rfx_exec_group_by(g_x, &Q, &R);
            tm_mark();
            if (grc == RFX_EXEC_NULL_KEY) { why = "null group key"; goto out; }
            if (grc == RFX_ESTATE && strstr(rfx_exec_last_error(g_x), "collision")) { why = "row-hash collision between two key tuples"; goto out; }
            if (grc == RFX_ELIMIT && g_nshards > 1) { why = "sharded table: shape the planner runs on one shard"; goto out; }
            if (grc != RFX_OK) { res = fail(rfx_exec_last_error(g_x)); goto done; }
            const sel_keys_t K = {nkeys, key_out_type, kenum, kcs};
            res = sel_build_groups(&R, &M, &K, knames, &why);
            rfx_exec_groups_free(g_x, &R);
            tm_mark();
            if (!res) goto out;
            g_last_gpu = res->type == RFX_TYPE_TABLE;
            goto done;
        }
        rfx_value_t vals[RFX_MAX_AGGS];
        int64_t selected = 0;
        if (rfx_exec_filter_aggr(g_x, &Q, vals, &selected) != RFX_OK) { res = fail(rfx_exec_last_error(g_x)); goto done; }
        res = sel_build_scalar(vals, &M);
        g_last_gpu = 1;
        goto done;
    }
out: /* hand the query to the host -- with this call's device scratch released first (the host may fan out to its pool: HOST_CALL) */
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    ntmp = 0;
    qtmp_release();
    if (parted) parted_view_release();
    parted = 0;
    res = delegate_select(dict, why ? why : "unsupported");
done:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    qtmp_release();
    if (parted) parted_view_release();
    H.drop(host_tab);
    tm_mark();
    tm_print();
    if (take && g_last_gpu && res && res->type == RFX_TYPE_TABLE) { /* (a delegated query had its take: applied by ray_select) */
        obj_p tv = HOST_CALL(H.eval(take));
        if (tv && tv->type != RFX_TYPE_ERR) {
            obj_p cut = HOST_CALL(((rfx_binary_f)H.f[F_TAKE])(res, tv));
            H.drop(res);
            res = cut;
        } else {
            H.drop(res);
            res = tv;
        }
        if (tv && res != tv) H.drop(tv);
    }
    return res;
}
rfx_obj_p rfx_select(rfx_obj_p dict) {
    op_begin();
    g_last_gpu = 0;
    obj_p r = select_impl(dict);
    g_stat[g_last_gpu ? ST_SELECT_GPU : ST_SELECT_DELEGATED]++;
    op_end();
    return r;
}

/* ------------------------------------------------------------------------------------------------ update
 * (update {col: mapping ... from: t [where: p] [by: k]}) -- ray_update, core/update.c:936-1106.  The reference turns `where:` into
 * row ids (ray_where), evaluates every mapping over the filtered / grouped table and writes: under a filter, value i goes to row
 * ids[i]; under `by:`, each group's aggregate goes to all of that group's selected rows; a name the table does not have becomes a
 * new column that is null elsewhere (__update_table).  Covered here: `from:` a table VALUE (the quoted-symbol form updates the
 * host's global in place: the host's own job), flat or nested `where:`, mappings that are an i64 / f64 atom, a column, an
 * element-wise expression (+ - * div, nested) -- and, with `by:` one 8-byte integer key column, (aggr column) for sum / avg / min /
 * max / count / first under a flat `where:`.  Value and column types must agree (the reference also casts f64 into i64 columns:
 * delegated).  Anything else is the host's ray_update. */
static obj_p delegate_update(obj_p dict, const char *why) {
    if (H.bound == 1 && H.f[F_UPDATE]) return HOST_CALL(((rfx_unary_f)H.f[F_UPDATE])(dict));
    char b[300];
    snprintf(b, sizeof(b), "rfx_update: shape not covered by the MI355X path (%s) and no host ray_update to delegate to", why);
    return fail(b);
}
/* ---- update, step by step: upd_where -> upd_by -> per mapping { upd_value -> upd_column } -> upd_result.  Every step answers UPD_GO, UPD_BACK
 * (the shape is the host's: u->why says which) or UPD_STOP (u->res is the error object); update_impl owns the cleanup. ---- */
enum { UPD_GO = 0, UPD_BACK = 1, UPD_STOP = 2 };
typedef struct {
    obj_p tab;
    int64_t nrows;
    void *tmp[3 * RFX_MAX_AGGS + 8]; /* device blocks of this call */
    int ntmp;
    const char *why;
    obj_p res;
    wplan_t wp;      /* where: as comparisons (when flat) */
    int64_t *d_ids;  /* ... as row ids (NULL: every row) */
    int64_t m;       /* rows written */
    const void *dk;  /* by: the key column on the device, its scope through the predicates */
    int64_t kmin, kmax, seen;
} upd_t;
typedef struct { /* what one mapping writes */
    int vtype;             /* RFX_I64 | RFX_F64: element type of the values */
    const void *dvals_col; /* a full-length value column (no by:), or */
    uint64_t atom_bits;    /* ... one value for every row, or */
    rfx_agg_t agg;         /* ... (by:) an aggregate per group */
} upd_val_t;
static int upd_back(upd_t *u, const char *why) { u->why = why; return UPD_BACK; }
static int upd_stop(upd_t *u, obj_p err) { u->res = err; return UPD_STOP; }

/* where: -> row ids (ray_where) */
static int upd_where(upd_t *u, obj_p where, obj_p by) {
    int flat = 1;
    const int prc = plan_where(u->tab, where, &u->wp);
    if (prc == -2) return upd_stop(u, fail_hip("column upload"));
    if (prc < 0) flat = 0;
    if (by && !flat) return upd_back(u, "by: with a nested where: tree");
    const int wrc = where_ids(u->tab, where, &u->wp, flat, u->nrows, &u->d_ids, &u->m);
    if (wrc == -2) return upd_stop(u, fail_hip("where"));
    if (wrc < 0) return upd_back(u, "where: shape");
    if (u->d_ids) u->tmp[u->ntmp++] = u->d_ids;
    return UPD_GO;
}
/* by: one 8-byte integer key column with a dense scope */
static int upd_by(upd_t *u, obj_p by) {
    if (by->type != -RFX_TYPE_SYMBOL) return upd_back(u, "by: is not one column");
    obj_p kc = table_col(u->tab, by->i64);
    if (!kc || !(kc->type == RFX_TYPE_I64 || kc->type == RFX_TYPE_SYMBOL || kc->type == RFX_TYPE_TIMESTAMP)) return upd_back(u, "by: key is not an 8-byte integer column");
    if (resident(kc, 0, &u->dk) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
    if (rfx_hip_scope_i64(g_ctx, (const int64_t *)u->dk, u->wp.preds, u->wp.npred, u->wp.logic, u->nrows, &u->kmin, &u->kmax, &u->seen) != RFX_OK) return upd_stop(u, fail_hip("scope"));
    const uint64_t range = u->seen > 0 ? (uint64_t)u->kmax - (uint64_t)u->kmin + 1 : 0;
    if (u->seen > 0 && !(range != 0 && range <= (uint64_t)u->seen && u->kmin != RFX_NULL_I64 && range <= (1ull << 31))) return upd_back(u, "by: sparse or null keys");
    return UPD_GO;
}
/* the values of one mapping: (aggr column) under by:, else an atom, a column or an element-wise expression (evaluated into a device column) */
static int upd_value(upd_t *u, obj_p e, obj_p by, upd_val_t *v) {
    obj_p tab = u->tab;
    memset(v, 0, sizeof(*v));
    if (by) {
        if (e->type != RFX_TYPE_LIST || e->len != 2) return upd_back(u, "by: mapping is not (aggr column)");
        const int f = fn_id(RFX_AS_LIST(e)[0]);
        static const int KIND[] = {RFX_AGG_SUM, RFX_AGG_AVG, RFX_AGG_MIN, RFX_AGG_MAX, RFX_AGG_COUNT, RFX_AGG_FIRST};
        obj_p a = RFX_AS_LIST(e)[1];
        if (f < F_SUM || f > F_FIRST || a->type != -RFX_TYPE_SYMBOL) return upd_back(u, "by: mapping is not (aggr column)");
        obj_p c = table_col(tab, a->i64);
        if (!c || !col_ctype(c) || c->type == RFX_TYPE_SYMBOL) return upd_back(u, "aggregate column type");
        const void *d;
        if (resident(c, 0, &d) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
        v->agg.d_col = d;
        v->agg.col_type = col_ctype(c);
        v->agg.kind = KIND[f - F_SUM];
        v->vtype = (f == F_AVG) ? RFX_F64 : (f == F_COUNT) ? RFX_I64 : col_ctype(c);
    } else if (e->type == -RFX_TYPE_I64) { v->vtype = RFX_I64; v->atom_bits = (uint64_t)e->i64; }
    else if (e->type == -RFX_TYPE_F64) { v->vtype = RFX_F64; memcpy(&v->atom_bits, &e->f64, 8); }
    else if (e->type == -RFX_TYPE_SYMBOL) {
        obj_p c = table_col(tab, e->i64);
        if (!c || !(c->type == RFX_TYPE_I64 || c->type == RFX_TYPE_F64)) return upd_back(u, "mapping column type");
        if (resident(c, 0, &v->dvals_col) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
        v->vtype = col_ctype(c);
    } else if (e->type == RFX_TYPE_LIST && e->len == 3) {
        rfx_xnode_t nodes[RFX_MAX_XNODES];
        int nn = 0, ncols = 0;
        const char *why = NULL;
        const int top = build_xnodes(tab, e, nodes, &nn, &ncols, &why);
        if (top == -2) return upd_stop(u, fail_hip("column upload"));
        if (top < 0) return upd_back(u, why);
        if (ncols == 0) return upd_back(u, "expression without a column");
        rfx_agg_t xa;
        memset(&xa, 0, sizeof(xa));
        xa.kind = RFX_AGG_SUM;
        xa.col_type = RFX_I64;
        xa.nxnodes = nn;
        xa.xnodes = nodes;
        void *dx = NULL;
        if (rfx_hip_malloc(g_ctx, &dx, (size_t)u->nrows * 8) != RFX_OK) return upd_stop(u, fail_hip("expression column"));
        u->tmp[u->ntmp++] = dx;
        int32_t ot = RFX_I64;
        if (rfx_hip_eval_expr(g_ctx, &xa, u->nrows, dx, &ot) != RFX_OK) return upd_stop(u, fail_hip("eval_expr"));
        v->dvals_col = dx;
        v->vtype = ot;
    } else return upd_back(u, "mapping is neither an atom, a column nor an element-wise expression");
    return UPD_GO;
}
/* a device copy of the column `tc` (or nulls for a new one), the writes of one mapping into it, and the host vector it becomes */
static int upd_column(upd_t *u, obj_p tc, obj_p by, const upd_val_t *v, obj_p *newcol) {
    const int64_t nrows = u->nrows;
    /* __suitable_types (update.c:81-107): same type, or an i64 column taking f64 values by conversion -- the latter is the host's */
    int8_t out_type = v->vtype == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64;
    if (tc) {
        if (!(tc->type == RFX_TYPE_I64 || tc->type == RFX_TYPE_F64)) return upd_back(u, "updated column is not i64 / f64");
        if (col_ctype(tc) != v->vtype) return upd_back(u, "value type differs from the column's (the reference converts; delegated)");
        out_type = tc->type;
    }
    void *dcol = NULL;
    if (rfx_hip_malloc(g_ctx, &dcol, (size_t)nrows * 8) != RFX_OK) return upd_stop(u, fail_hip("column copy"));
    u->tmp[u->ntmp++] = dcol;
    if (tc) {
        const void *dold;
        if (resident(tc, 0, &dold) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
        if (rfx_hip_update_set(g_ctx, dcol, NULL, nrows, dold, 0) != RFX_OK) return upd_stop(u, fail_hip("column copy")); /* copy: ids NULL, vals = old */
    } else if (rfx_hip_update_set(g_ctx, dcol, NULL, nrows, NULL, v->vtype == RFX_F64 ? 0x7FF8000000000000ull : 0x8000000000000000ull) != RFX_OK)
        return upd_stop(u, fail_hip("column fill"));
    if (u->m > 0) {
        if (!by) {
            if (rfx_hip_update_set(g_ctx, dcol, u->d_ids, u->m, v->dvals_col, v->atom_bits) != RFX_OK) return upd_stop(u, fail_hip("update_set"));
        } else if (u->seen > 0) {
            const int64_t range = (int64_t)((uint64_t)u->kmax - (uint64_t)u->kmin + 1);
            int narr = 0;
            rfx_hip_group_table_arrays(&v->agg, 1, &narr);
            void *store = NULL;
            if (rfx_hip_malloc(g_ctx, &store, (size_t)narr * (size_t)range * 8) != RFX_OK) return upd_stop(u, fail_hip("group tables"));
            u->tmp[u->ntmp++] = store;
            int64_t *base = (int64_t *)store;
            rfx_group_tables_t gt;
            memset(&gt, 0, sizeof(gt));
            gt.kmin = u->kmin;
            gt.range = range;
            gt.nagg = 1;
            gt.d_first = base;
            gt.d_acc[0] = base + range;
            gt.d_cnt[0] = narr > 2 ? base + 2 * range : NULL;
            if (rfx_hip_group_tables_init(g_ctx, &v->agg, &gt) != RFX_OK ||
                rfx_hip_group_dense_accumulate(g_ctx, (const int64_t *)u->dk, u->wp.preds, u->wp.npred, u->wp.logic, &v->agg, nrows, 0, &gt) != RFX_OK ||
                rfx_hip_update_group(g_ctx, dcol, (const int64_t *)u->dk, u->d_ids, u->m, &v->agg, &gt) != RFX_OK)
                return upd_stop(u, fail_hip("grouped update"));
        }
    }
    *newcol = H.vector(out_type, nrows);
    if (rfx_hip_d2h(g_ctx, RFX_AS_RAW(*newcol), dcol, (size_t)nrows * 8) != RFX_OK) return upd_stop(u, fail_hip("read-back"));
    return UPD_GO;
}
/* the result table: the old columns (shared), replaced or extended by the updated ones (which it takes over) */
static obj_p upd_result(obj_p tab, const int64_t *mnames, obj_p *newcols, int nmap) {
    obj_p tnames = RFX_AS_LIST(tab)[0], tcols = RFX_AS_LIST(tab)[1];
    int64_t nnew = 0;
    for (int i = 0; i < nmap; i++)
        if (!table_col(tab, mnames[i])) {
            int dup = 0;
            for (int j = 0; j < i; j++) dup |= mnames[j] == mnames[i];
            if (!dup) nnew++;
        }
    obj_p rk = H.vector(RFX_TYPE_SYMBOL, tnames->len + nnew), rv = H.vector(RFX_TYPE_LIST, tnames->len + nnew);
    for (int64_t c = 0; c < tnames->len; c++) {
        RFX_AS_I64(rk)[c] = RFX_AS_I64(tnames)[c];
        obj_p col = NULL;
        for (int i = nmap - 1; i >= 0 && !col; i--)
            if (mnames[i] == RFX_AS_I64(tnames)[c] && newcols[i]) { col = newcols[i]; newcols[i] = NULL; }
        RFX_AS_LIST(rv)[c] = col ? col : H.clone(RFX_AS_LIST(tcols)[c]);
    }
    int64_t at = tnames->len;
    for (int i = 0; i < nmap; i++)
        if (newcols[i] && !table_col(tab, mnames[i])) {
            RFX_AS_I64(rk)[at] = mnames[i];
            RFX_AS_LIST(rv)[at++] = newcols[i];
            newcols[i] = NULL;
        }
    return H.table(rk, rv);
}

static obj_p update_impl(obj_p dict) {
    rfx_host_bind();
    if (!dict || dict->type != RFX_TYPE_DICT || RFX_AS_LIST(dict)[0]->type != RFX_TYPE_SYMBOL) return fail("update: expected a dict");
    obj_p from = dict_get(dict, "from");
    if (!from) return fail("'update' expects 'from' param");
    /* `from: 't` parses as (quote t): the in-place form on a global -- evaluated by the host only */
    if (from->type == RFX_TYPE_LIST) return delegate_update(dict, "from: is an expression (in-place update of a global)");
    obj_p tab = HOST_CALL(H.eval(from)); /* (not under our lock: the host may fan the evaluation out) */
    if (!tab || tab->type == RFX_TYPE_ERR) return tab;
    if (tab->type != RFX_TYPE_TABLE) {
        H.drop(tab);
        return delegate_update(dict, "from: does not evaluate to a table value");
    }
    upd_t u;
    memset(&u, 0, sizeof(u));
    u.tab = tab;
    u.wp.logic = RFX_AND;
    u.kmax = -1;
    obj_p where = dict_get(dict, "where"), by = dict_get(dict, "by");
    obj_p dkeys = RFX_AS_LIST(dict)[0], dvals = RFX_AS_LIST(dict)[1];
    const int64_t s_from = H.intern("from", 4), s_where = H.intern("where", 5), s_by = H.intern("by", 2);
    obj_p tcols = RFX_AS_LIST(tab)[1];
    u.nrows = u.m = tcols->len ? RFX_AS_LIST(tcols)[0]->len : 0;
    obj_p newcols[RFX_MAX_AGGS] = {0};
    int64_t mnames[RFX_MAX_AGGS];
    obj_p mexpr[RFX_MAX_AGGS];
    int nmap = 0, st = UPD_GO;
    for (int64_t i = 0; i < dkeys->len && st == UPD_GO; i++) {
        const int64_t k = RFX_AS_I64(dkeys)[i];
        if (k == s_from || k == s_where || k == s_by) continue;
        if (nmap >= RFX_MAX_AGGS) { st = upd_back(&u, "more than 8 mappings"); break; }
        mnames[nmap] = k;
        mexpr[nmap++] = RFX_AS_LIST(dvals)[i];
    }
    if (st == UPD_GO && nmap == 0) st = upd_back(&u, "no mapping");
    if (st == UPD_GO && u.nrows == 0) st = upd_back(&u, "empty table");
    for (int64_t i = 0; i < tcols->len && st == UPD_GO; i++)
        if (RFX_AS_LIST(tcols)[i]->len != u.nrows) st = upd_back(&u, "ragged table");
    if (st == UPD_GO && ensure_ctx1() != RFX_OK) st = g_refused_sharded ? upd_back(&u, "sharded operator layer: update is the host's") : upd_stop(&u, fail_ctx());
    if (st == UPD_GO && where) st = upd_where(&u, where, by);
    if (st == UPD_GO && by) st = upd_by(&u, by);
    for (int i = 0; i < nmap && st == UPD_GO; i++) {
        upd_val_t v;
        st = upd_value(&u, mexpr[i], by, &v);
        if (st == UPD_GO) st = upd_column(&u, table_col(tab, mnames[i]), by, &v, &newcols[i]);
    }
    obj_p res;
    if (st == UPD_GO) {
        res = upd_result(tab, mnames, newcols, nmap);
        g_last_gpu = 1;
    } else if (st == UPD_STOP) res = u.res;
    else { /* the host's: nothing of ours is left behind */
        for (int i = 0; i < u.ntmp; i++) rfx_hip_free(g_ctx, u.tmp[i]);
        u.ntmp = 0;
        qtmp_release();
        res = delegate_update(dict, u.why ? u.why : "unsupported");
    }
    for (int i = 0; i < nmap && i < RFX_MAX_AGGS; i++)
        if (newcols[i]) H.drop(newcols[i]);
    for (int i = 0; i < u.ntmp; i++) rfx_hip_free(g_ctx, u.tmp[i]);
    qtmp_release();
    H.drop(tab);
    return res;
}
rfx_obj_p rfx_update(rfx_obj_p dict) {
    op_begin();
    g_last_gpu = 0;
    obj_p r = update_impl(dict);
    op_end();
    return r;
}

/* ------------------------------------------------------------------------------------------------ single operators */
static obj_p cmp_impl(int op, obj_p x, obj_p y) {
    rfx_host_bind();
    if (!x || !y) return fail("cmp: null argument");
    if (!(x->type > 0 && col_ctype(x) && (y->type == -RFX_TYPE_I64 || y->type == -RFX_TYPE_F64 || (y->type > 0 && col_ctype(y))))) {
        if (H.bound == 1 && H.f[F_EQ + op]) return HOST_CALL(((rfx_binary_f)H.f[F_EQ + op])(x, y));
        return fail("cmp: only i64/f64 column (x) atom|column runs on the MI355X path");
    }
    if (y->type > 0 && y->len != x->len) return fail("length"); /* err_length, core/cmp.c:633-640 */
    if (ensure_ctx1() != RFX_OK) return refused2(F_EQ + op, x, y);
    rfx_pred_t p;
    memset(&p, 0, sizeof(p));
    const void *d;
    if (resident(x, 0, &d) != RFX_OK) return fail_hip("column upload");
    p.d_col = d;
    p.col_type = col_ctype(x);
    p.op = op;
    if (y->type == -RFX_TYPE_I64) { p.rhs_type = RFX_I64; p.rhs_i = y->i64; }
    else if (y->type == -RFX_TYPE_F64) { p.rhs_type = RFX_F64; p.rhs_f = y->f64; }
    else {
        if (resident(y, 0, &d) != RFX_OK) return fail_hip("column upload");
        p.d_rhs_col = d;
        p.rhs_type = col_ctype(y);
    }
    void *dm = NULL;
    if (rfx_hip_malloc(g_ctx, &dm, (size_t)x->len + 8) != RFX_OK) return fail_hip("mask");
    obj_p out = H.vector(RFX_TYPE_B8, x->len);
    int ok = rfx_hip_cmp_mask(g_ctx, &p, x->len, (int8_t *)dm) == RFX_OK && rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dm, (size_t)x->len) == RFX_OK;
    rfx_hip_free(g_ctx, dm);
    if (!ok) { H.drop(out); return fail_hip("cmp_mask"); }
    return out;
}
static obj_p cmp_op(int op, obj_p x, obj_p y) {
    op_begin();
    obj_p r = cmp_impl(op, x, y);
    op_end();
    return r;
}
/* ray_add / ray_sub / ray_mul / ray_fdiv / ray_div / ray_mod over an i64 / f64 vector and a vector or atom (binop_map, core/math.c:2280-2345) */
static obj_p arith_impl(int xop, int fidx, obj_p x, obj_p y) {
    rfx_host_bind();
    if (!x || !y) return fail("arith: null argument");
    const int xv = x->type > 0 && col_ctype(x) && x->type != RFX_TYPE_SYMBOL, yv = y->type > 0 && col_ctype(y) && y->type != RFX_TYPE_SYMBOL;
    const int xa = x->type == -RFX_TYPE_I64 || x->type == -RFX_TYPE_F64, ya = y->type == -RFX_TYPE_I64 || y->type == -RFX_TYPE_F64;
    if (!((xv && (yv || ya)) || (xa && yv))) {
        if (H.bound == 1 && H.f[fidx]) return HOST_CALL(((rfx_binary_f)H.f[fidx])(x, y));
        return fail("arith: only i64/f64 vector (x) vector|atom runs on the MI355X path");
    }
    if (xv && yv && x->len != y->len) return fail("length");
    if (ensure_ctx1() != RFX_OK) return refused2(fidx, x, y);
    rfx_agg_t a;
    memset(&a, 0, sizeof(a));
    a.kind = RFX_AGG_SUM;
    a.xop = xop;
    obj_p col = xv ? x : y, other = xv ? y : x;
    if (!xv) a.xflags = RFX_XF_SWAP; /* atom (op) vector */
    const void *d;
    if (resident(col, 0, &d) != RFX_OK) return fail_hip("column upload");
    a.d_col = d;
    a.col_type = col_ctype(col);
    if (other->type > 0) {
        if (resident(other, 0, &d) != RFX_OK) return fail_hip("column upload");
        a.d_xrhs_col = d;
        a.xrhs_type = col_ctype(other);
    } else if (other->type == -RFX_TYPE_I64) { a.xrhs_type = RFX_I64; a.xrhs_i = other->i64; }
    else { a.xrhs_type = RFX_F64; a.xrhs_f = other->f64; }
    const int64_t n = col->len;
    void *dout = NULL;
    if (rfx_hip_malloc(g_ctx, &dout, (size_t)(n ? n : 1) * 8) != RFX_OK) return fail_hip("arith");
    int32_t ot = RFX_I64;
    int ok = rfx_hip_eval_expr(g_ctx, &a, n, dout, &ot) == RFX_OK;
    obj_p out = NULL;
    if (ok) {
        out = H.vector(ot == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64, n);
        ok = n == 0 || rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dout, (size_t)n * 8) == RFX_OK;
    }
    rfx_hip_free(g_ctx, dout);
    if (!ok) { if (out) H.drop(out); return fail_hip("eval_expr"); }
    return out;
}
static obj_p arith_op(int xop, int fidx, obj_p x, obj_p y) {
    op_begin();
    obj_p r = arith_impl(xop, fidx, x, y);
    op_end();
    return r;
}
rfx_obj_p rfx_add(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_ADD, F_ADD, x, y); }
rfx_obj_p rfx_sub(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_SUB, F_SUB, x, y); }
rfx_obj_p rfx_mul(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_MUL, F_MUL, x, y); }
rfx_obj_p rfx_div(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_FDIV, F_FDIV, x, y); }
rfx_obj_p rfx_floordiv(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_DIV, F_DIV, x, y); } /* the reference's `/` (ray_div) */
rfx_obj_p rfx_mod(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_MOD, F_MOD, x, y); }      /* `%` (ray_mod) */

rfx_obj_p rfx_eq(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_EQ, x, y); }
rfx_obj_p rfx_ne(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_NE, x, y); }
rfx_obj_p rfx_lt(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_LT, x, y); }
rfx_obj_p rfx_gt(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_GT, x, y); }
rfx_obj_p rfx_le(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_LE, x, y); }
rfx_obj_p rfx_ge(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_GE, x, y); }

static obj_p logic_op(int logic, obj_p *x, int64_t n) {
    rfx_host_bind();
    if (n == 0) return rfx_host_b8(0); /* logic_map: (and) -> false, core/logic.c:96-97 */
    for (int64_t i = 0; i < n; i++)
        if (!x[i] || x[i]->type != RFX_TYPE_B8 || x[i]->len != x[0]->len) return fail("and/or: expected B8 masks of one length");
    if (ensure_ctx1() != RFX_OK) return refusedn(logic == RFX_AND ? F_AND : F_OR, x, n);
    int64_t len = x[0]->len;
    void *acc = NULL, *nxt = NULL;
    if (rfx_hip_malloc(g_ctx, &acc, (size_t)len + 8) != RFX_OK || rfx_hip_malloc(g_ctx, &nxt, (size_t)len + 8) != RFX_OK) return fail_hip("mask");
    int ok = rfx_hip_h2d(g_ctx, acc, RFX_AS_RAW(x[0]), (size_t)len) == RFX_OK;
    for (int64_t i = 1; i < n && ok; i++)
        ok = rfx_hip_h2d(g_ctx, nxt, RFX_AS_RAW(x[i]), (size_t)len) == RFX_OK && rfx_hip_mask_logic(g_ctx, logic, (int8_t *)acc, (const int8_t *)nxt, 0, len) == RFX_OK;
    obj_p out = H.vector(RFX_TYPE_B8, len);
    ok = ok && rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), acc, (size_t)len) == RFX_OK;
    rfx_hip_free(g_ctx, acc);
    rfx_hip_free(g_ctx, nxt);
    if (!ok) { H.drop(out); return fail_hip("mask_logic"); }
    return out;
}
rfx_obj_p rfx_and(rfx_obj_p *x, int64_t n) { return logic_op(RFX_AND, x, n); }
rfx_obj_p rfx_or(rfx_obj_p *x, int64_t n) { return logic_op(RFX_OR, x, n); }

/* ---- `and` / `or` as the SPECIAL FORMS the reference registers (FN_SPECIAL_FORM, core/env.c:224-225): the arms arrive UNEVALUATED and
 * logic_map evaluates them itself (core/logic.c:89-260).  rfx_and_sf / rfx_or_sf take the same (obj_p *arms, n): when every arm is a
 * comparison -- or a nested and / or of comparisons -- over i64 / f64 vectors (a symbol the host's eval resolves, or the vector object
 * itself) and atoms, the whole tree becomes one B8 mask on the device (K2 masks + rfx_hip_mask_logic, no host round trip between the arms);
 * arms that are already B8 masks take rfx_and / rfx_or; anything else is the host's own ray_and / ray_or. ---- */
#define SF_MAX_COLS 16
typedef struct {
    int n;
    obj_p src[SF_MAX_COLS];  /* the operand as written: a symbol atom or a vector object */
    obj_p val[SF_MAX_COLS];  /* what it evaluates to (owned) */
    int64_t name[SF_MAX_COLS];
} sf_cols_t;
/* a copy of `e` whose vector / symbol operands are replaced by synthetic column symbols (collected in c); NULL: shape not covered */
static obj_p sf_rewrite(obj_p e, sf_cols_t *c, int top) {
    if (!e) return NULL;
    if (e->type == RFX_TYPE_LIST) {
        if (e->len != 3 && !(e->len >= 2 && (fn_id(RFX_AS_LIST(e)[0]) == F_AND || fn_id(RFX_AS_LIST(e)[0]) == F_OR))) return NULL;
        const int f = fn_id(RFX_AS_LIST(e)[0]);
        if (f < 0 || (top && !((f >= F_EQ && f <= F_GE) || f == F_AND || f == F_OR))) return NULL;
        obj_p out = H.vector(RFX_TYPE_LIST, e->len);
        RFX_AS_LIST(out)[0] = H.clone(RFX_AS_LIST(e)[0]);
        for (int64_t i = 1; i < e->len; i++) {
            const int sub_top = (f == F_AND || f == F_OR); /* arms of and / or must be boolean trees again; operands of a comparison may be arithmetic */
            obj_p r = sf_rewrite(RFX_AS_LIST(e)[i], c, sub_top);
            if (!r) {
                for (int64_t j = i; j < e->len; j++) RFX_AS_LIST(out)[j] = H.null_obj ? H.null_obj : rfx_host_null();
                H.drop(out);
                return NULL;
            }
            RFX_AS_LIST(out)[i] = r;
        }
        return out;
    }
    if (top) return NULL; /* an arm that is not a call */
    if (e->type == -RFX_TYPE_I64 || e->type == -RFX_TYPE_F64) return H.clone(e);
    if (e->type == -RFX_TYPE_SYMBOL || (e->type > 0 && col_ctype(e) && e->type != RFX_TYPE_SYMBOL)) {
        int k = 0;
        for (; k < c->n; k++)
            if (c->src[k] == e || (e->type == -RFX_TYPE_SYMBOL && c->src[k]->type == -RFX_TYPE_SYMBOL && c->src[k]->i64 == e->i64)) break;
        if (k == c->n) {
            if (c->n >= SF_MAX_COLS) return NULL;
            obj_p v = H.eval(e); /* a symbol: the host's binding; a vector: itself */
            if (!v || v->type <= 0 || !col_ctype(v) || v->type == RFX_TYPE_SYMBOL || (c->n > 0 && v->len != c->val[0]->len)) {
                if (v) H.drop(v);
                return NULL;
            }
            char nm[16];
            snprintf(nm, sizeof(nm), "rfxsf%d", c->n);
            c->src[c->n] = e;
            c->val[c->n] = v;
            c->name[c->n] = H.intern(nm, (int64_t)strlen(nm));
            c->n++;
        }
        obj_p sym = H.i64(c->name[k]);
        sym->type = -RFX_TYPE_SYMBOL;
        return sym;
    }
    return NULL;
}
static obj_p sf_logic_impl(int f, obj_p *x, int64_t n) {
    rfx_host_bind();
    if (n == 0) return rfx_host_b8(0); /* logic_map: (and) -> false, core/logic.c:96-97 */
    int all_masks = 1;
    for (int64_t i = 0; i < n; i++) all_masks = all_masks && x[i] && x[i]->type == RFX_TYPE_B8;
    if (all_masks) return logic_op(f == F_AND ? RFX_AND : RFX_OR, x, n); /* bound through a loader that evaluates the arguments first */
    const char *why = "an arm is not a comparison tree over i64 / f64 vectors";
    sf_cols_t c;
    memset(&c, 0, sizeof(c));
    obj_p tree = H.vector(RFX_TYPE_LIST, n + 1), tab = NULL, res = NULL;
    obj_p fo = H.i64((int64_t)(intptr_t)OUR_FN[f]);
    fo->type = RFX_TYPE_VARY;
    RFX_AS_LIST(tree)[0] = fo;
    int ok = 1;
    for (int64_t i = 0; i < n; i++) {
        obj_p r = ok ? sf_rewrite(x[i], &c, 1) : NULL;
        if (!r) ok = 0;
        RFX_AS_LIST(tree)[1 + i] = r ? r : (H.null_obj ? H.null_obj : rfx_host_null());
    }
    if (ok && c.n == 0) { ok = 0; why = "no vector operand"; }
    if (ok && ensure_ctx1() != RFX_OK) {
        if (g_refused_sharded) why = "sharded operator layer: the comparison tree is the host's"; /* (handed back below, like any shape that is not ours) */
        else res = fail_ctx();
        ok = 0;
    }
    if (ok) {
        obj_p names = H.vector(RFX_TYPE_SYMBOL, c.n), cols = H.vector(RFX_TYPE_LIST, c.n);
        for (int k = 0; k < c.n; k++) {
            RFX_AS_I64(names)[k] = c.name[k];
            RFX_AS_LIST(cols)[k] = c.val[k];
            c.val[k] = NULL; /* the table owns it now */
        }
        tab = H.table(names, cols);
        const int64_t nrows = RFX_AS_LIST(RFX_AS_LIST(tab)[1])[0]->len;
        int8_t *mask = NULL;
        const int rc = mask_of_expr(tab, tree, nrows, &mask);
        if (rc == 0) {
            res = H.vector(RFX_TYPE_B8, nrows);
            if (nrows && rfx_hip_d2h(g_ctx, RFX_AS_RAW(res), mask, (size_t)nrows) != RFX_OK) {
                H.drop(res);
                res = fail_hip("mask read-back");
            }
            rfx_hip_free(g_ctx, mask);
        } else if (rc == -2) res = fail_hip("and/or: device");
        else ok = 0;
        qtmp_release();
    }
    for (int k = 0; k < c.n; k++)
        if (c.val[k]) H.drop(c.val[k]);
    H.drop(tree);
    if (tab) H.drop(tab);
    if (res) return res;
    /* not covered: the host's own special form evaluates the arms */
    if (H.bound == 1 && H.f[f]) return HOST_CALL(((rfx_vary_f)H.f[f])(x, n));
    char b[256];
    snprintf(b, sizeof(b), "and/or (special form): not covered by the MI355X path (%s) and no host function to delegate to", why);
    return fail(b);
}
static obj_p sf_logic(int f, obj_p *x, int64_t n) {
    op_begin();
    obj_p r = sf_logic_impl(f, x, n);
    op_end();
    return r;
}
rfx_obj_p rfx_and_sf(rfx_obj_p *x, int64_t n) { return sf_logic(F_AND, x, n); }
rfx_obj_p rfx_or_sf(rfx_obj_p *x, int64_t n) { return sf_logic(F_OR, x, n); }

/* ---- the operators beside rfx_select over the shards (round 5): what the reference parallelises over its pool for every FN_AGGR
 * built-in (aggr_map core/aggr.c:375, unop_fold core/math.c:2176-2231), planned through rfx_exec on every shard ---- */
/* a per-call device copy of a host vector, every shard its row range (rfx_exec_split) -- a column of the query like any other */
static int transient_sharded(obj_p v, const void **dev) {
    if (g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) return RFX_ELIMIT;
    const size_t esz = v->type == RFX_TYPE_B8 ? 1 : 8;
    void *devs[RFX_MAX_SHARDS];
    int rc = shards_alloc(devs, v->len, esz);
    if (rc != RFX_OK) return rc;
    memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
    for (int s = 0; s < g_nshards; s++) g_qtmp[g_nqtmp].d[s] = devs[s];
    g_nqtmp++; /* (released by qtmp_release, shard by shard) */
    rc = payload_upload(v->type == RFX_TYPE_B8 ? RFX_TYPE_B8 : RFX_TYPE_I64, devs, RFX_AS_RAW(v), v->len);
    g_stat[ST_UPLOADS]++;
    if (rc != RFX_OK) return rc;
    *dev = devs[0];
    return qcol_add(devs);
}
/* the ids of a lazy MAPFILTER pair cut at the shards' row boundaries: piece s = the ids inside shard s's rows of an nrows-row column.  Filter ids
 * ascend (ops_where, core/ops.c:254-273), so the pieces are sub-ranges of the vector, found by binary search; every piece is then PROVEN to lie
 * inside its shard's rows on the device (min / max of the piece) -- ids in any other order answer 1 and the caller hands the pair to the host */
static int sel_ids_sharded(obj_p ids, int64_t nrows, const int64_t **d_ids, int64_t *cnt) {
    const int64_t *p = RFX_AS_I64(ids), n = ids->len;
    int64_t cut[RFX_MAX_SHARDS + 1];
    cut[0] = 0;
    for (int s = 1; s < g_nshards; s++) {
        int64_t r0, lo = cut[s - 1], hi = n;
        rfx_exec_split(nrows, g_nshards, s, &r0, NULL);
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (p[mid] < r0) lo = mid + 1;
            else hi = mid;
        }
        cut[s] = lo;
    }
    cut[g_nshards] = n;
    if (g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) return RFX_ELIMIT;
    memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
    void **devs = g_qtmp[g_nqtmp++].d;
    int rc = RFX_OK, outside = 0;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        int64_t r0, len;
        rfx_exec_split(nrows, g_nshards, s, &r0, &len);
        cnt[s] = cut[s + 1] - cut[s];
        rfx_hip_ctx_bind_thread(g_ctxs[s]);
        rc = rfx_hip_malloc(g_ctxs[s], &devs[s], (size_t)(cnt[s] ? cnt[s] : 1) * 8);
        if (rc == RFX_OK && cnt[s]) rc = rfx_hip_h2d_pipelined(g_ctxs[s], devs[s], p + cut[s], (size_t)cnt[s] * 8);
        if (rc == RFX_OK && cnt[s]) {
            int64_t mn = 0, mx = -1, seen = 0;
            rc = rfx_hip_scope_i64(g_ctxs[s], (const int64_t *)devs[s], NULL, 0, RFX_AND, cnt[s], &mn, &mx, &seen);
            if (rc == RFX_OK && (mn < r0 || mx >= r0 + len)) outside = 1;
        }
        d_ids[s] = (const int64_t *)devs[s];
    }
    rfx_hip_ctx_bind_thread(g_ctx);
    g_stat[ST_UPLOADS]++;
    return rc != RFX_OK ? rc : (outside ? 1 : RFX_OK);
}
/* one aggregate of a column over every shard (the whole column, or its rows at per-shard ids) */
static int fold_sharded(const rfx_agg_t *a, int64_t nrows, const int64_t *const *d_ids, const int64_t *cnt, rfx_value_t *v) {
    rfx_query_t Q;
    memset(&Q, 0, sizeof(Q));
    Q.aggs = a;
    Q.nagg = 1;
    Q.logic = RFX_AND;
    Q.nrows = nrows;
    Q.cols = g_qcols;
    Q.ncols = g_nqcols;
    Q.d_sel_ids = d_ids;
    Q.sel_count = cnt;
    return rfx_exec_filter_aggr(g_x, &Q, v, NULL);
}

static obj_p where_impl(obj_p mask) {
    rfx_host_bind();
    if (!mask || mask->type != RFX_TYPE_B8) return fail("where: expected a B8 mask"); /* err_type, core/items.c:1395 */
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1) { /* every shard turns ITS rows of the mask into ids (global: its row offset added), the runs concatenated in shard order */
        const void *dms = NULL;
        rfx_query_t Q;
        rfx_ids_t ids;
        memset(&Q, 0, sizeof(Q));
        int rc = mask->len ? transient_sharded(mask, &dms) : RFX_OK;
        Q.d_mask = (const int8_t *)dms;
        Q.logic = RFX_AND;
        Q.nrows = mask->len;
        Q.cols = g_qcols;
        Q.ncols = g_nqcols;
        if (rc == RFX_OK && mask->len == 0) { qtmp_release(); return H.vector(RFX_TYPE_I64, 0); }
        if (rc == RFX_OK) rc = rfx_exec_where(g_x, &Q, &ids);
        if (rc != RFX_OK) { qtmp_release(); return fail(rc == RFX_OK ? "where" : (rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error())); }
        obj_p outv = H.vector(RFX_TYPE_I64, ids.total);
        int64_t at = 0;
        int ok2 = 1;
        for (int sh = 0; sh < ids.nshards && ok2; sh++) {
            if (!ids.count[sh]) continue;
            rfx_hip_ctx_bind_thread(g_ctxs[sh]);
            ok2 = rfx_hip_d2h(g_ctxs[sh], (char *)RFX_AS_RAW(outv) + (size_t)at * 8, ids.d_ids[sh], (size_t)ids.count[sh] * 8) == RFX_OK;
            at += ids.count[sh];
        }
        rfx_hip_ctx_bind_thread(g_ctx);
        rfx_exec_ids_free(g_x, &ids);
        qtmp_release();
        if (!ok2) { H.drop(outv); return fail_hip("where"); }
        return outv;
    }
    const void *dm;
    if (transient(mask, &dm) != RFX_OK) return fail_hip("mask upload"); /* a mask is a temporary: per-call scratch, never cached */
    int64_t count = 0;
    if (rfx_hip_where_begin(g_ctx, NULL, 0, RFX_AND, (const int8_t *)dm, mask->len, &count) != RFX_OK) return fail_hip("where");
    obj_p out = H.vector(RFX_TYPE_I64, count);
    void *di = NULL;
    int ok = 1;
    if (count > 0) {
        ok = rfx_hip_malloc(g_ctx, &di, (size_t)count * 8) == RFX_OK && rfx_hip_where_emit(g_ctx, 0, (int64_t *)di) == RFX_OK &&
             rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), di, (size_t)count * 8) == RFX_OK;
        if (di) rfx_hip_free(g_ctx, di);
    }
    if (!ok) { H.drop(out); return fail_hip("where"); }
    return out;
}

rfx_obj_p rfx_where(rfx_obj_p mask) {
    op_begin();
    obj_p r = where_impl(mask);
    op_end();
    return r;
}

/* ------------------------------------------------------------------------------------------------ equi-joins
 * (left-join [keys] x y) / (inner-join [keys] x y): ray_left_join / ray_inner_join, core/join.c:158-298 -- vary_f over (key symbols,
 * left table, right table).  Index = per left row the first right row with an equal key tuple (index_left_join_obj,
 * core/index.c:2886-2928): the group-by's first-occurrence table over the right keys (zero aggregates), probed with the left keys. */
static obj_p join_impl(int inner, obj_p *x, int64_t n) {
    rfx_host_bind();
    const int fidx = inner ? F_IJ : F_LJ;
    if (n != 3 || !x[0] || !x[1] || !x[2]) return fail("join: expected (keys, left table, right table)");
    if (x[0]->type != RFX_TYPE_SYMBOL || x[1]->type != RFX_TYPE_TABLE || x[2]->type != RFX_TYPE_TABLE) return fail("join: expected (symbol vector, table, table)");
    obj_p ksyms = x[0], lt = x[1], rt = x[2];
    obj_p lnames = RFX_AS_LIST(lt)[0], lcols = RFX_AS_LIST(lt)[1], rnames = RFX_AS_LIST(rt)[0], rcols = RFX_AS_LIST(rt)[1];
    const int64_t nl = lcols->len ? RFX_AS_LIST(lcols)[0]->len : 0, nr = rcols->len ? RFX_AS_LIST(rcols)[0]->len : 0;
    const int nk = (int)ksyms->len;
    const char *why = NULL;
    void *tmp[4 * RFX_MAX_KEYS + 8];
    int ntmp = 0;
    obj_p res = NULL;
    if (nl == 0 || nr == 0) return H.clone(lt); /* core/join.c:171-172 */
    if (nk < 1 || nk > RFX_MAX_KEYS) { why = "1..8 key columns"; goto out; }
    obj_p lk[RFX_MAX_KEYS], rk[RFX_MAX_KEYS];
    const void *dlk[RFX_MAX_KEYS], *drk[RFX_MAX_KEYS];
    for (int i = 0; i < nk; i++) {
        lk[i] = table_col(lt, RFX_AS_I64(ksyms)[i]);
        rk[i] = table_col(rt, RFX_AS_I64(ksyms)[i]);
        if (!lk[i] || !rk[i] || col_ctype(lk[i]) != RFX_I64 || col_ctype(rk[i]) != RFX_I64 || lk[i]->type != rk[i]->type) { why = "join key is not an 8-byte integer column of both tables"; goto out; }
    }
    for (int64_t i = 0; i < lcols->len; i++) if (!col_ctype(RFX_AS_LIST(lcols)[i])) { why = "non-8-byte column"; goto out; }
    for (int64_t i = 0; i < rcols->len; i++) {
        obj_p rc = RFX_AS_LIST(rcols)[i], lc = table_col(lt, RFX_AS_I64(rnames)[i]);
        if (!col_ctype(rc)) { why = "non-8-byte column"; goto out; }
        if (lc && lc->type != rc->type) return fail("join: a column has different types in the two tables"); /* err_type, core/join.c:50-51 */
    }
    if (ensure_ctx1() != RFX_OK) return refusedn(fidx, x, n);
    for (int i = 0; i < nk; i++)
        if (resident(lk[i], 0, &dlk[i]) != RFX_OK || resident(rk[i], 0, &drk[i]) != RFX_OK) { res = fail_hip("column upload"); goto done; }
#define JOIN_TMP(ptr, bytes) do { ptr = NULL; if (rfx_hip_malloc(g_ctx, &ptr, (bytes)) != RFX_OK) { res = fail_hip("join scratch"); goto done; } tmp[ntmp++] = ptr; } while (0)
    /* the join index -- per left row the first right row with an equal key tuple, or null -- is the planner's (rfx_exec_join_index: dense
     * first-occurrence table or the hashed one, composite key or the reference's row hash + the tuple check) */
    void *ids = NULL;
    JOIN_TMP(ids, (size_t)nl * 8);
    {
        int collision = 0;
        const int jrc = rfx_exec_join_index(g_x, dlk, drk, nk, nl, nr, (int64_t *)ids, &collision);
        if (jrc != RFX_OK && collision) { why = "row-hash collision between two key tuples"; goto out; }
        if (jrc != RFX_OK) { res = fail(rfx_exec_last_error(g_x)); goto done; }
    }
    /* result columns: keys, then the other left columns, then the right-only ones (ray_union / ray_except order, core/join.c:83-156) */
    {
        int64_t names[64];
        int ncol = 0;
        for (int i = 0; i < nk; i++) names[ncol++] = RFX_AS_I64(ksyms)[i];
        for (int pass = 0; pass < 2; pass++) {
            obj_p nm = pass ? rnames : lnames;
            for (int64_t i = 0; i < nm->len && ncol < 64; i++) {
                int64_t sy = RFX_AS_I64(nm)[i];
                int dup = 0;
                for (int j = 0; j < ncol; j++) dup |= names[j] == sy;
                if (!dup) names[ncol++] = sy;
            }
        }
        if (ncol >= 64) { why = "too many columns"; goto out; }
        void *lids = NULL, *rids = NULL, *dcol = NULL;
        int64_t nout = nl;
        if (inner) { /* matched left rows in order, paired with their right rows (index_inner_join_obj) */
            rfx_pred_t p;
            memset(&p, 0, sizeof(p));
            p.d_col = ids; p.col_type = RFX_I64; p.op = RFX_NE; p.rhs_type = RFX_I64; p.rhs_i = RFX_NULL_I64;
            if (rfx_hip_where_begin(g_ctx, &p, 1, RFX_AND, NULL, nl, &nout) != RFX_OK) { res = fail_hip("join where"); goto done; }
            JOIN_TMP(lids, (size_t)(nout ? nout : 1) * 8);
            JOIN_TMP(rids, (size_t)(nout ? nout : 1) * 8);
            if (rfx_hip_where_emit(g_ctx, 0, (int64_t *)lids) != RFX_OK || (nout && rfx_hip_gather(g_ctx, ids, (const int64_t *)lids, nout, rids) != RFX_OK)) { res = fail_hip("join where"); goto done; }
        }
        JOIN_TMP(dcol, (size_t)(nout ? nout : 1) * 8);
        obj_p rk_ = H.vector(RFX_TYPE_SYMBOL, ncol), rv = H.vector(RFX_TYPE_LIST, ncol);
        int ok = 1;
        for (int c = 0; c < ncol; c++) {
            RFX_AS_I64(rk_)[c] = names[c];
            obj_p lc = table_col(lt, names[c]), rc = table_col(rt, names[c]);
            const int iskey = c < nk;
            obj_p o = NULL;
            if (!inner && (iskey || !rc)) o = H.clone(lc); /* left join: key columns and left-only columns are the left table's own */
            else {
                obj_p src = (inner ? (rc ? rc : lc) : rc);
                o = H.vector(src->type, nout);
                const void *dsrc, *dleft = NULL;
                ok = ok && resident(src, 0, &dsrc) == RFX_OK;
                if (ok && !inner && lc) ok = resident(lc, 0, &dleft) == RFX_OK;
                if (ok && nout) {
                    if (inner) ok = rfx_hip_gather(g_ctx, dsrc, (const int64_t *)(rc ? rids : lids), nout, dcol) == RFX_OK;
                    else ok = rfx_hip_gather_or(g_ctx, dsrc, dleft, (const int64_t *)ids, nout, col_ctype(src) == RFX_F64 ? 0x7FF8000000000000ull : 0x8000000000000000ull, dcol) == RFX_OK;
                    ok = ok && rfx_hip_d2h(g_ctx, RFX_AS_RAW(o), dcol, (size_t)nout * 8) == RFX_OK;
                }
            }
            RFX_AS_LIST(rv)[c] = o;
        }
        if (!ok) { H.drop(rk_); H.drop(rv); res = fail_hip("join columns"); goto done; }
        res = H.table(rk_, rv);
        g_last_gpu = 1;
        goto done;
    }
out:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    ntmp = 0;
    if (H.bound == 1 && H.f[fidx]) res = HOST_CALL(((rfx_vary_f)H.f[fidx])(x, n));
    else {
        char b[320];
        snprintf(b, sizeof(b), "join: shape not covered by the MI355X path (%s) and no host function to delegate to", why ? why : "unsupported");
        res = fail(b);
    }
done:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    return res;
#undef JOIN_TMP
}
static obj_p join_op(int inner, obj_p *x, int64_t n) {
    op_begin();
    g_last_gpu = 0;
    obj_p r = join_impl(inner, x, n);
    g_stat[g_last_gpu ? ST_JOIN_GPU : ST_JOIN_DELEGATED]++;
    op_end();
    return r;
}
rfx_obj_p rfx_left_join(rfx_obj_p *x, int64_t n) { return join_op(0, x, n); }
rfx_obj_p rfx_inner_join(rfx_obj_p *x, int64_t n) { return join_op(1, x, n); }

static obj_p at_impl(obj_p col, obj_p ids) {
    rfx_host_bind();
    if (!col || !ids || !col_ctype(col) || ids->type != RFX_TYPE_I64) return fail("at: expected (i64|f64 column, I64 ids)");
    if (ensure_ctx1() != RFX_OK) return (g_refused_sharded && H.bound == 1 && g_host_at) ? HOST_CALL(((rfx_binary_f)g_host_at)(col, ids)) : fail_ctx();
    const void *dc, *di;
    if (resident(col, 0, &dc) != RFX_OK || transient(ids, &di) != RFX_OK) return fail_hip("upload");
    obj_p out = H.vector(col->type, ids->len);
    void *dout = NULL;
    /* ids come from the caller: null / negative / out-of-range ids read as the typed null (at_vec_*_by_i64, core/items.c:53-72) */
    int ok = rfx_hip_malloc(g_ctx, &dout, (size_t)ids->len * 8 + 8) == RFX_OK &&
             rfx_hip_gather_checked(g_ctx, dc, col->len, col_ctype(col), (const int64_t *)di, ids->len, dout) == RFX_OK &&
             rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dout, (size_t)ids->len * 8) == RFX_OK;
    if (dout) rfx_hip_free(g_ctx, dout);
    if (!ok) { H.drop(out); return fail_hip("gather"); }
    return out;
}
rfx_obj_p rfx_at(rfx_obj_p col, rfx_obj_p ids) {
    op_begin();
    obj_p r = at_impl(col, ids);
    op_end();
    return r;
}

/* ---- grouped aggregates over a lazy MAPGROUP (val, index) pair (core/group.c:26-46; aggr_sum(val, index) ... core/aggr.c:1078-2063) ----
 * `index` is the reference's 7-slot group index (index_group_build, core/index.c:1696-1699):
 *   [0] type  [1] group count  [2] group ids  [3] shift  [4] source column  [5] filter ids  [6] first ids
 * INDEX_TYPE_IDS:   row i (= position in the filter, if any) belongs to group [2][i]; value row x = filter ? filter[i] : i
 * INDEX_TYPE_SHIFT: [2] is the key TABLE (slot -> group id) and the group of row x is [2][source[x] - shift]
 * (AGGR_ITER, core/aggr.c:73-161).  On the device both are a dense group-by: IDS keyed by the id column over [0, groups), SHIFT keyed
 * by the source column over [shift, shift + table length) -- whose first-occurrence ranking reproduces the table's ids, so the table
 * itself is not even read.  Under a filter the value (and source) column is gathered by the filter ids first.  The parted /
 * window index flavours (core/aggr.c:126-159) go back to the host's own aggregate. */
static obj_p fold_mapgroup(int f, int kind, obj_p x) {
    obj_p val = RFX_AS_LIST(x)[0], index = RFX_AS_LIST(x)[1];
    const char *why = NULL;
    void *tmp[8];
    int ntmp = 0;
    obj_p res = NULL;
    if (!index || index->type != RFX_TYPE_LIST || index->len != 7) return fail("aggregate: malformed group index");
    obj_p *ix = RFX_AS_LIST(index);
    const int64_t itype = ix[0]->i64, groups = ix[1]->i64;
    obj_p gids = ix[2], source = ix[4], filter = ix[5];
    if (!(val->type > 0 && col_ctype(val) && val->type != RFX_TYPE_SYMBOL)) { why = "value column type"; goto out; }
    if (itype != RFX_INDEX_TYPE_IDS && itype != RFX_INDEX_TYPE_SHIFT) { why = "parted / window index"; goto out; }
    if (!gids || gids->type != RFX_TYPE_I64 || groups < 0) { why = "group ids"; goto out; }
    if (itype == RFX_INDEX_TYPE_SHIFT && !(source && source->type > 0 && col_ctype(source) == RFX_I64)) { why = "source column"; goto out; }
    const int filtered = filter && filter->type == RFX_TYPE_I64;
    const int64_t n = filtered ? filter->len : (itype == RFX_INDEX_TYPE_IDS ? gids->len : source->len);
    if (itype == RFX_INDEX_TYPE_IDS && gids->len != n) return fail("aggregate: group ids / filter length mismatch");
    if (!filtered && val->len != n) return fail("length");
    const int out_f64 = kind == RFX_AGG_AVG || (kind != RFX_AGG_COUNT && col_ctype(val) == RFX_F64);
    if (groups == 0 || n == 0) return H.vector(out_f64 ? RFX_TYPE_F64 : RFX_TYPE_I64, 0);
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1) {
        /* over the shards: a dense group-by keyed by the index's id column (IDS) or its source column (SHIFT), planned like any by: -- every
         * shard scatters its rows, the tables merge, the groups come out in first-occurrence order = the index's group ids.  A filtered index
         * aligns its ids with filter positions, not rows: the host's own aggregate. */
        if (filtered) { g_refused_sharded = 1; return refused1(f, x); }
        const void *dvs = NULL, *dks = NULL;
        int rc = resident(val, 0, &dvs);
        if (rc == RFX_OK) rc = itype == RFX_INDEX_TYPE_IDS ? transient_sharded(gids, &dks) : resident(source, 0, &dks);
        if (rc != RFX_OK) { qtmp_release(); return fail_hip("column upload"); }
        rfx_agg_t as;
        memset(&as, 0, sizeof(as));
        as.d_col = dvs;
        as.col_type = col_ctype(val);
        as.kind = kind;
        rfx_query_t Q;
        memset(&Q, 0, sizeof(Q));
        const void *dkeys[1] = {dks};
        Q.aggs = &as;
        Q.nagg = 1;
        Q.logic = RFX_AND;
        Q.nkeys = 1;
        Q.d_keys = dkeys;
        Q.nrows = n;
        Q.cols = g_qcols;
        Q.ncols = g_nqcols;
        Q.flags = RFX_Q_SLICED;
        rfx_groups_t R;
        rc = rfx_exec_group_by(g_x, &Q, &R);
        if (rc != RFX_OK) { qtmp_release(); return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error()); }
        if (R.groups != groups) {
            rfx_exec_groups_free(g_x, &R);
            qtmp_release();
            why = "group count of the index does not match its rows";
            goto out;
        }
        obj_p outv = H.vector(out_f64 ? RFX_TYPE_F64 : RFX_TYPE_I64, groups);
        const void *srcs[1] = {R.d_results[0]};
        void *dsts[1] = {RFX_AS_RAW(outv)};
        rc = rfx_exec_groups_fetch_all(g_x, &R, 1, srcs, dsts);
        rfx_exec_groups_free(g_x, &R);
        qtmp_release();
        if (rc != RFX_OK) { H.drop(outv); return fail_hip("group emit"); }
        return outv;
    }
    {
        const void *dv = NULL, *dk = NULL, *dfl = NULL;
        if (resident(val, 0, &dv) != RFX_OK) { res = fail_hip("column upload"); goto done; }
        if (filtered) {
            if (transient(filter, &dfl) != RFX_OK) { res = fail_hip("filter upload"); goto done; }
            void *g = NULL;
            if (rfx_hip_malloc(g_ctx, &g, (size_t)n * 8) != RFX_OK) { res = fail_hip("scratch"); goto done; }
            tmp[ntmp++] = g;
            if (rfx_hip_gather_checked(g_ctx, dv, val->len, col_ctype(val), (const int64_t *)dfl, n, g) != RFX_OK) { res = fail_hip("gather"); goto done; }
            dv = g;
        }
        int64_t kmin = 0, range = groups;
        if (itype == RFX_INDEX_TYPE_IDS) {
            if (transient(gids, &dk) != RFX_OK) { res = fail_hip("group ids upload"); goto done; }
        } else {
            kmin = ix[3]->i64;
            range = gids->len;
            if (resident(source, 0, &dk) != RFX_OK) { res = fail_hip("column upload"); goto done; }
            if (filtered) {
                void *g = NULL;
                if (rfx_hip_malloc(g_ctx, &g, (size_t)n * 8) != RFX_OK) { res = fail_hip("scratch"); goto done; }
                tmp[ntmp++] = g;
                if (rfx_hip_gather_checked(g_ctx, dk, source->len, RFX_I64, (const int64_t *)dfl, n, g) != RFX_OK) { res = fail_hip("gather"); goto done; }
                dk = g;
            }
        }
        if (range <= 0) { why = "empty key table"; goto out; }
        rfx_agg_t a;
        memset(&a, 0, sizeof(a));
        a.d_col = dv;
        a.col_type = col_ctype(val);
        a.kind = kind;
        int narr = 0;
        rfx_hip_group_table_arrays(&a, 1, &narr);
        void *store = NULL;
        if (rfx_hip_malloc(g_ctx, &store, (size_t)narr * (size_t)range * 8) != RFX_OK) { res = fail_hip("group tables"); goto done; }
        tmp[ntmp++] = store;
        int64_t *base = (int64_t *)store;
        rfx_group_tables_t gt;
        memset(&gt, 0, sizeof(gt));
        gt.kmin = kmin;
        gt.range = range;
        gt.nagg = 1;
        gt.d_first = base;
        gt.d_acc[0] = base + range;
        gt.d_cnt[0] = narr > 2 ? base + 2 * range : NULL;
        int64_t ng = 0;
        if (rfx_hip_group_tables_init(g_ctx, &a, &gt) != RFX_OK || rfx_hip_group_dense_accumulate(g_ctx, (const int64_t *)dk, NULL, 0, RFX_AND, &a, n, 0, &gt) != RFX_OK ||
            rfx_hip_group_rank(g_ctx, &gt, n, &ng) != RFX_OK) { res = fail_hip("group-by over the index"); goto done; }
        if (ng != groups) { why = "group count of the index does not match its rows"; goto out; }
        void *dout = NULL;
        if (rfx_hip_malloc(g_ctx, &dout, (size_t)groups * 8) != RFX_OK) { res = fail_hip("result"); goto done; }
        tmp[ntmp++] = dout;
        void *ptrs[1] = {dout};
        obj_p out = H.vector(out_f64 ? RFX_TYPE_F64 : RFX_TYPE_I64, groups);
        if (rfx_hip_group_emit(g_ctx, &a, &gt, NULL, NULL, ptrs) != RFX_OK || rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dout, (size_t)groups * 8) != RFX_OK) {
            H.drop(out);
            res = fail_hip("group emit");
            goto done;
        }
        res = out;
        goto done;
    }
out:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    ntmp = 0;
    if (H.bound == 1 && H.f[f]) res = HOST_CALL(((rfx_unary_f)H.f[f])(x));
    else {
        char b[256];
        snprintf(b, sizeof(b), "aggregate over a MAPGROUP pair: not covered by the MI355X path (%s) and no host function to delegate to", why ? why : "unsupported");
        res = fail(b);
    }
done:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    return res;
}

/* (rfx_group keys): the reference's group index of an I64 key column (index_group_i64_scoped, core/index.c:2002-2092) built on the
 * device: first-occurrence table (K7), rank (K8), then either the key table (INDEX_TYPE_SHIFT, range <= INDEX_SCOPE_LIMIT = 524 288)
 * or the per-row id vector (INDEX_TYPE_IDS).  Slots as index_group_build lays them out; sparse keys (range > rows) are the host's. */
static obj_p group_impl(obj_p keys) {
    rfx_host_bind();
    if (!keys || keys->type <= 0 || col_ctype(keys) != RFX_I64) return fail("group: expected an i64-like vector");
    const int64_t n = keys->len;
    if (ensure_ctx1() != RFX_OK) return (g_refused_sharded && H.bound == 1 && g_host_group) ? HOST_CALL(((rfx_unary_f)g_host_group)(keys)) : fail_ctx();
    const void *dk = NULL;
    if (n && resident(keys, 0, &dk) != RFX_OK) return fail_hip("column upload");
    int64_t kmin = 0, kmax = -1, seen = 0;
    if (n && rfx_hip_scope_i64(g_ctx, (const int64_t *)dk, NULL, 0, RFX_AND, n, &kmin, &kmax, &seen) != RFX_OK) return fail_hip("scope");
    const uint64_t range = n ? (uint64_t)kmax - (uint64_t)kmin + 1 : 0;
    if (n && !(range != 0 && range <= (uint64_t)n && kmin != RFX_NULL_I64)) return fail("group: sparse or null keys are not built on the MI355X path");
    void *store = NULL, *dfirst = NULL, *dids = NULL;
    obj_p res = NULL, gids = NULL, firsts = NULL;
    int64_t groups = 0;
    const int shift_form = range <= RFX_INDEX_SCOPE_LIMIT;
    if (n) {
        rfx_agg_t none;
        memset(&none, 0, sizeof(none));
        rfx_group_tables_t gt;
        memset(&gt, 0, sizeof(gt));
        if (rfx_hip_malloc(g_ctx, &store, (size_t)range * 8) != RFX_OK) return fail_hip("group tables");
        gt.kmin = kmin;
        gt.range = (int64_t)range;
        gt.nagg = 0;
        gt.d_first = (int64_t *)store;
        int ok = rfx_hip_group_tables_init(g_ctx, &none, &gt) == RFX_OK &&
                 rfx_hip_group_dense_accumulate(g_ctx, (const int64_t *)dk, NULL, 0, RFX_AND, &none, n, 0, &gt) == RFX_OK &&
                 rfx_hip_group_rank(g_ctx, &gt, n, &groups) == RFX_OK;
        ok = ok && rfx_hip_malloc(g_ctx, &dfirst, (size_t)(groups ? groups : 1) * 8) == RFX_OK &&
             rfx_hip_group_emit(g_ctx, &none, &gt, NULL, (int64_t *)dfirst, NULL) == RFX_OK;
        const int64_t nid = shift_form ? (int64_t)range : n;
        ok = ok && rfx_hip_malloc(g_ctx, &dids, (size_t)nid * 8) == RFX_OK &&
             (shift_form ? rfx_hip_group_slot_ids(g_ctx, &gt, (int64_t *)dids) : rfx_hip_group_ids_dense(g_ctx, (const int64_t *)dk, n, &gt, (int64_t *)dids)) == RFX_OK;
        if (ok) {
            gids = H.vector(RFX_TYPE_I64, nid);
            firsts = H.vector(RFX_TYPE_I64, groups);
            ok = rfx_hip_d2h(g_ctx, RFX_AS_RAW(gids), dids, (size_t)nid * 8) == RFX_OK && (groups == 0 || rfx_hip_d2h(g_ctx, RFX_AS_RAW(firsts), dfirst, (size_t)groups * 8) == RFX_OK);
        }
        if (store) rfx_hip_free(g_ctx, store);
        if (dfirst) rfx_hip_free(g_ctx, dfirst);
        if (dids) rfx_hip_free(g_ctx, dids);
        if (!ok) {
            if (gids) H.drop(gids);
            if (firsts) H.drop(firsts);
            return fail_hip("group index");
        }
    } else {
        gids = H.vector(RFX_TYPE_I64, 0);
        firsts = H.vector(RFX_TYPE_I64, 0);
    }
    res = H.vector(RFX_TYPE_LIST, 7);
    obj_p *ix = RFX_AS_LIST(res);
    ix[0] = H.i64(shift_form && n ? RFX_INDEX_TYPE_SHIFT : RFX_INDEX_TYPE_IDS);
    ix[1] = H.i64(groups);
    ix[2] = gids;
    ix[3] = H.i64(shift_form && n ? kmin : RFX_NULL_I64);
    ix[4] = shift_form && n ? H.clone(keys) : H.null_obj; /* NULL_OBJ is the host's static null, passed as index_group_build passes it */
    ix[5] = H.null_obj;
    ix[6] = firsts;
    return res;
}
rfx_obj_p rfx_group(rfx_obj_p keys) {
    op_begin();
    obj_p r = group_impl(keys);
    op_end();
    return r;
}

/* scalar aggregates of a vector or of a lazy MAPFILTER (val, ids) pair (core/filter.c:29-49, core/math.c:1874-1890) */
static obj_p fold_impl(int f, int kind, obj_p x) {
    rfx_host_bind();
    if (!x) return fail("aggregate: null argument");
    if (x->type == RFX_TYPE_MAPGROUP) return fold_mapgroup(f, kind, x);
    if (x->type == RFX_TYPE_MAPFILTER) {
        /* the lazy (val, ids) pair an FN_AGGR built-in receives (core/eval.c:723-728): gather on the device, fold there --
         * the filtered vector the reference would materialise (filter_collect) never exists on the host */
        obj_p val = RFX_AS_LIST(x)[0], ids = RFX_AS_LIST(x)[1];
        if (!(val->type > 0 && col_ctype(val) && val->type != RFX_TYPE_SYMBOL) || ids->type != RFX_TYPE_I64) {
            if (H.bound == 1 && H.f[f]) return HOST_CALL(((rfx_unary_f)H.f[f])(x));
            return fail("aggregate: only (i64/f64 vector, i64 ids) MAPFILTER pairs run on the MI355X path");
        }
        if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
        if (g_nshards > 1) { /* every shard gathers and folds the ids inside its rows; partials folded in shard order (rfx_exec_filter_aggr) */
            const void *dvs;
            const int64_t *dsel[RFX_MAX_SHARDS];
            int64_t nsel[RFX_MAX_SHARDS];
            rfx_agg_t as;
            rfx_value_t vs;
            if (resident(val, 0, &dvs) != RFX_OK) return fail_hip("column upload");
            int rc = sel_ids_sharded(ids, val->len, dsel, nsel);
            if (rc == 1) { /* ids that do not ascend through the shards' row ranges: not a filter's -- the host's own aggregate */
                qtmp_release();
                g_refused_sharded = 1;
                return refused1(f, x);
            }
            memset(&as, 0, sizeof(as));
            as.d_col = dvs;
            as.col_type = col_ctype(val);
            as.kind = kind;
            if (rc == RFX_OK) rc = fold_sharded(&as, val->len, dsel, nsel, &vs);
            qtmp_release();
            if (rc != RFX_OK) return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
            return value_atom(&vs);
        }
        const void *dv, *di;
        if (resident(val, 0, &dv) != RFX_OK || transient(ids, &di) != RFX_OK) return fail_hip("column upload");
        void *dg = NULL;
        rfx_agg_t a;
        memset(&a, 0, sizeof(a));
        a.col_type = col_ctype(val);
        a.kind = kind;
        rfx_value_t v;
        int ok = rfx_hip_malloc(g_ctx, &dg, (size_t)(ids->len ? ids->len : 1) * 8) == RFX_OK &&
                 rfx_hip_gather_checked(g_ctx, dv, val->len, col_ctype(val), (const int64_t *)di, ids->len, dg) == RFX_OK;
        a.d_col = dg;
        ok = ok && rfx_hip_filter_aggr_host(g_ctx, NULL, 0, RFX_AND, &a, 1, ids->len, &v, NULL) == RFX_OK;
        if (dg) rfx_hip_free(g_ctx, dg);
        if (!ok) return fail_hip("filter_aggr over a MAPFILTER");
        return value_atom(&v);
    }
    if (!(x->type > 0 && col_ctype(x) && x->type != RFX_TYPE_SYMBOL)) {
        if (H.bound == 1 && H.f[f]) return HOST_CALL(((rfx_unary_f)H.f[f])(x));
        return fail("aggregate: only i64/f64 vectors run on the MI355X path");
    }
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    const void *d;
    if (resident(x, 0, &d) != RFX_OK) return fail_hip("column upload");
    rfx_agg_t a;
    memset(&a, 0, sizeof(a));
    a.d_col = d;
    a.col_type = col_ctype(x);
    a.kind = kind;
    rfx_value_t v;
    if (g_nshards > 1) { /* the fold on every shard, the partials in shard order (unop_fold's two levels, core/math.c:2176-2231) */
        if (fold_sharded(&a, x->len, NULL, NULL, &v) != RFX_OK) return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
        return value_atom(&v);
    }
    if (rfx_hip_filter_aggr_host(g_ctx, NULL, 0, RFX_AND, &a, 1, x->len, &v, NULL) != RFX_OK) return fail_hip("filter_aggr");
    return value_atom(&v);
}
static obj_p fold_op(int f, int kind, obj_p x) {
    op_begin();
    obj_p r = fold_impl(f, kind, x);
    op_end();
    return r;
}
rfx_obj_p rfx_sum(rfx_obj_p x) { return fold_op(F_SUM, RFX_AGG_SUM, x); }
rfx_obj_p rfx_avg(rfx_obj_p x) { return fold_op(F_AVG, RFX_AGG_AVG, x); }
rfx_obj_p rfx_min(rfx_obj_p x) { return fold_op(F_MIN, RFX_AGG_MIN, x); }
rfx_obj_p rfx_max(rfx_obj_p x) { return fold_op(F_MAX, RFX_AGG_MAX, x); }
rfx_obj_p rfx_count(rfx_obj_p x) { return fold_op(F_COUNT, RFX_AGG_COUNT, x); }
rfx_obj_p rfx_first(rfx_obj_p x) { return fold_op(F_FIRST, RFX_AGG_FIRST, x); }

/* ------------------------------------------------------------------------------------------------ residency verbs */
static obj_p pin_impl(obj_p x, int pin);
static obj_p pin_op(obj_p x, int pin) {
    op_begin();
    obj_p r = pin_impl(x, pin);
    op_end();
    return r;
}
static obj_p pin_impl(obj_p x, int pin) {
    rfx_host_bind();
    if (!x) return fail("pin: null argument");
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    obj_p cols = (x->type == RFX_TYPE_TABLE) ? RFX_AS_LIST(x)[1] : NULL;
    int64_t n = cols ? cols->len : 1;
    if (cols && is_parted_table(x)) { /* a get-parted table: its columns are known to the cache by their LIST objects (parted_view) */
        if (g_nshards > 1) return H.clone(x); /* (selects over parted tables are the host's under RFX_SHARDS / RFX_DEVICES: nothing to keep resident) */
        obj_p view = pin ? parted_view(x) : NULL;
        int bad = pin && !view;
        for (int64_t i = 0; i < n && !bad; i++) {
            obj_p c = RFX_AS_LIST(cols)[i];
            if (pin) {
                obj_p pc = RFX_AS_LIST(RFX_AS_LIST(view)[1])[i];
                const void *d;
                /* only 8-byte proxies: proxy_upload / proxy_sum address partitions as 8-byte cells (a B8 proxy would overrun its 1-byte-per-row
                 * device block and read past the mmapped partition files) */
                if (col_ctype(pc) && pc->type > 0 && resident(pc, 1, &d) != RFX_OK) bad = 1;
            } else {
                for (int j = 0; j < g_nres; j++)
                    if (g_res[j].host == (const void *)c || g_res[j].host == RFX_AS_RAW(c)) { res_free(j); break; }
            }
        }
        parted_view_release();
        return bad ? fail_hip("pin") : H.clone(x);
    }
    for (int64_t i = 0; i < n; i++) {
        obj_p c = cols ? RFX_AS_LIST(cols)[i] : x;
        if (!(c->type > 0 && (col_ctype(c) || c->type == RFX_TYPE_B8))) continue;
        if (pin) {
            const void *d;
            if (resident(c, 1, &d) != RFX_OK) return fail_hip("pin");
        } else {
            for (int j = 0; j < g_nres; j++)
                if (g_res[j].host == RFX_AS_RAW(c)) { res_free(j); break; }
        }
    }
    return H.clone(x);
}
rfx_obj_p rfx_pin(rfx_obj_p x) { return pin_op(x, 1); }
rfx_obj_p rfx_unpin(rfx_obj_p x) { return pin_op(x, 0); }
/* (rfx_invalidate x): the host is about to write (or has just written) into vector x / the columns of table x in place: every
 * cached device copy that overlaps their payload is dropped, pinned or not.  The hook a host patch calls from `set` on a column and
 * from the rc == 1 in-place arithmetic (core/math.c:2248, :2310), see INTEGRATION.md. */
rfx_obj_p rfx_invalidate(rfx_obj_p x) {
    rfx_host_bind();
    if (!x) return fail("invalidate: null argument");
    op_begin();
    if (x->type == RFX_TYPE_TABLE) {
        obj_p cols = RFX_AS_LIST(x)[1];
        for (int64_t i = 0; i < cols->len; i++) invalidate_payload(RFX_AS_LIST(cols)[i]);
    } else invalidate_payload(x);
    op_end();
    return H.clone(x);
}
/* (rfx_stats 0): counters since load as an I64 vector -- [selects answered on the GPU, selects handed back to the host's
 * ray_select, joins on the GPU, joins delegated, host-to-device uploads, cache hits, stale cache entries refreshed, operator
 * calls].  What a drop-in test asserts to know that an answer really came from the device. */
rfx_obj_p rfx_stats(rfx_obj_p x) {
    (void)x;
    rfx_host_bind();
    obj_p out = H.vector(RFX_TYPE_I64, 12);
    for (int i = 0; i < 10; i++) RFX_AS_I64(out)[i] = g_stat[i];
    if (g_x) { /* scopes sampled / sampled scopes retried exactly: the planner's counters */
        RFX_AS_I64(out)[ST_SCOPE_SAMPLED] = rfx_exec_stat(g_x, RFX_XSTAT_SCOPE_SAMPLED);
        RFX_AS_I64(out)[ST_SCOPE_RETRIED] = rfx_exec_stat(g_x, RFX_XSTAT_SCOPE_RETRIED);
    }
    RFX_AS_I64(out)[10] = g_ctx ? rfx_hip_ctx_stat(g_ctx, RFX_STAT_MASK_PASSES) : 0;
    RFX_AS_I64(out)[11] = g_sd_hits; /* unpinned columns proven current by soft-dirty page bits (0: the kernel has no such tracking) */
    return out;
}
