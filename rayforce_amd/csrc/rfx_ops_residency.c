/* rfx_ops_residency.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * device contexts / shards of the operator layer, the residency cache (checksums, soft-dirty pages, per-shard copies), parted-table views. */
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
__attribute__((unused)) static int ensure_ctx1(void) {
    const int rc = ensure_ctx();
    g_refused_sharded = rc == RFX_OK && g_nshards > 1;
    return g_refused_sharded ? RFX_ELIMIT : rc;
}

/* Residency cache: host vector -> device copy.
 *
 * A cached copy is used again only if it is PROVEN current:
 *   - default (round 6), BY OWNERSHIP: the entry holds a reference to the host vector (`owner` = clone_obj(col), rayforce.syms:25-26).  The
 *     reference never writes a vector in place unless it is the only owner -- cow_obj core/rayforce.c:3003-3026 (rc == 1 or copy), the
 *     in-place arithmetic core/math.c:2248,2310 (rc_obj(x) == 1), every writer of core/update.c through cow_obj -- and never frees what is
 *     referenced, so while the cache holds its reference the cells cannot change and the address cannot be recycled: validation is ONE
 *     POINTER COMPARE (the same object), pinned or not.  An entry whose count has fallen to 1 (only the cache is left: nobody can name the
 *     vector any more) is released at the next operator call (op_begin), LRU eviction / rfx_unpin / rfx_invalidate / rfx_cache_clear drop
 *     the reference with the copy.  References are taken and dropped through the HOST's clone_obj / drop_obj on the calling thread (they
 *     switch to atomics under pool_run themselves, VM->rc_sync).  Cost to the host: an owner that would have written in place copies
 *     instead (its column is rc >= 2 while cached) -- the copy the device needs anyway, since new cells mean a new upload;
 *   - RFX_VALIDATE=checksum / rfx_ops_set_validation(1), for a host that writes vectors in place WITHOUT looking at the count (raw numpy
 *     views over the standalone host's payloads in tests/test_ops_gpu.py): keyed by (payload address, length, type), the FULL payload is checksummed on every use (threaded multiply-xor over every 8-byte word, position dependent) and
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
    obj_p owner;       /* validation by ownership: OUR reference to the host vector (a parted column: to its LIST); NULL in checksum mode */
    int amax_ok;       /* reproducible sums (resident_fixed): 1 = amax is max |x| over this f64 copy, 2 = the copy holds a NaN / an infinity; 0 = not looked at */
    double amax;
    int fix_k, fix_m;  /* a DERIVED entry (type code + 256, + 512 for the second limb): this column's fixed-point image llrint(x * 2^fix_k) as i64 (the second limb:
                        * what that rounded away, scaled by 2^fix_m more), owned like the copy it was made from */
} resident_t;
static resident_t *g_res;
static int g_nres, g_capres;
static uint64_t g_tick, g_epoch = 1;
static size_t g_res_bytes;
static int64_t g_stat[10]; /* see rfx_stats */
static int64_t g_sd_hits;  /* uses of an unpinned cached column proven current by its pages' soft-dirty bits instead of the checksum */
static int64_t g_sum_validations; /* uses of a cached column that cost a full-payload checksum (0 under validation by ownership) */
static int64_t g_own_hits;        /* uses of a cached column proven current by being the very object the cache holds a reference to */
static int64_t g_own_released;    /* entries released because the cache's reference was the last one */
static int64_t g_fix_built, g_fix_hits; /* fixed-point images made / found again (reproducible sums) */
static int g_validate = -1;       /* 0 ownership (default), 1 checksum: RFX_VALIDATE=checksum, rfx_ops_set_validation */
static int validate_mode(void) {
    if (g_validate < 0) {
        const char *e = getenv("RFX_VALIDATE");
        g_validate = (e && (strcmp(e, "checksum") == 0 || strcmp(e, "1") == 0)) ? 1 : 0;
    }
    return g_validate;
}
static inline uint32_t obj_rc(obj_p o) { return __atomic_load_n(&o->rc, __ATOMIC_RELAXED); } /* (rc_obj, core/rayforce.c:3028-3032) */
enum { ST_SELECT_GPU, ST_SELECT_DELEGATED, ST_JOIN_GPU, ST_JOIN_DELEGATED, ST_UPLOADS, ST_CACHE_HITS, ST_CACHE_STALE, ST_OPS, ST_SCOPE_SAMPLED, ST_SCOPE_RETRIED };

typedef struct {
    const unsigned char *p;
    size_t bytes;
    uint64_t h;
} sum_job_t;
#define SUM_ROTL(x, r) (((x) << (r)) | ((x) >> (64 - (r))))
static uint64_t sum_range(const unsigned char *p, size_t bytes) {
    /* four independent rotate-multiply lanes (the multiply's latency is the limit of a single chain), folded in a fixed order.  The ROTATE is
     * what carries a word's high bits into the low half: a bare (h ^ w) * K only ever moves a difference UPWARD, so cells that differ in bit 63
     * alone -- NULL_I64 against 0, an f64 against its negative -- were remembered by their parity per lane, and a column of nulls over a
     * recycled column of zeros of the same length passed for current (rounds 2-5, found by tools/fuzz_null_tuples.py) */
    const uint64_t K = 0x9E3779B97F4A7C15ULL;
    uint64_t h0 = 0x243F6A8885A308D3ULL, h1 = 0x13198A2E03707344ULL, h2 = 0xA4093822299F31D0ULL, h3 = 0x082EFA98EC4E6C89ULL;
    size_t nw = bytes / 8, i = 0;
    const uint64_t *w = (const uint64_t *)p; /* payloads are 8-byte aligned (obj + 16, 32-byte aligned blocks) */
    if (((uintptr_t)p & 7) == 0) {
        for (; i + 4 <= nw; i += 4) {
            h0 = SUM_ROTL(h0 ^ w[i], 31) * K;
            h1 = SUM_ROTL(h1 ^ w[i + 1], 31) * K;
            h2 = SUM_ROTL(h2 ^ w[i + 2], 31) * K;
            h3 = SUM_ROTL(h3 ^ w[i + 3], 31) * K;
        }
        for (; i < nw; i++) h0 = SUM_ROTL(h0 ^ w[i], 31) * K;
    } else i = 0, nw = 0;
    uint64_t h = (SUM_ROTL(h0 ^ (h1 >> 29), 31) * K) ^ (SUM_ROTL(h2 ^ (h3 >> 31) ^ (h1 << 35) ^ (h3 << 33), 27) * K);
    for (size_t b = nw * 8; b < bytes; b++) h = SUM_ROTL(h ^ p[b], 31) * K;
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
        h = SUM_ROTL(h ^ job[i].h, 31) * 0x9E3779B97F4A7C15ULL; /* chunk order matters: a value moved between chunks changes the sum */
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
    obj_p owner = g_res[i].owner;
    g_res[i] = g_res[--g_nres];
    if (owner) H.drop(owner); /* (last: when ours was the last reference the host frees the vector here) */
}
/* entries nobody but the cache refers to any more: the host dropped the vector, no operator call can name it again */
static void release_unowned(void) {
    for (int i = 0; i < g_nres;) {
        obj_p o = g_res[i].owner;
        if (!o) { i++; continue; }
        uint32_t mine = 0; /* (one vector may have two copies -- row ranges and a join's whole copy per shard -- each with its own reference) */
        for (int j = 0; j < g_nres; j++) mine += g_res[j].owner == o;
        if (obj_rc(o) > mine) { i++; continue; }
        for (int j = g_nres - 1; j >= 0; j--)
            if (g_res[j].owner == o) { res_free(j); g_own_released++; }
        i = 0; /* (the array was compacted: start over -- releases are rare) */
    }
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
    if (H.bound == 2) rfx_host_trim(); /* (the standalone host's recycled result blocks too) */
    op_end();
}
int64_t rfx_cache_bytes(void) { return (int64_t)g_res_bytes; }
/* how cached copies are proven current: 0 by ownership (default), 1 by checksum (hosts that write payloads in place whatever the reference count).
 * Entries validated the other way are dropped. */
int rfx_ops_set_validation(int mode) {
    if (mode != 0 && mode != 1) return RFX_EINVAL;
    rfx_host_bind();
    op_begin();
    if (validate_mode() != mode) {
        while (g_nres) res_free(g_nres - 1);
        g_validate = mode;
    }
    op_end();
    return RFX_OK;
}

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
        if (g_nres) release_unowned();
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
typedef struct {
    int type, esz;
    void *const *devs;
    const char *host;
    int64_t len;
    int whole; /* every shard takes the WHOLE payload (a join's build side) instead of its row range */
} upload_job_t;
static int upload_shard(void *arg, int s) {
    const upload_job_t *u = (const upload_job_t *)arg;
    int64_t r0 = 0, n = u->len;
    if (!u->whole) rfx_exec_split(u->len, g_nshards, s, &r0, &n);
    return n > 0 ? payload_upload_one(g_ctxs[s], u->type, u->devs[s], u->host + (size_t)r0 * u->esz, n) : RFX_OK;
}
/* Round 6: every shard's row range at once, each on its shard's own thread (rfx_exec_run), through its own context's staging set and stream: on an
 * 8-device node eight copy engines and eight PCIe links move the column together (rounds 4-5 walked the shards in a loop around a blocking copy: 24 GB of
 * c3w columns crossed one link at a time).  The staging copies of all of them share the library's persistent workers (rfx_io.hip). */
static int payload_upload(int type, void *const *devs, const void *host, int64_t len, int whole) {
    upload_job_t u = {type, type == RFX_TYPE_B8 ? 1 : (IS_I32_FAMILY(type) ? 4 : 8), devs, (const char *)host, len, whole};
    if (g_nshards == 1) return upload_shard(&u, 0);
    const int rc = rfx_exec_run(g_x, upload_shard, &u);
    rfx_hip_ctx_bind_thread(g_ctx);
    return rc;
}
static int shards_alloc(void **devs, int64_t len, size_t desz, int whole) {
    int rc = RFX_OK;
    for (int s = 0; s < RFX_MAX_SHARDS; s++) devs[s] = NULL;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        int64_t n = len;
        if (!whole) rfx_exec_split(len, g_nshards, s, NULL, &n);
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
/* whole = 1 (round 6, several shards only): the column WHOLE on every shard's device -- the BUILD side of a join over sharded tables (every shard
 * probes its own left rows against all of the right table: a broadcast join) -- cached beside the row-range copies as an entry of its own
 * (type code + 128); *devs_out receives the copies' addresses, the planner's column table (g_qcols) does not learn them. */
static void res_make_room(size_t dbytes) {
    while (g_nres && g_res_bytes + dbytes > cache_budget()) {
        int victim = -1; /* least recently used, not pinned, not in use by the call in flight */
        for (int i = 0; i < g_nres; i++)
            if (!g_res[i].pinned && g_res[i].epoch != g_epoch && (victim < 0 || g_res[i].tick < g_res[victim].tick)) victim = i;
        if (victim < 0) break; /* everything left is pinned or in use: go over budget rather than free what the call reads */
        res_free(victim);
    }
}
static void res_append(const resident_t *e) {
    if (g_nres == g_capres) {
        g_capres = g_capres ? g_capres * 2 : 32;
        g_res = (resident_t *)realloc(g_res, sizeof(resident_t) * (size_t)g_capres);
    }
    g_res[g_nres++] = *e;
    g_res_bytes += e->dbytes;
}
static int resident_ex(obj_p col, int pin, int whole, const void **dev, void **devs_out) {
    if (whole && (col->mmod == RFX_MMOD_DEVICE || g_nshards <= 1)) return RFX_ELIMIT; /* (device handles hold row ranges: nothing to replicate from) */
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
    if (px && (g_nshards > 1 || whole)) return RFX_ELIMIT; /* (parted views run on one shard: the caller hands such tables to the host) */
#define RES_DONE(e) do { *dev = (e)->dev; if (devs_out) for (int s_ = 0; s_ < g_nshards; s_++) devs_out[s_] = (e)->devs[s_]; return whole ? RFX_OK : qcol_add((e)->devs); } while (0)
    const int narrow = !px && IS_I32_FAMILY(col->type);
    const int esz = (col->type == RFX_TYPE_B8) ? 1 : (narrow ? 4 : 8);
    const void *host = px ? (const void *)px->src : RFX_AS_RAW(col); /* a parted column is known by its LIST object */
    const size_t bytes = (size_t)col->len * esz;          /* of the HOST payload: what is validated */
    const size_t dbytes = (size_t)col->len * (narrow ? 8 : esz) * (size_t)(whole ? g_nshards : 1); /* of the device copy (copies): what the budget counts */
    const int ktype = (px ? 64 + col->type : col->type) + (whole ? 128 : 0);
    int have_sum = 0;
    uint64_t sum = 0;
    const int own = validate_mode() == 0;
    obj_p keyobj = px ? px->src : col; /* the host object the cells belong to (a parted column: its LIST of partition vectors) */
    for (int i = 0; i < g_nres; i++) {
        if (g_res[i].owner) { /* by ownership: the very object we hold a reference to -- nothing could have written its cells or reused its address */
            if (g_res[i].owner != keyobj || g_res[i].type != ktype) continue; /* (another object -- or the same one's OTHER copy: row ranges and a whole copy per shard live side by side) */
            if (g_res[i].len != col->len) { res_free(i); break; } /* (cannot happen under the host's rule: take it as new) */
            g_own_hits++;
            g_res[i].tick = ++g_tick;
            g_res[i].epoch = g_epoch;
            g_res[i].pinned |= pin;
            g_stat[ST_CACHE_HITS]++;
            RES_DONE(&g_res[i]);
        }
        if (g_res[i].host == host && g_res[i].len == col->len && g_res[i].type == ktype) {
            int track = 0;
            if (!g_res[i].pinned) { /* unpinned: prove the copy current */
                if (!px && g_res[i].tracked && sd_entry_clean(&g_res[i])) { /* by its pages (soft-dirty bits): nothing wrote there */
                    g_sd_hits++;
                    g_res[i].tick = ++g_tick;
                    g_res[i].epoch = g_epoch;
                    g_res[i].pinned |= pin;
                    g_stat[ST_CACHE_HITS]++;
                    RES_DONE(&g_res[i]);
                }
                g_res[i].tracked = 0;
                /* by its checksum -- taken AFTER the pages were clean-marked, so that it can vouch for them from now on */
                track = !px && !g_res[i].sd_never && g_res[i].stable >= SD_STABLE_USES && sd_usable(host, bytes) && sd_call_clear() == 0;
                sum = px ? proxy_sum(px) : payload_sum(host, bytes);
                g_sum_validations++;
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
                RES_DONE(&g_res[i]);
            }
            /* stale: the payload changed under the same address -- refresh the device copy in place */
            g_stat[ST_CACHE_STALE]++;
            int rc = px ? proxy_upload(px, g_res[i].dev) : payload_upload(col->type, g_res[i].devs, host, col->len, whole);
            if (rc != RFX_OK) { res_free(i); return rc; }
            g_stat[ST_UPLOADS]++;
            g_res[i].scope_ok = 0; /* (new cells: the scope remembered for the old ones is gone) */
            g_res[i].amax_ok = 0;
            g_res[i].sum = sum;
            g_res[i].tick = ++g_tick;
            g_res[i].epoch = g_epoch;
            g_res[i].pinned |= pin;
            RES_DONE(&g_res[i]);
        }
    }
    res_make_room(dbytes);
    void *devs[RFX_MAX_SHARDS];
    int rc = shards_alloc(devs, col->len, narrow ? 8 : (size_t)esz, whole);
    if (rc != RFX_OK) return rc;
    /* the checksum, THEN the copy: a host write racing with this call is either in both, or in the copy only and costs one refresh at
     * the next use -- never a device copy older than what vouches for it */
    if (!have_sum && !own) sum = px ? proxy_sum(px) : payload_sum(host, bytes);
    rc = px ? proxy_upload(px, devs[0]) : payload_upload(col->type, devs, host, col->len, whole); /* heap vector or mmapped column file alike: staged through pinned buffers */
    if (rc != RFX_OK) {
        for (int s = 0; s < g_nshards; s++) { if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]); rfx_hip_free(g_ctxs[s], devs[s]); }
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
        return rc;
    }
    g_stat[ST_UPLOADS]++;
    resident_t e;
    memset(&e, 0, sizeof(e));
    e.host = host; e.len = col->len; e.type = ktype; e.sum = sum; e.dev = devs[0]; e.bytes = bytes; e.pinned = pin; e.tick = ++g_tick; e.epoch = g_epoch; e.dbytes = dbytes;
    e.owner = own ? H.clone(keyobj) : NULL; /* from here on the host copies before it writes, and cannot free */
    for (int s = 0; s < g_nshards; s++) e.devs[s] = devs[s];
    res_append(&e); /* (page tracking starts once the column has proven stable) */
    RES_DONE(&g_res[g_nres - 1]);
#undef RES_DONE
}
static int resident(obj_p col, int pin, const void **dev) { return resident_ex(col, pin, 0, dev, NULL); }
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
