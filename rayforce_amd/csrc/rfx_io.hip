// rfx_io.hip -- getting host / on-disk columns into HBM fast: the caller side of the hot path.
//
// Reference: a column file is a 16-byte header {mmod = 0xfd, order, type, attrs, rc, len:i64} followed by the raw
// little-endian payload (core/binary.c:263-311 writes it, core/unary.c:48-136 mmaps it); a splayed table is a directory of
// such files plus `.d`, the serialised symbol vector of the column names (core/io.c:1194-1364).  The reference never copies:
// `get` maps the file and the evaluator reads the mapping.  A GPU has to move the bytes once; what matters is that this one
// move runs at PCIe speed rather than at the speed of a pageable memcpy:
//
//   rfx_hip_h2d_pipelined   source = ANY host memory (heap vector, mmapped file).  The range is cut into chunks; worker
//                           threads copy chunk i+1 into one of four pinned staging buffers (touching mmapped pages, i.e. doing
//                           the file I/O) while the DMA engine moves chunk i from another -- staging copy and transfer overlap,
//                           and the DMA only ever sees pinned memory.
//   rfx_hip_column_file_load  mmap + the above.  rfx_column_file_stat reads the header only (no device needed).
//
// rfx_ops.c's residency cache uploads through rfx_hip_h2d_pipelined, so a reference process that `get`s a 1e9-row column
// pays the transfer once at link speed and keeps the column in HBM afterwards.
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "rfx_common.hpp"

#define IO_CHUNK ((size_t)32 << 20) /* staging buffer size */
#define IO_NBUF 4
#define IO_PIECE ((size_t)2 << 20) /* what one worker copies at a time */
#define IO_MAX_WORKERS 64
#define IO_QUEUE 4096

// The staging copies run on a PERSISTENT set of worker threads shared by every context of the process (round 6; rounds 2-5 created and joined 16
// threads per 32 MB chunk: 4 096 pthread_create per 8 GB column, ~0.4 ms of every chunk's ~0.6 ms).  A copy is cut into 2 MB pieces queued to the
// workers; the submitter gets a ticket (a counter of pieces left) and waits for it only when it needs the bytes -- so a transfer keeps TWO chunks of
// read-ahead in flight behind the one the DMA engine is moving, and the uploads of several shards / devices (one host thread each, rfx_exec_run)
// share the workers.  Workers: a quarter of the online CPUs, 16..64 (one device at PCIe speed needs ~16; eight devices at once are bound by host
// memory bandwidth, not by threads).  They live as long as the process (the library is never unloaded by the reference: core/dynlib.c keeps handles).
struct IoJob {
    char *dst;
    const char *src; // host address -- or, with fd >= 0, the byte OFFSET in that file
    size_t bytes;
    int fd;
    int *left; // pieces of this copy still to do (guarded by g_io.mu)
};
static struct {
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    IoJob q[IO_QUEUE];
    unsigned head, tail; // jobs [head, tail) are queued
    int nworkers, state; // state: 0 not started, 1 running, -1 no threads to be had (copies run on the caller)
    pthread_t th[IO_MAX_WORKERS];
} g_io = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {}, 0, 0, 0, 0, {}};
static void *io_worker(void *) {
    pthread_mutex_lock(&g_io.mu);
    for (;;) {
        while (g_io.head == g_io.tail) pthread_cond_wait(&g_io.cv_work, &g_io.mu);
        const IoJob j = g_io.q[g_io.head++ % IO_QUEUE];
        pthread_mutex_unlock(&g_io.mu);
        if (j.fd >= 0) { // a column FILE (rfx_hip_column_file_load): read straight into the pinned staging buffer -- no mapping, no page faults
            size_t got = 0;
            while (got < j.bytes) {
                const ssize_t r = pread(j.fd, j.dst + got, j.bytes - got, (off_t)((size_t)j.src + got));
                if (r <= 0) { // (short file / error: the tail stays zero; the loader checked the size before)
                    memset(j.dst + got, 0, j.bytes - got);
                    break;
                }
                got += (size_t)r;
            }
        } else
        memcpy(j.dst, j.src, j.bytes); // (touching an mmapped source's pages -- the file I/O -- happens here, in parallel)
        pthread_mutex_lock(&g_io.mu);
        if (--*j.left == 0) pthread_cond_broadcast(&g_io.cv_done);
    }
    return NULL;
}
static void io_pool_start_locked(void) {
    long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    int want = (int)(cpus / 4);
    if (want < 16) want = cpus >= 16 ? 16 : (cpus > 1 ? (int)cpus : 1);
    if (want > IO_MAX_WORKERS) want = IO_MAX_WORKERS;
    if (const char *e = getenv("RFX_IO_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= IO_MAX_WORKERS) want = v;
    }
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
    for (int i = 0; i < want; i++) {
        if (pthread_create(&g_io.th[g_io.nworkers], &at, io_worker, NULL) != 0) break;
        g_io.nworkers++;
    }
    pthread_attr_destroy(&at);
    g_io.state = g_io.nworkers > 0 ? 1 : -1;
}
// queue dst <- src for the workers; *left counts the pieces (the caller keeps it alive until io_wait returns)
static void io_read_here(char *dst, const char *src, size_t bytes, int fd) {
    if (fd < 0) {
        memcpy(dst, src, bytes);
        return;
    }
    size_t got = 0;
    while (got < bytes) {
        const ssize_t r = pread(fd, dst + got, bytes - got, (off_t)((size_t)src + got));
        if (r <= 0) {
            memset(dst + got, 0, bytes - got);
            return;
        }
        got += (size_t)r;
    }
}
static void io_submit(char *dst, const char *src, size_t bytes, int *left, int fd = -1) {
    *left = 0;
    if (!bytes) return;
    pthread_mutex_lock(&g_io.mu);
    if (g_io.state == 0) io_pool_start_locked();
    if (g_io.state < 0 || bytes < IO_PIECE) { // nothing to hand it to / not worth a hand-over
        pthread_mutex_unlock(&g_io.mu);
        io_read_here(dst, src, bytes, fd);
        return;
    }
    size_t off = 0;
    int queued = 0;
    while (off < bytes && g_io.tail - g_io.head < IO_QUEUE) {
        const size_t n = bytes - off < IO_PIECE + IO_PIECE / 2 ? bytes - off : IO_PIECE;
        g_io.q[g_io.tail++ % IO_QUEUE] = IoJob{dst + off, src + off, n, fd, left};
        off += n;
        queued++;
    }
    *left = queued;
    if (queued) pthread_cond_broadcast(&g_io.cv_work);
    pthread_mutex_unlock(&g_io.mu);
    if (off < bytes) io_read_here(dst + off, src + off, bytes - off, fd); // (the queue was full: the rest here)
}
static void io_wait(int *left) {
    pthread_mutex_lock(&g_io.mu);
    while (*left > 0) pthread_cond_wait(&g_io.cv_done, &g_io.mu);
    pthread_mutex_unlock(&g_io.mu);
}
static void parallel_copy(char *dst, const char *src, size_t bytes) {
    int left;
    io_submit(dst, src, bytes, &left);
    io_wait(&left);
}

static int io_stage_ready(rfx_ctx *c);
void rfx_plane_invalidate(rfx_ctx *c); // rfx_group_plane.hip
static int h2d_pipelined_from(rfx_ctx_t *c, void *d_dst, const void *src, size_t bytes, int fd);
extern "C" int rfx_hip_h2d_pipelined(rfx_ctx_t *c, void *d_dst, const void *src, size_t bytes) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!bytes) return RFX_OK;
    RFX_REQUIRE(d_dst && src, RFX_EINVAL, "NULL argument");
    if (bytes < IO_CHUNK) return rfx_hip_h2d(c, d_dst, src, bytes);
    return h2d_pipelined_from(c, d_dst, src, bytes, -1);
}
// src: a host address, or (fd >= 0) the byte offset in the file the staging workers pread from
static int h2d_pipelined_from(rfx_ctx_t *c, void *d_dst, const void *src, size_t bytes, int fd) {
    c->ck_valid = 0; // as rfx_hip_h2d: partitions left by a scope pass do not survive an upload
    c->pc_valid = 0;
    rfx_plane_invalidate(c);
    {
        const int src_rc = io_stage_ready(c);
        if (src_rc != RFX_OK) return src_rc;
    }
    // chunk k is staged by the workers while chunks k - 1, k - 2 ... are on the wire: up to IO_AHEAD staging copies are queued beyond the one
    // the loop is waiting for, each into a buffer whose previous transfer has completed
    enum { IO_AHEAD = 2 };
    const size_t nchunks = (bytes + IO_CHUNK - 1) / IO_CHUNK;
    bool used[IO_NBUF] = {false, false, false, false};
    int left[IO_NBUF] = {0, 0, 0, 0};
    size_t staged = 0; // chunks whose staging copy has been queued
    hipError_t err = hipSuccess;
    for (size_t k = 0; k < nchunks && err == hipSuccess; k++) {
        for (; staged < nchunks && staged <= k + IO_AHEAD && err == hipSuccess; staged++) {
            const int b = (int)(staged % IO_NBUF);
            if (used[b]) err = hipEventSynchronize(c->io_done[b]); // its previous transfer has left the buffer
            if (err != hipSuccess) break;
            const size_t off = staged * IO_CHUNK, n = (bytes - off < IO_CHUNK) ? bytes - off : IO_CHUNK;
            io_submit((char *)c->io_stage[b], (const char *)src + off, n, &left[b], fd);
        }
        if (err != hipSuccess) break;
        const int b = (int)(k % IO_NBUF);
        const size_t off = k * IO_CHUNK, n = (bytes - off < IO_CHUNK) ? bytes - off : IO_CHUNK;
        io_wait(&left[b]);
        err = hipMemcpyAsync((char *)d_dst + off, c->io_stage[b], n, hipMemcpyHostToDevice, c->stream);
        if (err == hipSuccess) err = hipEventRecord(c->io_done[b], c->stream);
        used[b] = true;
    }
    for (int b = 0; b < IO_NBUF; b++) io_wait(&left[b]); // (an error above: no worker may still write into a counter on this stack)
    RFX_HIP_CHECK(err);
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RFX_OK;
}

static int io_stage_ready(rfx_ctx *c) {
    if (c->io_stage[IO_NBUF - 1]) return RFX_OK; // all buffers and events exist, or none is published (a half-built set is torn down)
    void *st[IO_NBUF] = {NULL, NULL, NULL, NULL};
    hipEvent_t ev[IO_NBUF];
    bool have_ev[IO_NBUF] = {false, false, false, false};
    hipError_t e = hipSuccess;
    for (int i = 0; i < IO_NBUF && e == hipSuccess; i++) {
        e = hipHostMalloc(&st[i], IO_CHUNK, hipHostMallocDefault);
        if (e == hipSuccess) {
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            have_ev[i] = e == hipSuccess;
        }
    }
    if (e != hipSuccess) {
        for (int i = 0; i < IO_NBUF; i++) {
            if (have_ev[i]) (void)hipEventDestroy(ev[i]);
            if (st[i]) (void)hipHostFree(st[i]);
        }
        RFX_HIP_CHECK(e);
    }
    for (int i = 0; i < IO_NBUF; i++) {
        c->io_done[i] = ev[i];
        c->io_stage[i] = st[i];
    }
    return RFX_OK;
}

// The other direction, for LARGE results (the 1e8-group row-hash query returns 6.4 GB of host columns): a plain copy into a freshly
// allocated vector takes every page fault of the destination one after the other inside the driver's pinning path (7 GB/s measured:
// 908 ms for that result).  Here the DMA engine fills pinned staging buffers, up to four chunks ahead, and the staging workers write
// each chunk into the destination -- the first touch of its pages is theirs, in parallel -- while the next chunks are in flight.
extern "C" int rfx_hip_d2h_pipelined(rfx_ctx_t *c, void *dst, const void *d_src, size_t bytes) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (!bytes) return RFX_OK;
    RFX_REQUIRE(dst && d_src, RFX_EINVAL, "NULL argument");
    if (bytes < 2 * IO_CHUNK) return rfx_hip_d2h(c, dst, d_src, bytes);
    int rc = io_stage_ready(c);
    if (rc != RFX_OK) return rc;
    const size_t nchunks = (bytes + IO_CHUNK - 1) / IO_CHUNK;
    size_t issued = 0;
    for (size_t k = 0; k < nchunks; k++) {
        for (; issued < nchunks && issued < k + IO_NBUF; issued++) { // keep up to IO_NBUF transfers in flight
            const size_t off = issued * IO_CHUNK, n = (bytes - off < IO_CHUNK) ? bytes - off : IO_CHUNK;
            const int b = (int)(issued % IO_NBUF);
            RFX_HIP_CHECK(hipMemcpyAsync(c->io_stage[b], (const char *)d_src + off, n, hipMemcpyDeviceToHost, c->stream));
            RFX_HIP_CHECK(hipEventRecord(c->io_done[b], c->stream));
        }
        const size_t off = k * IO_CHUNK, n = (bytes - off < IO_CHUNK) ? bytes - off : IO_CHUNK;
        const int b = (int)(k % IO_NBUF);
        RFX_HIP_CHECK(hipEventSynchronize(c->io_done[b]));
        parallel_copy((char *)dst + off, (const char *)c->io_stage[b], n);
    }
    return RFX_OK;
}

void rfx_io_release(rfx_ctx *c) {
    for (int i = 0; i < IO_NBUF; i++) {
        if (c->io_stage[i]) {
            (void)hipHostFree(c->io_stage[i]);
            (void)hipEventDestroy(c->io_done[i]);
            c->io_stage[i] = NULL;
        }
    }
}

// ---- column files (core/binary.c:263-311) ----
struct ColHeader {
    uint8_t mmod, order;
    int8_t type;
    uint8_t attrs;
    uint32_t rc;
    int64_t len;
};
static_assert(sizeof(ColHeader) == 16, "column file header");

static int elem_size(int type) {
    switch (type) {
        case 5:  /* I64 */
        case 6:  /* SYMBOL (ids) */
        case 9:  /* TIMESTAMP */
        case 10: /* F64 */
            return 8;
        default: return 0;
    }
}

extern "C" int rfx_column_file_stat(const char *path, int32_t *type, int64_t *len) {
    RFX_REQUIRE(path && type && len, RFX_EINVAL, "NULL argument");
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        rfx_set_error("rfx_column_file_stat: cannot open %s", path);
        return RFX_EINVAL;
    }
    ColHeader h;
    struct stat st;
    const bool ok = read(fd, &h, sizeof(h)) == (ssize_t)sizeof(h) && fstat(fd, &st) == 0;
    close(fd);
    if (!ok || h.mmod != 0xfd) {
        rfx_set_error("rfx_column_file_stat: %s is not a RayforceDB column file (16-byte header with mmod 0xfd expected)", path);
        return RFX_EINVAL;
    }
    const int es = elem_size(h.type);
    if (!es) {
        rfx_set_error("rfx_column_file_stat: %s holds type %d; only 8-byte columns (i64 / symbol ids / timestamp / f64) are on this path", path, (int)h.type);
        return RFX_EINVAL;
    }
    if (h.len < 0 || st.st_size < 16 || h.len > ((int64_t)st.st_size - 16) / es) { // (no 16 + len * es: a corrupt length must not overflow the test)
        rfx_set_error("rfx_column_file_stat: %s is truncated (%lld rows declared)", path, (long long)h.len);
        return RFX_EINVAL;
    }
    *type = h.type;
    *len = h.len;
    return RFX_OK;
}

extern "C" int rfx_hip_column_file_load(rfx_ctx_t *c, const char *path, void *d_dst, int64_t nrows) {
    RFX_REQUIRE(c && path, RFX_EINVAL, "NULL argument");
    int32_t type;
    int64_t len;
    int rc = rfx_column_file_stat(path, &type, &len);
    if (rc != RFX_OK) return rc;
    RFX_REQUIRE(nrows == len, RFX_EINVAL, "nrows does not match the file's length");
    if (len == 0) return RFX_OK;
    RFX_REQUIRE(d_dst != NULL, RFX_EINVAL, "d_dst is NULL");
    int fd = open(path, O_RDONLY);
    RFX_REQUIRE(fd >= 0, RFX_EINVAL, "cannot open the column file");
    // Round 6: the file is READ (pread by the staging workers, straight into the pinned buffers), not mapped: an mmapped source reaches 33-37 GB/s -- its
    // pages enter the process one minor fault (16 pages with fault-around) at a time under the copy; MADV_POPULATE_READ per piece was measured and does not
    // help: 33.9 against 36.6 GB/s, profiles/r06_h2d.txt -- where a heap source reaches 55.  (Columns the HOST has mapped -- the reference's `get` -- still
    // arrive as addresses: rfx_hip_h2d_pipelined from the mapping.)
    const size_t bytes = (size_t)len * 8;
    if (bytes < IO_CHUNK) { // small: one read into the staging area of rfx_hip_h2d
        void *tmp = malloc(bytes);
        if (!tmp) {
            close(fd);
            return RFX_ENOMEM;
        }
        io_read_here((char *)tmp, (const char *)(size_t)16, bytes, fd);
        rc = rfx_hip_h2d(c, d_dst, tmp, bytes);
        free(tmp);
    } else rc = h2d_pipelined_from(c, d_dst, (const void *)(size_t)16, bytes, fd);
    close(fd);
    return rc;
}

// ---- 4-byte integer columns (I32 / DATE / TIME, core/rayforce.h:54-58) as 8-byte device columns ----
// The kernels read 8-byte elements; a 4-byte column is uploaded as it is (half the PCIe bytes) and widened on the device with the
// reference's own promotion (i32_to_i64, core/ops.h:240: NULL_I32 -> NULL_I64, everything else sign-extended), which keeps order and
// equality -- nulls included -- so that comparisons on the widened column answer what cmp.c's i32 arms answer (core/cmp.c:150-166).
__global__ __launch_bounds__(256) void k_widen_i32(const int32_t *__restrict__ in, long long n, long long *__restrict__ out) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256LL) {
        const int32_t x = in[i];
        out[i] = x == (int32_t)0x80000000 ? (long long)0x8000000000000000ULL : (long long)x;
    }
}
// a B8 mask as an i64 column of 0 / 1: what lets a selection that exists only as a mask (a `where:` tree beyond the fused form) run as ONE
// comparison `(!= m 0)` of the fused pass on every shard, instead of mask -> ids -> gathered columns on one
__global__ __launch_bounds__(256) void k_widen_b8(const int8_t *__restrict__ in, long long n, long long *__restrict__ out) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256LL) out[i] = in[i] ? 1LL : 0LL;
}
extern "C" int rfx_hip_widen_b8(rfx_ctx_t *c, const int8_t *d_in, int64_t n, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in && d_out, RFX_EINVAL, "NULL argument");
    long long blocks = (n + 255) / 256;
    int grid = c->num_cus * 16;
    if (blocks < grid) grid = (int)blocks;
    hipLaunchKernelGGL(k_widen_b8, dim3(grid), dim3(256), 0, c->stream, d_in, (long long)n, (long long *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_widen_i32(rfx_ctx_t *c, const int32_t *d_in, int64_t n, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in && d_out, RFX_EINVAL, "NULL argument");
    long long blocks = (n + 255) / 256;
    int grid = c->num_cus * 16;
    if (blocks < grid) grid = (int)blocks;
    hipLaunchKernelGGL(k_widen_i32, dim3(grid), dim3(256), 0, c->stream, d_in, (long long)n, (long long *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
