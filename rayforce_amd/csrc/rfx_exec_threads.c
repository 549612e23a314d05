/* rfx_exec_threads.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  timing, the shard threads (one persistent host thread per shard, phases handed over without a lock), create / destroy, probes, communicators and the inter-process transport. */
void rfx_exec_timing(rfx_exec_t *x, int on) {
    if (!x) return;
    if (on) for (int i = RFX_XSTAT_NS_SCOPE; i <= RFX_XSTAT_NS_TOTAL; i++) x->stat[i] = 0;
    x->timing = on ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------------ shards and workers */
typedef struct {
    rfx_exec_t *x;
    int s;
} worker_arg_t;

/* A phase hand-over is on the query's critical path four to six times (a condition-variable round trip is ~20 us per phase: 0.1 ms of a
 * 0.8 ms query at 8 devices): workers and the caller SPIN on the generation / pending words for a bounded time first (a phase follows the
 * previous one within microseconds while a query runs) and only then sleep on the condition variable (between queries). */
#define SPIN_ROUNDS 4000 /* ~50 us of polling (a `pause` is ~40-60 cycles) */
static inline void cpu_relax(void) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
static void *worker_main(void *p) {
    worker_arg_t *wa = (worker_arg_t *)p;
    rfx_exec_t *x = wa->x;
    const int s = wa->s;
    free(wa);
    if (x->ctx[s]) rfx_hip_ctx_bind_thread(x->ctx[s]);
    uint64_t seen = 0;
    for (;;) {
        for (int i = 0; i < SPIN_ROUNDS; i++) {
            if (__atomic_load_n(&x->gen, __ATOMIC_ACQUIRE) != seen || __atomic_load_n(&x->stop, __ATOMIC_ACQUIRE)) break;
            cpu_relax();
        }
        if (__atomic_load_n(&x->gen, __ATOMIC_ACQUIRE) == seen && !__atomic_load_n(&x->stop, __ATOMIC_ACQUIRE)) { /* nothing came while polling: sleep */
            pthread_mutex_lock(&x->mu);
            while (x->gen == seen && !x->stop) {
                x->sleepers++;
                pthread_cond_wait(&x->cv_go, &x->mu);
                x->sleepers--;
            }
            pthread_mutex_unlock(&x->mu);
        }
        if (__atomic_load_n(&x->stop, __ATOMIC_ACQUIRE)) return NULL;
        /* (the fast path takes no lock: fn / arg were written before the generation's release store -- seven workers queueing for one mutex were
         *  most of a hand-over's 11 us at 8 shards) */
        seen = __atomic_load_n(&x->gen, __ATOMIC_ACQUIRE);
        shard_fn fn = x->fn;
        void *arg = x->arg;
        const int rc = fn(arg, s);
        if (rc != RFX_OK) snprintf(x->errs[s], sizeof(x->errs[s]), "shard %d: %s", s, rfx_hip_last_error());
        x->rcs[s] = rc;
        if (__atomic_sub_fetch(&x->pending, 1, __ATOMIC_ACQ_REL) == 0) {
            pthread_mutex_lock(&x->mu); /* (the caller may be asleep on cv_done by now) */
            pthread_cond_signal(&x->cv_done);
            pthread_mutex_unlock(&x->mu);
        }
    }
}

/* one phase: fn on every shard, the first failure's code back (its text in x->err) */
static int run_shards(rfx_exec_t *x, shard_fn fn, void *arg) {
    if (x->nshards == 1) {
        const int rc = fn(arg, 0);
        if (rc != RFX_OK) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
        return rc;
    }
    pthread_mutex_lock(&x->mu);
    x->fn = fn;
    x->arg = arg;
    __atomic_store_n(&x->pending, x->nshards - 1, __ATOMIC_RELEASE);
    __atomic_store_n(&x->gen, x->gen + 1, __ATOMIC_RELEASE);
    if (x->sleepers) pthread_cond_broadcast(&x->cv_go);
    pthread_mutex_unlock(&x->mu);
    x->rcs[0] = fn(arg, 0);
    if (x->rcs[0] != RFX_OK) snprintf(x->errs[0], sizeof(x->errs[0]), "shard 0: %s", rfx_hip_last_error());
    for (int i = 0; i < SPIN_ROUNDS && __atomic_load_n(&x->pending, __ATOMIC_ACQUIRE); i++) cpu_relax();
    if (__atomic_load_n(&x->pending, __ATOMIC_ACQUIRE)) {
        pthread_mutex_lock(&x->mu);
        while (__atomic_load_n(&x->pending, __ATOMIC_ACQUIRE)) pthread_cond_wait(&x->cv_done, &x->mu);
        pthread_mutex_unlock(&x->mu);
    }
    for (int s = 0; s < x->nshards; s++)
        if (x->rcs[s] != RFX_OK) {
            snprintf(x->err, sizeof(x->err), "%s", x->errs[s]);
            return x->rcs[s];
        }
    return RFX_OK;
}

int rfx_exec_run(rfx_exec_t *x, int (*fn)(void *arg, int shard), void *arg) {
    if (!x || !fn) return RFX_EINVAL;
    return run_shards(x, fn, arg);
}

int rfx_exec_create(rfx_ctx_t *const *ctxs, int nshards, rfx_exec_t **out) {
    if (!ctxs || !out || nshards < 1 || nshards > RFX_MAX_SHARDS) return RFX_EINVAL;
    rfx_exec_t *x = (rfx_exec_t *)calloc(1, sizeof(*x));
    if (!x) return RFX_ENOMEM;
    x->nshards = nshards;
    x->slice_shards = getenv("RFX_EXEC_SLICE_SHARDS") != NULL;
    x->no_d2h_pipeline = getenv("RFX_NO_D2H_PIPELINE") != NULL;
    x->two_step_rank = getenv("RFX_TWO_STEP_RANK") != NULL;
    for (int s = 0; s < nshards; s++) {
        if (!ctxs[s]) { free(x); return RFX_EINVAL; }
        x->ctx[s] = ctxs[s];
        x->dev[s] = rfx_hip_ctx_device(ctxs[s]);
        x->lead[s] = s;
        for (int t = 0; t < s; t++)
            if (x->dev[t] == x->dev[s]) { x->lead[s] = x->lead[t]; break; }
        if (x->lead[s] == s) x->devlead[x->ndev++] = s;
    }
    pthread_mutex_init(&x->mu, NULL);
    pthread_cond_init(&x->cv_go, NULL);
    pthread_cond_init(&x->cv_done, NULL);
    for (int s = 1; s < nshards; s++) {
        worker_arg_t *wa = (worker_arg_t *)malloc(sizeof(*wa));
        if (!wa) break;
        wa->x = x;
        wa->s = s;
        if (pthread_create(&x->th[s], NULL, worker_main, wa) != 0) { free(wa); break; }
        x->nthreads = s;
    }
    if (x->nthreads != nshards - 1) {
        rfx_exec_destroy(x);
        return RFX_ENOMEM;
    }
    *out = x;
    return RFX_OK;
}

int rfx_exec_destroy(rfx_exec_t *x) {
    if (!x) return RFX_OK;
    pthread_mutex_lock(&x->mu);
    __atomic_store_n(&x->stop, 1, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&x->cv_go);
    pthread_mutex_unlock(&x->mu);
    for (int s = 1; s <= x->nthreads; s++) pthread_join(x->th[s], NULL);
    if (x->comm_all)
        for (int d = 0; d < x->ndev; d++) rfx_dist_finalize(x->ctx[x->devlead[d]]);
    pthread_mutex_destroy(&x->mu);
    pthread_cond_destroy(&x->cv_go);
    pthread_cond_destroy(&x->cv_done);
    free(x);
    return RFX_OK;
}

/* what one phase hand-over costs the calling thread with `nshards` shards: a pool of nshards - 1 bare worker threads (no device), `reps`
 * empty phases, microseconds per phase.  bench.py's predicted T(N) charges it per phase of a sharded query. */
static int ph_nothing(void *arg, int s) { (void)arg; (void)s; return RFX_OK; }
double rfx_exec_probe_handover_us(int nshards, int reps) {
    if (nshards < 1 || nshards > RFX_MAX_SHARDS || reps < 1) return -1.0;
    rfx_exec_t *x = (rfx_exec_t *)calloc(1, sizeof(*x));
    if (!x) return -1.0;
    x->nshards = nshards;
    pthread_mutex_init(&x->mu, NULL);
    pthread_cond_init(&x->cv_go, NULL);
    pthread_cond_init(&x->cv_done, NULL);
    for (int s = 1; s < nshards; s++) {
        worker_arg_t *wa = (worker_arg_t *)malloc(sizeof(*wa));
        if (!wa) break;
        wa->x = x;
        wa->s = s;
        if (pthread_create(&x->th[s], NULL, worker_main, wa) != 0) { free(wa); break; }
        x->nthreads = s;
    }
    double us = -1.0;
    if (x->nthreads == nshards - 1) {
        for (int i = 0; i < 16; i++) run_shards(x, ph_nothing, NULL);
        const int64_t t0 = now_ns();
        for (int i = 0; i < reps; i++) run_shards(x, ph_nothing, NULL);
        us = (double)(now_ns() - t0) / 1e3 / reps;
    }
    rfx_exec_destroy(x);
    return us;
}
int rfx_exec_shards(const rfx_exec_t *x) { return x ? x->nshards : 0; }
rfx_ctx_t *rfx_exec_ctx(const rfx_exec_t *x, int shard) { return (x && shard >= 0 && shard < x->nshards) ? x->ctx[shard] : NULL; }
int64_t rfx_exec_stat(const rfx_exec_t *x, int which) { return (x && which >= 0 && which < RFX_XSTAT_N) ? x->stat[which] : -1; }
const char *rfx_exec_last_error(const rfx_exec_t *x) { return x ? x->err : "rfx_exec: NULL"; }
void rfx_exec_forget_scopes(rfx_exec_t *x) {
    if (x) x->nspec_failed = x->spec_ring = x->nspec_wide = x->wide_ring = 0;
}

void rfx_exec_split(int64_t nrows, int nshards, int shard, int64_t *row0, int64_t *len) {
    int64_t span = nshards > 0 ? (nrows + nshards - 1) / nshards : nrows;
    span = (span + 511) & ~(int64_t)511; /* whole 4 KB of every 8-byte column per shard boundary */
    int64_t r0 = (int64_t)shard * span;
    if (r0 > nrows) r0 = nrows;
    int64_t n = nrows - r0 < span ? nrows - r0 : span;
    if (row0) *row0 = r0;
    if (len) *len = n;
}

int rfx_exec_comm_init_all(rfx_exec_t *x) {
    if (!x) return RFX_EINVAL;
    /* (RFX_EXEC_FORCE_RCCL=1: communicators even over ONE device -- a one-rank RCCL world: how the fused exchange's code path runs on a
     * one-GPU box, with the shards beside the lead still merged by the kernel) */
    if ((x->ndev <= 1 && !getenv("RFX_EXEC_FORCE_RCCL")) || x->comm_all) return RFX_OK;
    rfx_ctx_t *leads[RFX_MAX_SHARDS];
    for (int d = 0; d < x->ndev; d++) leads[d] = x->ctx[x->devlead[d]];
    const int rc = rfx_dist_init_all(leads, x->ndev);
    if (rc != RFX_OK) {
        snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
        return rc;
    }
    x->comm_all = 1;
    return RFX_OK;
}

int rfx_exec_set_transport(rfx_exec_t *x, const rfx_transport_t *t) {
    if (!x) return RFX_EINVAL;
    if (t) {
        x->tr = *t;
        x->has_tr = 1;
    } else x->has_tr = 0;
    return RFX_OK;
}

/* ---- the inter-process side: a transport of the host's, else the lead context's RCCL communicator (unless that one is process-local) ---- */
/* returns 1 when there IS an inter-process exchange (a one-rank communicator still runs it: that is how its fixed cost is measured) */
static int world_rank(rfx_exec_t *x, int *world, int *rank) {
    *world = 1;
    *rank = 0;
    if (x->has_tr && x->tr.world_rank) {
        x->tr.world_rank(x->tr.user, world, rank);
        return 1;
    }
    if (!x->comm_all && !rfx_dist_is_local(x->ctx[0])) {
        int w = 0, r = 0;
        if (rfx_dist_world(x->ctx[0], &w, &r) == RFX_OK && w >= 1) {
            *world = w;
            *rank = r;
        }
        return rfx_dist_has_comm(x->ctx[0]);
    }
    return 0;
}
static int xp_allgather_host(rfx_exec_t *x, const void *in, size_t bytes, void *out) {
    x->stat[RFX_XSTAT_MERGES_TRANSPORT]++;
    if (x->has_tr) return x->tr.allgather_host(x->tr.user, in, bytes, out);
    return rfx_dist_allgather_host(x->ctx[0], in, bytes, out);
}
int rfx_exec_ranks(rfx_exec_t *x, int *rank) {
    int w = 1, r = 0;
    if (rank) *rank = 0;
    if (!x || !world_rank(x, &w, &r) || w <= 1) return 1;
    if (rank) *rank = r;
    return w;
}
int rfx_exec_allgather_host(rfx_exec_t *x, const void *in, size_t bytes, void *out) {
    if (!x || !in || !out || !bytes) return RFX_EINVAL;
    return xp_allgather_host(x, in, bytes, out);
}
static int xp_allreduce(rfx_exec_t *x, void *d_buf, int64_t n, int type, int op) {
    x->stat[RFX_XSTAT_MERGES_TRANSPORT]++;
    if (x->has_tr) return x->tr.allreduce(x->tr.user, d_buf, n, type, op);
    (void)type; /* RCCL sums of f64 cells go through the tables' own exchange; this form carries integers */
    return rfx_dist_allreduce_i64(x->ctx[0], (int64_t *)d_buf, n, op);
}
static int xp_allgather_dev(rfx_exec_t *x, const void *d_in, size_t bytes, void *d_out) {
    x->stat[RFX_XSTAT_MERGES_TRANSPORT]++;
    if (x->has_tr) return x->tr.allgather_dev(x->tr.user, d_in, bytes, d_out);
    return rfx_dist_allgather(x->ctx[0], d_in, bytes, d_out);
}
/* logical OR of one flag over the processes */
static int xp_any(rfx_exec_t *x, int world, int flag, int *any) {
    *any = flag;
    if (world <= 0) return RFX_OK; /* (0: no exchange at all) */
    int64_t mine = flag, all[256];
    if (world > 256) return RFX_ELIMIT;
    const int rc = xp_allgather_host(x, &mine, 8, all);
    if (rc != RFX_OK) return rc;
    for (int r = 0; r < world; r++) *any |= all[r] != 0;
    return RFX_OK;
}
