/*
 * rfx_exec.c -- the planner (include/rfx_exec.h): one select / where / by query over one or several row-range shards, in C.
 *
 * Every decision that is not a kernel lives here and nowhere else: sampled vs. exact key scope and the retry when a pass reports a key
 * outside a sampled one; dense "perfect hash" vs. open addressing (range <= rows, core/index.c:2013); table sizing and growth; the
 * composite key of several `by:` columns (index_group_list_perfect, core/index.c:2308-2424) and the row-hash route beyond 64 bits
 * (index_group_list, core/index.c:2731-2790); how many passes a long output list takes; and -- the part the reference does inside
 * ray_select with its pool (core/query.c:607-654 -> aggr_map core/aggr.c:375 -> pool_run core/pool.c:369-424, merged by AGGR_COLLECT
 * core/aggr.c:163-181) -- how the shards' partial states become one answer.  rfx_ops.c (obj_p door), the Python test host and bench.py
 * are callers; none of them chooses a path.
 *
 * Execution model: a query is a sequence of PHASES; a phase runs the same step on every shard, each shard on its own host thread
 * (a worker bound to the shard's device; shard 0 on the calling thread), and ends when every shard's stream is idle.  Between phases
 * the calling thread folds what the shards report (scopes, flags, scalar partials) and issues the exchanges.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rfx_exec.h"

#include <time.h>

#define NULL_I64 ((int64_t)0x8000000000000000LL)
#define INF_I64 ((int64_t)0x7FFFFFFFFFFFFFFFLL)

typedef int (*shard_fn)(void *arg, int s);

struct rfx_exec {
    int nshards;
    rfx_ctx_t *ctx[RFX_MAX_SHARDS];
    int dev[RFX_MAX_SHARDS];
    int lead[RFX_MAX_SHARDS]; /* the first shard on this shard's device */
    int ndev;
    int devlead[RFX_MAX_SHARDS]; /* lead shard of every distinct device */
    int comm_all;                /* process-local communicators among the device leads */
    /* workers: shard s > 0 runs on th[s] */
    pthread_t th[RFX_MAX_SHARDS];
    int nthreads;
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    uint64_t gen;
    int pending, stop, sleepers;
    shard_fn fn;
    void *arg;
    int rcs[RFX_MAX_SHARDS];
    char errs[RFX_MAX_SHARDS][256];
    /* inter-process exchange */
    rfx_transport_t tr;
    int has_tr;
    /* sampled scopes that were reported too small: not sampled again (a rare extreme value would be missed again) */
    const void *spec_failed[32];
    int64_t spec_failed_n[32];
    int nspec_failed, spec_ring;
    /* ... and key columns whose sample said "wider than the LDS forms": the sample (a launch and a round trip) is not taken again for them */
    const void *spec_wide[32];
    int64_t spec_wide_n[32];
    int nspec_wide, wide_ring;
    int64_t stat[RFX_XSTAT_N];
    int timing;       /* rfx_exec_timing: per-phase wall time into stat[RFX_XSTAT_NS_*], a sync at every phase end */
    int64_t rh_kmin[RFX_MAX_KEYS], rh_kmax[RFX_MAX_KEYS]; /* the key columns' scopes of the last row-hash query (a null key's stand-in in the sharded proof) */
    int two_step_rank;   /* RFX_TWO_STEP_RANK=1: rank, the group count back, then emit (rounds 1-4; A/B) instead of rank -> emit without the round trip */
    int no_d2h_pipeline; /* RFX_NO_D2H_PIPELINE=1: large result columns by one plain copy each (A/B) */
    int slice_shards; /* RFX_EXEC_SLICE_SHARDS=1: every SHARD owns a slice of a sliced result, not only every device's lead (how the sharded
                       * tail runs on a one-GPU box: the merged tables are copied to the shards beside their lead first) */
    char err[512];
};
static inline int64_t now_ns(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}
#define T_BEGIN(x) const int64_t t0_ = (x)->timing ? now_ns() : 0
#define T_END(x, which) do { if ((x)->timing) (x)->stat[which] += now_ns() - t0_; } while (0)
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

/* ------------------------------------------------------------------------------------------------ one shard's view of a query */
#define SH_TMP 64
typedef struct {
    rfx_pred_t preds[RFX_MAX_PREDS];
    rfx_agg_t aggs[RFX_MAX_AGGS];
    rfx_xnode_t xn[RFX_MAX_AGGS][RFX_MAX_XNODES];
    const void *keys[RFX_MAX_KEYS];
    const void *key; /* the column grouped on: key 0, the composite key or the row hash */
    const int8_t *mask;
    int64_t nrows, row0; /* row0: GLOBAL id of this shard's row 0 */
    void *tmp[SH_TMP];
    int ntmp;
    /* tables */
    void *store;
    rfx_group_tables_t gt;
    rfx_hash_tables_t ht;
    /* what a phase reports */
    int64_t mn[RFX_MAX_KEYS], mx[RFX_MAX_KEYS], seen;
    int flag, arc;
    rfx_partial_t part[RFX_MAX_AGGS + 1];
    /* rank + emit */
    int64_t groups;
    void *dout, *dfirst;
    int64_t g0, gn;              /* the slice of the groups this shard emitted (the whole result: 0, groups) */
    int64_t gstride;             /* cells between two columns of dout (gn, or the bound the one-launch rank + emit sized them by) */
    void *kc[RFX_MAX_KEYS];      /* sliced result, several keys: this slice's key columns */
    int64_t t_rank;              /* timing: when this shard's ranking was done */
    /* where */
    int64_t *d_ids, count;
    /* the selection of a mask query as ids (first rows are translated back through them) */
    int64_t *sel_ids;
} shard_t;

static const void *xlate(const rfx_query_t *q, int s, const void *p, int *bad) {
    if (!p || s == 0) return p;
    for (int i = 0; i < q->ncols; i++)
        if (q->cols[i].d[0] == p) return q->cols[i].d[s];
    *bad = 1;
    return NULL;
}
/* shard s's copy of the comparisons, of aggregates [a0, a0 + na) and of the key columns */
static int shard_view(const rfx_query_t *q, int S, int s, int a0, int na, shard_t *h) {
    int bad = 0;
    if (S > 1 && !q->cols) return RFX_EINVAL;
    for (int i = 0; i < q->npred; i++) {
        h->preds[i] = q->preds[i];
        h->preds[i].d_col = xlate(q, s, q->preds[i].d_col, &bad);
        h->preds[i].d_rhs_col = xlate(q, s, q->preds[i].d_rhs_col, &bad);
    }
    for (int a = 0; a < na; a++) {
        const rfx_agg_t *src = &q->aggs[a0 + a];
        h->aggs[a] = *src;
        h->aggs[a].d_col = xlate(q, s, src->d_col, &bad);
        h->aggs[a].d_xrhs_col = xlate(q, s, src->d_xrhs_col, &bad);
        if (src->nxnodes > 0) {
            if (src->nxnodes > RFX_MAX_XNODES || !src->xnodes) return RFX_EINVAL;
            for (int j = 0; j < src->nxnodes; j++) {
                h->xn[a][j] = src->xnodes[j];
                if (h->xn[a][j].l.kind == RFX_XK_COL) h->xn[a][j].l.d_col = xlate(q, s, src->xnodes[j].l.d_col, &bad);
                if (h->xn[a][j].r.kind == RFX_XK_COL) h->xn[a][j].r.d_col = xlate(q, s, src->xnodes[j].r.d_col, &bad);
            }
            h->aggs[a].xnodes = h->xn[a];
        }
    }
    for (int k = 0; k < q->nkeys; k++) h->keys[k] = xlate(q, s, q->d_keys[k], &bad);
    h->key = q->nkeys ? h->keys[0] : NULL;
    h->mask = (const int8_t *)xlate(q, s, q->d_mask, &bad);
    return bad ? RFX_EINVAL : RFX_OK;
}
static int sh_malloc(rfx_exec_t *x, shard_t *h, int s, void **p, size_t bytes) {
    *p = NULL;
    if (h->ntmp >= SH_TMP) return RFX_ELIMIT;
    const int rc = rfx_hip_malloc(x->ctx[s], p, bytes ? bytes : 8);
    if (rc == RFX_OK) h->tmp[h->ntmp++] = *p;
    return rc;
}
static void sh_release(rfx_exec_t *x, shard_t *h, int s) {
    for (int i = 0; i < h->ntmp; i++) rfx_hip_free(x->ctx[s], h->tmp[i]);
    h->ntmp = 0;
    if (h->store) rfx_hip_free(x->ctx[s], h->store);
    h->store = NULL;
    if (h->dout) rfx_hip_free(x->ctx[s], h->dout);
    if (h->dfirst) rfx_hip_free(x->ctx[s], h->dfirst);
    h->dout = h->dfirst = NULL;
    for (int k = 0; k < RFX_MAX_KEYS; k++) {
        if (h->kc[k]) rfx_hip_free(x->ctx[s], h->kc[k]);
        h->kc[k] = NULL;
    }
    if (h->sel_ids) rfx_hip_free(x->ctx[s], h->sel_ids);
    h->sel_ids = NULL;
}

/* how many of the aggregates from a0 on one pass carries: <= RFX_MAX_AGGS, <= RFX_MAX_EXPRS expressions and a handful of distinct argument
 * columns (predicate and key columns need plan slots too: RFX_MAX_COLS in all) */
static int agg_chunk(const rfx_query_t *q, int a0) {
    const void *cols[4 * RFX_MAX_AGGS];
    int ncols = 0, nx = 0, n = 0;
    for (int a = a0; a < q->nagg && n < RFX_MAX_AGGS; a++, n++) {
        const rfx_agg_t *g = &q->aggs[a];
        const void *mine[2 + 2 * RFX_MAX_XNODES];
        int nm = 0;
        const int isx = g->nxnodes > 0 || g->xop != RFX_X_NONE;
        if (g->nxnodes < 0 || g->nxnodes > RFX_MAX_XNODES || (g->nxnodes > 0 && !g->xnodes)) return -1; /* (the callers answer RFX_EINVAL) */
        if (g->nxnodes > 0) {
            for (int j = 0; j < g->nxnodes; j++) {
                if (g->xnodes[j].l.kind == RFX_XK_COL) mine[nm++] = g->xnodes[j].l.d_col;
                if (g->xnodes[j].r.kind == RFX_XK_COL) mine[nm++] = g->xnodes[j].r.d_col;
            }
        } else {
            if (g->d_col) mine[nm++] = g->d_col;
            if (g->d_xrhs_col) mine[nm++] = g->d_xrhs_col;
        }
        int add = 0;
        for (int i = 0; i < nm; i++) {
            int known = 0;
            for (int j = 0; j < ncols + add && !known; j++) known = cols[j] == mine[i];
            if (!known) cols[ncols + add++] = mine[i];
        }
        if (n > 0 && (nx + isx > RFX_MAX_EXPRS || ncols + add > 4)) break;
        ncols += add;
        nx += isx;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------ a mask query: the selection as ids,
 * every column the query reads gathered at them (the reference's own plan for trees it cannot fuse either: filter_collect, then fold /
 * group -- core/filter.c:51-165).  One shard. */
static int gather_selected(rfx_exec_t *x, shard_t *h, int na, int nkeys) {
    rfx_ctx_t *c = x->ctx[0];
    int64_t nsel = 0;
    int rc = rfx_hip_where_begin(c, NULL, 0, RFX_AND, h->mask, h->nrows, &nsel);
    if (rc != RFX_OK) return rc;
    void *ids = NULL;
    rc = rfx_hip_malloc(c, &ids, (size_t)(nsel ? nsel : 1) * 8);
    if (rc != RFX_OK) return rc;
    h->sel_ids = (int64_t *)ids;
    if (nsel && (rc = rfx_hip_where_emit(c, 0, h->sel_ids)) != RFX_OK) return rc;
    const void **slots[RFX_MAX_AGGS * (2 + 2 * RFX_MAX_XNODES) + RFX_MAX_KEYS];
    int nslots = 0;
    for (int a = 0; a < na; a++) {
        slots[nslots++] = &h->aggs[a].d_col;
        slots[nslots++] = &h->aggs[a].d_xrhs_col;
        for (int j = 0; j < h->aggs[a].nxnodes; j++) {
            if (h->xn[a][j].l.kind == RFX_XK_COL) slots[nslots++] = &h->xn[a][j].l.d_col;
            if (h->xn[a][j].r.kind == RFX_XK_COL) slots[nslots++] = &h->xn[a][j].r.d_col;
        }
    }
    for (int k = 0; k < nkeys; k++) slots[nslots++] = &h->keys[k];
    const void *src[RFX_MAX_AGGS * (2 + 2 * RFX_MAX_XNODES) + RFX_MAX_KEYS];
    void *dst[RFX_MAX_AGGS * (2 + 2 * RFX_MAX_XNODES) + RFX_MAX_KEYS];
    int nseen = 0;
    for (int i = 0; i < nslots; i++) {
        if (!*slots[i]) continue;
        int j = 0;
        for (; j < nseen; j++)
            if (src[j] == *slots[i]) break;
        if (j == nseen) { /* a column several descriptors read is gathered once */
            void *g = NULL;
            if ((rc = sh_malloc(x, h, 0, &g, (size_t)(nsel ? nsel : 1) * 8)) != RFX_OK) return rc;
            if (nsel && (rc = rfx_hip_gather(c, *slots[i], h->sel_ids, nsel, g)) != RFX_OK) return rc;
            src[nseen] = *slots[i];
            dst[nseen++] = g;
        }
        *slots[i] = dst[j];
    }
    h->key = nkeys ? h->keys[0] : NULL;
    h->nrows = nsel;
    h->mask = NULL;
    return RFX_OK;
}

/* ------------------------------------------------------------------------------------------------ scalar aggregates */
typedef struct {
    rfx_exec_t *x;
    const rfx_query_t *q;
    int S, na, npred;
    int64_t proc_row0;
    shard_t *sh;
} fa_t;
/* a selection by row ids: every column the aggregates read, gathered at this shard's ids (filter_collect, core/filter.c:51-165, on the
 * device); the fold then runs over the gathered rows, positioned after the lower shards' ids */
static int gather_at_ids(rfx_exec_t *x, shard_t *h, int s, int na, const int64_t *d_ids, int64_t n, int64_t shard_row0) {
    rfx_ctx_t *c = x->ctx[s];
    const void **slots[RFX_MAX_AGGS * (2 + 2 * RFX_MAX_XNODES)];
    int nslots = 0;
    for (int a = 0; a < na; a++) {
        slots[nslots++] = &h->aggs[a].d_col;
        slots[nslots++] = &h->aggs[a].d_xrhs_col;
        for (int j = 0; j < h->aggs[a].nxnodes; j++) {
            if (h->xn[a][j].l.kind == RFX_XK_COL) slots[nslots++] = &h->xn[a][j].l.d_col;
            if (h->xn[a][j].r.kind == RFX_XK_COL) slots[nslots++] = &h->xn[a][j].r.d_col;
        }
    }
    const void *src[RFX_MAX_AGGS * (2 + 2 * RFX_MAX_XNODES)];
    void *dst[RFX_MAX_AGGS * (2 + 2 * RFX_MAX_XNODES)];
    int nseen = 0, rc;
    for (int i = 0; i < nslots; i++) {
        if (!*slots[i]) continue;
        int j = 0;
        for (; j < nseen; j++)
            if (src[j] == *slots[i]) break;
        if (j == nseen) {
            void *g = NULL;
            if ((rc = sh_malloc(x, h, s, &g, (size_t)(n ? n : 1) * 8)) != RFX_OK) return rc;
            /* (the shard's piece addressed by GLOBAL ids: its base moved back by the shard's first row) */
            if (n && (rc = rfx_hip_gather(c, (const char *)*slots[i] - (size_t)shard_row0 * 8, d_ids, n, g)) != RFX_OK) return rc;
            src[nseen] = *slots[i];
            dst[nseen++] = g;
        }
        *slots[i] = dst[j];
    }
    return RFX_OK;
}
static int ph_filter_aggr(void *arg, int s) {
    fa_t *F = (fa_t *)arg;
    shard_t *h = &F->sh[s];
    rfx_ctx_t *c = F->x->ctx[s];
    void *d = NULL;
    int rc = sh_malloc(F->x, h, s, &d, sizeof(rfx_partial_t) * (size_t)(F->na + 1));
    if (rc != RFX_OK) return rc;
    if (F->q->d_sel_ids) {
        int64_t before = 0;
        for (int t = 0; t < s; t++) before += F->q->sel_count[t];
        if ((rc = gather_at_ids(F->x, h, s, F->na, F->q->d_sel_ids[s], F->q->sel_count[s], h->row0)) != RFX_OK) return rc;
        h->nrows = F->q->sel_count[s];
        h->row0 = before;
    }
    rc = rfx_hip_filter_aggr(c, h->preds, F->npred, F->q->logic, h->aggs, F->na, h->nrows, h->row0, (rfx_partial_t *)d);
    if (rc != RFX_OK) return rc;
    return rfx_hip_d2h(c, h->part, d, sizeof(rfx_partial_t) * (size_t)(F->na + 1));
}

int rfx_exec_filter_aggr(rfx_exec_t *x, const rfx_query_t *q, rfx_value_t *values, int64_t *selected) {
    if (!x || !q || !values || q->nagg < 0 || q->nagg > RFX_EXEC_MAX_AGGS || q->npred < 0 || q->npred > RFX_MAX_PREDS) return RFX_EINVAL;
    const int S = x->nshards;
    int world, rank;
    const int exch = world_rank(x, &world, &rank);
    x->err[0] = 0;
    if (q->d_mask && (S > 1 || exch || q->npred)) {
        snprintf(x->err, sizeof(x->err), "rfx_exec: a mask selection runs on one shard, without comparisons beside it");
        return RFX_ELIMIT;
    }
    if (q->d_sel_ids && (q->npred || q->d_mask || exch || !q->sel_count)) {
        snprintf(x->err, sizeof(x->err), "rfx_exec: a selection by row ids stands alone (no comparisons, no mask) inside one process");
        return RFX_EINVAL;
    }
    rfx_hip_ctx_bind_thread(x->ctx[0]);
    x->stat[RFX_XSTAT_QUERIES]++;
    x->err[0] = 0;
    shard_t *sh = (shard_t *)calloc((size_t)S, sizeof(shard_t));
    if (!sh) return RFX_ENOMEM;
    int rc = RFX_OK;
    if (selected) *selected = 0;
    for (int a0 = 0; (a0 < q->nagg || (a0 == 0 && q->nagg == 0)) && rc == RFX_OK;) {
        const int na = q->nagg ? agg_chunk(q, a0) : 0;
        if (na < 0) { snprintf(x->err, sizeof(x->err), "rfx_exec: aggregate %d: nxnodes outside 0..%d or xnodes NULL", a0, RFX_MAX_XNODES); rc = RFX_EINVAL; break; }
        fa_t F = {x, q, S, na, q->npred, 0, sh};
        for (int s = 0; s < S && rc == RFX_OK; s++) {
            rc = shard_view(q, S, s, a0, na, &sh[s]);
            rfx_exec_split(q->nrows, S, s, &sh[s].row0, &sh[s].nrows);
        }
        if (rc != RFX_OK) snprintf(x->err, sizeof(x->err), "rfx_exec: a column of the query has no per-shard address");
        if (rc == RFX_OK && q->d_mask) {
            rc = gather_selected(x, &sh[0], na, 0);
            F.npred = 0;
        }
        if (rc == RFX_OK) rc = run_shards(x, ph_filter_aggr, &F);
        if (rc == RFX_OK) {
            rfx_partial_t acc[RFX_MAX_AGGS + 1];
            memcpy(acc, sh[0].part, sizeof(rfx_partial_t) * (size_t)(na + 1));
            for (int s = 1; s < S; s++) { /* shard order = row order: FIRST keeps the lowest row, f64 sums add in a fixed order */
                for (int a = 0; a < na; a++) rfx_partial_merge(sh[0].aggs[a].kind, rfx_agg_input_type(&sh[0].aggs[a]), &acc[a], &sh[s].part[a]);
                rfx_partial_merge(RFX_AGG_COUNT, RFX_I64, &acc[na], &sh[s].part[na]);
            }
            if (exch) { /* one exchange: every process' folded partials, folded again in rank order */
                rfx_partial_t *all = (rfx_partial_t *)malloc(sizeof(rfx_partial_t) * (size_t)(na + 1) * (size_t)world);
                if (!all) rc = RFX_ENOMEM;
                else {
                    /* FIRST positions are local to a process: make them global by the process' row offset, which the ranks do not know
                     * of each other -- rank order IS row order, so a lower rank's FIRST wins whatever the positions say */
                    rc = xp_allgather_host(x, acc, sizeof(rfx_partial_t) * (size_t)(na + 1), all);
                    if (rc == RFX_OK) {
                        memcpy(acc, all, sizeof(rfx_partial_t) * (size_t)(na + 1));
                        for (int r = 1; r < world; r++) {
                            rfx_partial_t *o = all + (size_t)r * (size_t)(na + 1);
                            for (int a = 0; a < na; a++) {
                                if (sh[0].aggs[a].kind == RFX_AGG_FIRST) { /* the first rank that selected a row holds the first row */
                                    if (acc[a].pos == INF_I64 && o[a].pos != INF_I64) acc[a] = o[a];
                                    continue;
                                }
                                rfx_partial_merge(sh[0].aggs[a].kind, rfx_agg_input_type(&sh[0].aggs[a]), &acc[a], &o[a]);
                            }
                            rfx_partial_merge(RFX_AGG_COUNT, RFX_I64, &acc[na], &o[na]);
                        }
                    }
                    free(all);
                }
            }
            for (int a = 0; a < na && rc == RFX_OK; a++) rc = rfx_agg_finalize(sh[0].aggs[a].kind, rfx_agg_input_type(&sh[0].aggs[a]), &acc[a], &values[a0 + a]);
            if (selected) *selected = acc[na].cnt;
        }
        for (int s = 0; s < S; s++) sh_release(x, &sh[s], s);
        a0 += na;
        if (q->nagg == 0) break;
    }
    free(sh);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ where */
typedef struct {
    rfx_exec_t *x;
    const rfx_query_t *q;
    shard_t *sh;
} wh_t;
static int ph_where(void *arg, int s) {
    wh_t *W = (wh_t *)arg;
    shard_t *h = &W->sh[s];
    rfx_ctx_t *c = W->x->ctx[s];
    h->d_ids = NULL;
    h->count = 0;
    if (h->nrows == 0) return RFX_OK; /* (a shard without rows; an empty table's mask has no address at all) */
    if (h->mask) {
        int rc = rfx_hip_where_begin(c, NULL, 0, RFX_AND, h->mask, h->nrows, &h->count);
        if (rc != RFX_OK || h->count == 0) return rc;
        void *d = NULL;
        if ((rc = rfx_hip_malloc(c, &d, (size_t)h->count * 8)) != RFX_OK) return rc;
        h->d_ids = (int64_t *)d;
        return rfx_hip_where_emit(c, h->row0, h->d_ids);
    }
    /* one pass over the predicate columns (rfx_where_once.hip): the buffer by a sampled estimate, the count back exact, a second run if the
     * sample underestimated a clustered selection */
    int64_t cap = 0;
    int rc = rfx_hip_where_estimate(c, h->preds, W->q->npred, W->q->logic, h->nrows, &cap);
    if (rc != RFX_OK) return rc;
    for (int attempt = 0; attempt < 2; attempt++) {
        void *d = NULL;
        if (cap > 0 && (rc = rfx_hip_malloc(c, &d, (size_t)cap * 8)) != RFX_OK) return rc;
        rc = rfx_hip_where_once(c, h->preds, W->q->npred, W->q->logic, h->nrows, h->row0, (int64_t *)d, cap, &h->count);
        if (rc == RFX_OK) {
            if (h->count > 0) h->d_ids = (int64_t *)d;
            else if (d) rfx_hip_free(c, d);
            return RFX_OK;
        }
        if (d) rfx_hip_free(c, d);
        if (rc != RFX_ELIMIT || h->count <= cap) return rc;
        cap = h->count;
    }
    return rc;
}
int rfx_exec_where(rfx_exec_t *x, const rfx_query_t *q, rfx_ids_t *out) {
    if (!x || !q || !out || q->npred < 0 || q->npred > RFX_MAX_PREDS) return RFX_EINVAL;
    const int S = x->nshards;
    if (q->d_mask && q->npred) return RFX_EINVAL;
    rfx_hip_ctx_bind_thread(x->ctx[0]);
    x->stat[RFX_XSTAT_QUERIES]++;
    x->err[0] = 0;
    memset(out, 0, sizeof(*out));
    shard_t *sh = (shard_t *)calloc((size_t)S, sizeof(shard_t));
    if (!sh) return RFX_ENOMEM;
    int rc = RFX_OK;
    for (int s = 0; s < S && rc == RFX_OK; s++) {
        rc = shard_view(q, S, s, 0, 0, &sh[s]);
        rfx_exec_split(q->nrows, S, s, &sh[s].row0, &sh[s].nrows);
        sh[s].row0 += q->row0;
    }
    if (rc != RFX_OK) snprintf(x->err, sizeof(x->err), "rfx_exec: a column of the query has no per-shard address");
    wh_t W = {x, q, sh};
    if (rc == RFX_OK) rc = run_shards(x, ph_where, &W);
    out->nshards = S;
    for (int s = 0; s < S; s++) {
        if (rc == RFX_OK) {
            out->count[s] = sh[s].count;
            out->d_ids[s] = sh[s].d_ids;
            out->total += sh[s].count;
        } else if (sh[s].d_ids) rfx_hip_free(x->ctx[s], sh[s].d_ids);
        sh_release(x, &sh[s], s);
    }
    free(sh);
    return rc;
}
void rfx_exec_ids_free(rfx_exec_t *x, rfx_ids_t *ids) {
    if (!x || !ids) return;
    for (int s = 0; s < ids->nshards && s < x->nshards; s++)
        if (ids->d_ids[s]) rfx_hip_free(x->ctx[s], ids->d_ids[s]);
    memset(ids, 0, sizeof(*ids));
}

/* ------------------------------------------------------------------------------------------------ group-by */
typedef struct {
    rfx_exec_t *x;
    const rfx_query_t *q;
    shard_t *sh;
    int S, world, rank, exch; /* exch: there is an inter-process exchange (world processes) */
    int na, npred, nkeys;   /* aggregates of this pass; comparisons (0 once a mask was gathered) */
    int64_t total_rows;     /* rows of the whole table, all processes */
    int64_t proc_row0;      /* global id of this process' row 0 */
    /* the plan */
    int spec;               /* the scope is a sample: the pass reports keys outside it */
    int dense, fused_keys, rowhash, small;
    int64_t kmin, kmax, seen;
    uint64_t range;
    int64_t kmins[RFX_MAX_KEYS], kmaxs[RFX_MAX_KEYS], kmults[RFX_MAX_KEYS], comp_max;
    int64_t cap, cap_max;
    int narr;
    int want_first, need_first_values, all_rank;
    int nsl, slown[RFX_MAX_SHARDS], slidx[RFX_MAX_SHARDS], slice_all; /* result slices: their owners, a shard's slice (-1: none), owners beside device leads */
    int phase_key;          /* which key column a per-key phase works on */
    int scope_filtered;     /* per-key exact scopes: through the predicates */
    int sparse_sampled;     /* the sample alone sent the key to the hashed tables: a null key shows in their null slot */
    int64_t groups;
    /* the pass being run */
    int a0, first_pass, multi, any_xbar;
    int spec_ok, retried;   /* may the scope be sampled; did a sampled scope fail already */
    const void *spec_id;    /* what the planner remembers sampled scopes by */
    int64_t cap_hint;
    rfx_groups_t *out;
} gq_t;

static int has_cnt(const rfx_agg_t *a) { return a->kind == RFX_AGG_AVG || (a->kind == RFX_AGG_SUM && rfx_agg_input_type(a) == RFX_I64); }

/* bucketed keys: (xbar col width) is evaluated before grouping, as the reference does (ray_xbar, core/math.c:1635) */
static int ph_xbar(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    for (int k = 0; k < G->nkeys; k++) {
        if (!G->q->kxbar || G->q->kxbar[k] <= 0) continue;
        void *xb = NULL;
        int rc = sh_malloc(G->x, h, s, &xb, (size_t)(h->nrows ? h->nrows : 1) * 8);
        if (rc != RFX_OK) return rc;
        if ((rc = rfx_hip_xbar_i64(G->x->ctx[s], (const int64_t *)h->keys[k], h->nrows, G->q->kxbar[k], (int64_t *)xb)) != RFX_OK) return rc;
        h->keys[k] = xb;
    }
    h->key = h->keys[0];
    return RFX_OK;
}
/* scope of the key grouped on, sampled (one tiny launch) */
static int ph_scope_sample(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    h->seen = h->nrows;
    if (h->nrows == 0) return RFX_OK;
    for (int k = 0; k < G->nkeys; k++) {
        const int rc = rfx_hip_scope_sample_i64(G->x->ctx[s], (const int64_t *)h->keys[k], h->nrows, &h->mn[k], &h->mx[k]);
        if (rc != RFX_OK) return rc;
    }
    return RFX_OK;
}
/* exact scope of the single key through the predicates; for wide ranges the same read leaves the rows partitioned for the pass */
static int ph_scope_group(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    h->seen = 0;
    if (h->nrows == 0) return RFX_OK;
    return rfx_hip_group_scope(G->x->ctx[s], (const int64_t *)h->key, h->preds, G->npred, G->q->logic, h->aggs, G->na, h->nrows, &h->mn[0], &h->mx[0], &h->seen);
}
/* exact scope of one column (phase_key; -1: the column grouped on), with or without the predicates */
static int ph_scope_col(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    const int k = G->phase_key < 0 ? 0 : G->phase_key;
    const void *col = G->phase_key < 0 ? h->key : h->keys[k];
    h->seen = 0;
    if (h->nrows == 0) return RFX_OK;
    return rfx_hip_scope_i64(G->x->ctx[s], (const int64_t *)col, G->scope_filtered ? h->preds : NULL, G->scope_filtered ? G->npred : 0, G->q->logic, h->nrows, &h->mn[k], &h->mx[k], &h->seen);
}
/* fold the shards' (min, max, seen) of key k -- index_scope_i64 takes a null key as the value INT64_MIN, so a plain minimum keeps it -- and
 * agree with the other processes (one exchange: 32 bytes a rank; the fourth cell carries the process' row count) */
static int fold_scope(gq_t *G, int k, int64_t *mn, int64_t *mx, int64_t *seen) {
    int64_t lo = INF_I64, hi = NULL_I64, tot = 0;
    for (int s = 0; s < G->S; s++) {
        if (G->sh[s].seen <= 0) continue;
        tot += G->sh[s].seen;
        if (G->sh[s].mn[k] < lo) lo = G->sh[s].mn[k];
        if (G->sh[s].mx[k] > hi) hi = G->sh[s].mx[k];
    }
    if (G->exch) {
        int64_t mine[4] = {lo, hi, tot, G->q->nrows}, all[4 * 256];
        if (G->world > 256) return RFX_ELIMIT;
        const int rc = xp_allgather_host(G->x, mine, 32, all);
        if (rc != RFX_OK) return rc;
        lo = INF_I64, hi = NULL_I64, tot = 0;
        int64_t before = 0, rows = 0;
        for (int r = 0; r < G->world; r++) {
            if (r < G->rank) before += all[4 * r + 3];
            rows += all[4 * r + 3];
            if (all[4 * r + 2] <= 0) continue;
            tot += all[4 * r + 2];
            if (all[4 * r] < lo) lo = all[4 * r];
            if (all[4 * r + 1] > hi) hi = all[4 * r + 1];
        }
        G->proc_row0 = before;
        G->total_rows = rows;
    }
    *mn = lo;
    *mx = hi;
    *seen = tot;
    return RFX_OK;
}
/* several keys whose ranges overflow 64 bits / a null key among them: group on the reference's own row hash */
static int ph_row_hash(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    void *hh = NULL;
    int rc = sh_malloc(G->x, h, s, &hh, (size_t)(h->nrows ? h->nrows : 1) * 8);
    if (rc != RFX_OK) return rc;
    /* value_first: the argument order the reference uses for filtered rows (core/index.c:155-175) */
    if ((rc = rfx_hip_row_hash(G->x->ctx[s], h->keys, G->nkeys, h->nrows, G->npred > 0 ? 1 : 0, (int64_t *)hh)) != RFX_OK) return rc;
    h->key = hh;
    return RFX_OK;
}
/* sparse composite: the hashed path keys on the materialised column (core/index.c:2421 -> :2092) */
static int ph_composite(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    void *comp = NULL;
    int rc = sh_malloc(G->x, h, s, &comp, (size_t)(h->nrows ? h->nrows : 1) * 8);
    if (rc != RFX_OK) return rc;
    if ((rc = rfx_hip_composite_key(G->x->ctx[s], h->keys, G->kmins, G->kmults, G->nkeys, h->nrows, (int64_t *)comp)) != RFX_OK) return rc;
    h->key = comp;
    return RFX_OK;
}

/* tables of one shard: one block, arrays of `cells` 8-byte cells (what the exchanges and the merge kernel walk) */
static int tables_alloc(gq_t *G, int s) {
    shard_t *h = &G->sh[s];
    rfx_ctx_t *c = G->x->ctx[s];
    const int64_t cells = G->dense ? (int64_t)G->range : G->cap + 1;
    if (h->store) rfx_hip_free(c, h->store);
    h->store = NULL;
    int rc = rfx_hip_malloc(c, &h->store, (size_t)G->narr * (size_t)cells * 8);
    if (rc != RFX_OK) return rc;
    int64_t *base = (int64_t *)h->store;
    int k = 0;
    memset(&h->gt, 0, sizeof(h->gt));
    memset(&h->ht, 0, sizeof(h->ht));
    if (G->dense) {
        h->gt.kmin = G->kmin;
        h->gt.range = (int64_t)G->range;
        h->gt.nagg = G->na;
        h->gt.d_first = base + (k++) * cells;
    } else {
        h->ht.capacity = G->cap;
        h->ht.nagg = G->na;
        h->ht.d_keys = base + (k++) * cells;
        h->ht.d_first = base + (k++) * cells;
    }
    for (int a = 0; a < G->na; a++) {
        void *acc = base + (k++) * cells;
        int64_t *cnt = has_cnt(&h->aggs[a]) ? base + (k++) * cells : NULL;
        if (G->dense) { h->gt.d_acc[a] = acc; h->gt.d_cnt[a] = cnt; }
        else { h->ht.d_acc[a] = acc; h->ht.d_cnt[a] = cnt; }
    }
    return G->dense ? rfx_hip_group_tables_init(c, h->aggs, &h->gt) : rfx_hip_hash_tables_init(c, h->aggs, &h->ht);
}
/* the pass: tables + one scatter-aggregate over the shard's rows.  flag: 1 = a sampled scope did not hold / a hashed table is full */
static int ph_pass(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    shard_t *h = &G->sh[s];
    rfx_ctx_t *c = G->x->ctx[s];
    h->flag = 0;
    int rc = tables_alloc(G, s);
    if (rc != RFX_OK) return rc;
    if (G->dense) {
        if (G->spec && (rc = rfx_hip_ctx_speculative(c, 1)) != RFX_OK) return rc;
        rc = h->nrows == 0 ? RFX_OK
             : G->fused_keys ? rfx_hip_group_dense_accumulate_keys(c, h->keys, G->kmins, G->kmults, G->nkeys, h->preds, G->npred, G->q->logic, h->aggs, h->nrows, h->row0, &h->gt)
                             : rfx_hip_group_dense_accumulate(c, (const int64_t *)h->key, h->preds, G->npred, G->q->logic, h->aggs, h->nrows, h->row0, &h->gt);
        if (G->spec) {
            rfx_hip_ctx_speculative(c, 0);
            if (rc == RFX_ESTATE) { /* a path that cannot report keys outside the scope: nothing ran */
                h->flag = 1;
                return RFX_OK;
            }
            if (rc != RFX_OK) return rc;
            int bad = 0;
            if (h->nrows && (rc = rfx_hip_group_out_of_scope(c, &bad)) != RFX_OK) return rc;
            h->flag = bad;
        }
        if (rc != RFX_OK) return rc;
    } else {
        rc = h->nrows == 0 ? RFX_OK : rfx_hip_group_hash_accumulate(c, (const int64_t *)h->key, h->preds, G->npred, G->q->logic, h->aggs, h->nrows, h->row0, &h->ht);
        if (rc == RFX_ELIMIT) {
            h->flag = 1;
            return RFX_OK;
        }
        if (rc != RFX_OK) return rc;
    }
    /* a phase ends when the shard's stream is idle -- what the merge needs; ONE shard goes on in stream order (a sync is ~25 us of idle device) */
    return (G->S > 1 || G->exch) ? rfx_hip_ctx_sync(c) : RFX_OK;
}
/* shards that share a device: the device's lead folds their tables into its own (kernel / re-insertion), on its own stream */
static int ph_merge_local(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    if (x->lead[s] != s) return RFX_OK;
    shard_t *h = &G->sh[s];
    h->flag = 0;
    for (int t = s + 1; t < G->S; t++) {
        if (x->lead[t] != s) continue;
        int rc = G->dense ? rfx_hip_group_tables_merge(x->ctx[s], h->aggs, &h->gt, &G->sh[t].gt) : rfx_hip_hash_tables_merge(x->ctx[s], h->aggs, &h->ht, &G->sh[t].ht);
        if (rc == RFX_ELIMIT && !G->dense) {
            h->flag = 1;
            return RFX_OK;
        }
        if (rc != RFX_OK) return rc;
        __atomic_fetch_add(&x->stat[RFX_XSTAT_MERGES_KERNEL], 1, __ATOMIC_RELAXED);
    }
    return rfx_hip_ctx_sync(x->ctx[s]);
}
/* the merged tables back to the shards that will rank / emit beside their lead (FIRST values live with the rows) */
static int ph_copy_back(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    if (x->lead[s] == s) return RFX_OK;
    const int64_t cells = G->dense ? (int64_t)G->range : G->cap + 1;
    int rc = rfx_hip_d2d(x->ctx[s], G->sh[s].store, G->sh[x->lead[s]].store, (size_t)G->narr * (size_t)cells * 8);
    return rc == RFX_OK ? rfx_hip_ctx_sync(x->ctx[s]) : rc;
}
static int ph_sync(void *arg, int s) { return rfx_hip_ctx_sync(((gq_t *)arg)->x->ctx[s]); }
/* hashed tables of several devices of THIS process: every lead gathers all of them and re-inserts the others' occupied slots */
static int ph_merge_gathered(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    if (x->lead[s] != s) return RFX_OK;
    shard_t *h = &G->sh[s];
    const int64_t cells = G->cap + 1;
    const size_t one = (size_t)G->narr * (size_t)cells * 8;
    int64_t *all = (int64_t *)h->dout; /* the gathered stores (borrowed slot) */
    int me = 0;
    for (int d = 0; d < x->ndev; d++)
        if (x->devlead[d] == s) me = d;
    h->flag = 0;
    for (int d = 0; d < x->ndev; d++) {
        if (d == me) continue;
        rfx_hash_tables_t o = h->ht;
        int64_t *base = (int64_t *)((char *)all + (size_t)d * one);
        int k = 0;
        o.d_keys = base + (k++) * cells;
        o.d_first = base + (k++) * cells;
        for (int a = 0; a < G->na; a++) {
            o.d_acc[a] = base + (k++) * cells;
            o.d_cnt[a] = has_cnt(&h->aggs[a]) ? base + (k++) * cells : NULL;
        }
        const int rc = rfx_hip_hash_tables_merge(x->ctx[s], h->aggs, &h->ht, &o);
        if (rc == RFX_ELIMIT) {
            h->flag = 1;
            break;
        }
        if (rc != RFX_OK) return rc;
    }
    return rfx_hip_ctx_sync(x->ctx[s]);
}

/* every shard's partial tables -> the merged tables, on every shard that goes on to rank / emit.  *full: a hashed merge ran out of room */
static int merge_tables(gq_t *G, int *full) {
    rfx_exec_t *x = G->x;
    int rc = RFX_OK;
    *full = 0;
    if (G->S > x->ndev) { /* shards sharing a device */
        rc = run_shards(x, ph_merge_local, G);
        if (rc != RFX_OK) return rc;
        for (int s = 0; s < G->S; s++) *full |= (x->lead[s] == s && G->sh[s].flag);
    }
    /* a hashed re-insertion that ran out of room on ONE process must stop EVERY process before the exchange below: a process that skipped
     * it alone would meet the others' collective with the next pass' (a hang under RCCL, a size mismatch under gloo).  Asked by every
     * process alike, whatever its own shard layout; dense merges never fill up and need no such agreement. */
    if (G->exch && !G->dense) {
        int any = 0;
        if ((rc = xp_any(x, G->world, *full, &any)) != RFX_OK) return rc;
        *full = any;
    }
    if (x->ndev > 1 && !x->comm_all) {
        snprintf(x->err, sizeof(x->err), "rfx_exec: several devices without communicators (rfx_exec_comm_init_all)");
        return RFX_ESTATE;
    }
    if (x->comm_all && !*full) { /* the devices of this process: ONE fused exchange over xGMI */
        rfx_ctx_t *leads[RFX_MAX_SHARDS];
        for (int d = 0; d < x->ndev; d++) leads[d] = x->ctx[x->devlead[d]];
        if (G->dense) {
            const rfx_group_tables_t *ts[RFX_MAX_SHARDS];
            for (int d = 0; d < x->ndev; d++) ts[d] = &G->sh[x->devlead[d]].gt;
            rc = rfx_dist_group_tables_allreduce_all(leads, x->ndev, G->sh[0].aggs, ts);
        } else {
            const int64_t cells = G->cap + 1;
            const size_t one = (size_t)G->narr * (size_t)cells * 8;
            const void *ins[RFX_MAX_SHARDS];
            void *outs[RFX_MAX_SHARDS];
            for (int d = 0; d < x->ndev && rc == RFX_OK; d++) {
                shard_t *h = &G->sh[x->devlead[d]];
                rc = rfx_hip_ctx_bind_thread(leads[d]);
                if (rc == RFX_OK) rc = rfx_hip_malloc(leads[d], &h->dout, one * (size_t)x->ndev);
                ins[d] = h->store;
                outs[d] = h->dout;
            }
            rfx_hip_ctx_bind_thread(x->ctx[0]);
            if (rc == RFX_OK) rc = rfx_dist_allgather_all(leads, x->ndev, ins, one, outs);
            if (rc == RFX_OK) rc = run_shards(x, ph_merge_gathered, G);
            for (int d = 0; d < x->ndev; d++) {
                shard_t *h = &G->sh[x->devlead[d]];
                if (h->dout) {
                    rfx_hip_ctx_bind_thread(leads[d]);
                    rfx_hip_free(leads[d], h->dout);
                    h->dout = NULL;
                }
                *full |= h->flag;
            }
            rfx_hip_ctx_bind_thread(x->ctx[0]);
        }
        if (rc != RFX_OK) {
            if (!x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
            return rc;
        }
        x->stat[RFX_XSTAT_MERGES_RCCL]++;
        rc = run_shards(x, ph_sync, G);
        if (rc != RFX_OK) return rc;
    }
    if (G->exch && !*full) { /* the other processes: the lead context's exchange */
        shard_t *h = &G->sh[0];
        if (G->dense) {
            if (x->has_tr) {
                /* a host transport reduces array by array; neighbours of one class as one call */
                struct { void *p; int64_t n; int type, op; } calls[1 + 2 * RFX_MAX_AGGS];
                int n = 0;
                const int64_t cells = (int64_t)G->range;
                calls[n].p = h->gt.d_first; calls[n].n = cells; calls[n].type = 0; calls[n].op = 1; n++;
                for (int a = 0; a < G->na; a++) {
                    const int f64 = rfx_agg_input_type(&h->aggs[a]) == RFX_F64, kind = h->aggs[a].kind;
                    int type = 0, op = 0;
                    if (kind == RFX_AGG_MIN) op = 1;
                    else if (kind == RFX_AGG_MAX) op = 2;
                    else if (kind == RFX_AGG_AVG || (kind == RFX_AGG_SUM && f64)) type = 1;
                    if (n && calls[n - 1].type == type && calls[n - 1].op == op && (char *)calls[n - 1].p + calls[n - 1].n * 8 == (char *)h->gt.d_acc[a]) calls[n - 1].n += cells;
                    else { calls[n].p = h->gt.d_acc[a]; calls[n].n = cells; calls[n].type = type; calls[n].op = op; n++; }
                    if (h->gt.d_cnt[a]) {
                        if (calls[n - 1].type == 0 && calls[n - 1].op == 0 && (char *)calls[n - 1].p + calls[n - 1].n * 8 == (char *)h->gt.d_cnt[a]) calls[n - 1].n += cells;
                        else { calls[n].p = h->gt.d_cnt[a]; calls[n].n = cells; calls[n].type = 0; calls[n].op = 0; n++; }
                    }
                }
                for (int i = 0; i < n && rc == RFX_OK; i++) rc = xp_allreduce(x, calls[i].p, calls[i].n, calls[i].type, calls[i].op);
            } else {
                x->stat[RFX_XSTAT_MERGES_TRANSPORT]++;
                rc = rfx_dist_group_tables_allreduce(x->ctx[0], h->aggs, &h->gt);
            }
            if (rc == RFX_OK) rc = rfx_hip_ctx_sync(x->ctx[0]);
        } else {
            const int64_t cells = G->cap + 1;
            const size_t one = (size_t)G->narr * (size_t)cells * 8;
            void *all = NULL;
            rc = rfx_hip_malloc(x->ctx[0], &all, one * (size_t)G->world);
            if (rc == RFX_OK) rc = xp_allgather_dev(x, h->store, one, all);
            for (int r = 0; r < G->world && rc == RFX_OK; r++) {
                if (r == G->rank) continue;
                rfx_hash_tables_t o = h->ht;
                int64_t *base = (int64_t *)((char *)all + (size_t)r * one);
                int k = 0;
                o.d_keys = base + (k++) * cells;
                o.d_first = base + (k++) * cells;
                for (int a = 0; a < G->na; a++) {
                    o.d_acc[a] = base + (k++) * cells;
                    o.d_cnt[a] = has_cnt(&h->aggs[a]) ? base + (k++) * cells : NULL;
                }
                rc = rfx_hip_hash_tables_merge(x->ctx[0], h->aggs, &h->ht, &o);
                if (rc == RFX_ELIMIT) {
                    *full = 1;
                    rc = RFX_OK;
                    break;
                }
            }
            if (rc == RFX_OK) rc = rfx_hip_ctx_sync(x->ctx[0]);
            if (all) rfx_hip_free(x->ctx[0], all);
            int any = 0;
            if (rc == RFX_OK) rc = xp_any(x, G->exch ? G->world : 0, *full, &any);
            *full = any;
        }
        if (rc != RFX_OK) {
            if (!x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
            return rc;
        }
    }
    /* FIRST values are read where the rows are: the merged tables go down from every device's lead to the shards beside it */
    if ((G->all_rank || G->slice_all) && !*full && G->S > x->ndev) rc = run_shards(x, ph_copy_back, G);
    return rc;
}

/* rank by first row (first-occurrence order, core/index.c:2037-2055) and emit: on the lead; on every shard when FIRST values are asked
 * for (a group's first value is read by the shard that owns its first row, the others write 0); on every SLICE OWNER of a sliced result
 * (every owner ranks the same merged tables -- redundant, and parallel -- and emits only its range of the groups) */
static int ph_rank_emit(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    const int si = G->slidx[s];
    if (si < 0 && !G->all_rank) return RFX_OK;
    shard_t *h = &G->sh[s];
    rfx_ctx_t *c = x->ctx[s];
    /* FIRST: a first row is owned by the shard whose rows [row0, row0 + nloc) hold it (nloc 0 = the one shard owns every row); a shard
     * without rows owns none (a row offset no first row reaches) */
    const int multi = G->S > 1 || G->exch;
    const int64_t nloc = multi ? (h->nrows > 0 ? h->nrows : 1) : 0, r0 = (multi && h->nrows == 0) ? INF_I64 : h->row0;
    const int nsl = G->nsl > 1 ? G->nsl : 1, sl = G->nsl > 1 ? si : 0;
    void *ptrs[RFX_MAX_AGGS];
    int rc;
    h->g0 = h->gn = 0;
    const int64_t slots = G->dense ? (int64_t)G->range : G->cap + 1;
    if (!x->two_step_rank && slots <= RFX_RANK_EMIT_MAX) {
        /* rank -> emit with no host round trip between them: the outputs are sized before the group count is known -- groups <= min(slots, selected
         * rows), a slice its share + 1 -- and the count comes back once everything is enqueued */
        int64_t bound = slots < G->seen ? slots : G->seen;
        if (bound < 1) bound = 1;
        const int64_t cap = bound / nsl + 1;
        if ((rc = rfx_hip_malloc(c, &h->dout, (size_t)(G->na + 1) * (size_t)cap * 8)) != RFX_OK) return rc;
        if ((G->want_first || !G->dense) && (rc = rfx_hip_malloc(c, &h->dfirst, (size_t)cap * 8)) != RFX_OK) return rc;
        for (int a = 0; a < G->na; a++) ptrs[a] = (int64_t *)h->dout + (size_t)(a + 1) * (size_t)cap;
        h->gstride = cap;
        rc = G->dense ? rfx_hip_group_rank_emit(c, h->aggs, &h->gt, G->total_rows, r0, nloc, nsl, sl, cap, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs, &h->groups)
                      : rfx_hip_hash_rank_emit(c, h->aggs, &h->ht, G->total_rows, r0, nloc, nsl, sl, cap, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs, &h->groups);
        if (x->timing) h->t_rank = now_ns();
        if (rc != RFX_OK) return rc;
        const int64_t g = h->groups;
        h->g0 = nsl > 1 ? RFX_SLICE_G0(g, sl, nsl) : 0;
        h->gn = nsl > 1 ? RFX_SLICE_GN(g, sl, nsl) : g;
        if (g == 0 || h->gn == 0) return RFX_OK;
    } else {
        rc = G->dense ? rfx_hip_group_rank(c, &h->gt, G->total_rows, &h->groups) : rfx_hip_hash_rank(c, &h->ht, G->total_rows, &h->groups);
        if (x->timing) h->t_rank = now_ns();
        if (rc != RFX_OK || h->groups == 0) return rc;
        const int64_t g = h->groups;
        int64_t g0 = 0, gn = g;
        if (nsl > 1) { /* this owner's range of the groups */
            g0 = RFX_SLICE_G0(g, sl, nsl);
            gn = RFX_SLICE_GN(g, sl, nsl);
        }
        h->g0 = g0;
        h->gn = gn;
        h->gstride = gn;
        if (gn == 0) return RFX_OK; /* (fewer groups than slices) */
        if ((rc = rfx_hip_malloc(c, &h->dout, (size_t)(G->na + 1) * (size_t)gn * 8)) != RFX_OK) return rc;
        if ((G->want_first || !G->dense) && (rc = rfx_hip_malloc(c, &h->dfirst, (size_t)gn * 8)) != RFX_OK) return rc;
        for (int a = 0; a < G->na; a++) ptrs[a] = (int64_t *)h->dout + (size_t)(a + 1) * (size_t)gn;
        if (nsl > 1 && (rc = rfx_hip_ctx_emit_window(c, g0, gn)) != RFX_OK) return rc;
        rc = G->dense ? rfx_hip_group_emit_sharded(c, h->aggs, &h->gt, r0, nloc, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs)
                      : rfx_hip_hash_emit_sharded(c, h->aggs, &h->ht, r0, nloc, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs);
        if (nsl > 1) rfx_hip_ctx_emit_window(c, 0, 0);
        if (rc != RFX_OK) return rc;
    }
    if (G->nsl > 1 && G->first_pass && G->nkeys > 1 && !G->rowhash) /* this slice's key columns, decoded from its composite keys (core/query.c:110-135) */
        for (int k = 0; k < G->nkeys; k++) {
            if ((rc = rfx_hip_malloc(c, &h->kc[k], (size_t)h->gn * 8)) != RFX_OK) return rc;
            if ((rc = rfx_hip_composite_decode(c, (const int64_t *)h->dout, h->gn, G->kmins[k], G->kmults[k], G->kmaxs[k] - G->kmins[k] + 1, (int64_t *)h->kc[k])) != RFX_OK) return rc;
        }
    /* FIRST values merge across the shards next: their streams must be idle.  A slice is read back on its own stream (fetch_all) and one
     * shard goes on in stream order: no wait (the tables go back to the pool of the stream that read them) */
    return (G->all_rank || x->timing) ? rfx_hip_ctx_sync(c) : RFX_OK;
}
/* FIRST columns of the shards beside a lead, added into the lead's (exactly one shard wrote each value) */
static int ph_first_local(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    if (x->lead[s] != s) return RFX_OK;
    shard_t *h = &G->sh[s];
    const int64_t g = h->groups;
    for (int t = s + 1; t < G->S; t++) {
        if (x->lead[t] != s) continue;
        for (int a = 0; a < G->na; a++) {
            if (h->aggs[a].kind != RFX_AGG_FIRST) continue;
            const int rc = rfx_hip_add_i64(x->ctx[s], (int64_t *)h->dout + (size_t)(a + 1) * (size_t)h->gstride, (const int64_t *)G->sh[t].dout + (size_t)(a + 1) * (size_t)G->sh[t].gstride, g);
            if (rc != RFX_OK) return rc;
        }
    }
    return rfx_hip_ctx_sync(x->ctx[s]);
}
static int merge_first_values(gq_t *G) {
    rfx_exec_t *x = G->x;
    const int64_t g = G->sh[0].groups;
    int rc = RFX_OK;
    if (g == 0) return RFX_OK;
    if (G->S > x->ndev) rc = run_shards(x, ph_first_local, G);
    for (int a = 0; a < G->na && rc == RFX_OK; a++) {
        if (G->sh[0].aggs[a].kind != RFX_AGG_FIRST) continue;
        if (x->comm_all) {
            rfx_ctx_t *leads[RFX_MAX_SHARDS];
            int64_t *bufs[RFX_MAX_SHARDS];
            for (int d = 0; d < x->ndev; d++) {
                leads[d] = x->ctx[x->devlead[d]];
                bufs[d] = (int64_t *)G->sh[x->devlead[d]].dout + (size_t)(a + 1) * (size_t)G->sh[x->devlead[d]].gstride;
            }
            rc = rfx_dist_allreduce_i64_all(leads, x->ndev, bufs, g, 0);
            x->stat[RFX_XSTAT_MERGES_RCCL]++;
        }
        if (rc == RFX_OK && G->exch) rc = xp_allreduce(x, (int64_t *)G->sh[0].dout + (size_t)(a + 1) * (size_t)G->sh[0].gstride, g, 0, 0);
    }
    if (rc == RFX_OK && (x->comm_all || G->exch)) rc = run_shards(x, ph_sync, G);
    if (rc != RFX_OK && !x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
    return rc;
}

static int spec_known_bad(rfx_exec_t *x, const void *key, int64_t n) {
    for (int i = 0; i < x->nspec_failed; i++)
        if (x->spec_failed[i] == key && x->spec_failed_n[i] == n) return 1;
    return 0;
}
static void spec_remember_bad(rfx_exec_t *x, const void *key, int64_t n) {
    const int i = x->spec_ring++ % 32;
    x->spec_failed[i] = key;
    x->spec_failed_n[i] = n;
    if (x->nspec_failed < 32) x->nspec_failed++;
}
static int spec_known_wide(rfx_exec_t *x, const void *key, int64_t n) {
    for (int i = 0; i < x->nspec_wide; i++)
        if (x->spec_wide[i] == key && x->spec_wide_n[i] == n) return 1;
    return 0;
}
static void spec_remember_wide(rfx_exec_t *x, const void *key, int64_t n) {
    if (spec_known_wide(x, key, n)) return;
    const int i = x->wide_ring++ % 32;
    x->spec_wide[i] = key;
    x->spec_wide_n[i] = n;
    if (x->nspec_wide < 32) x->nspec_wide++;
}
static void own_on(rfx_groups_t *g, void *p, int shard) {
    if (p && g->nown < (int)(sizeof(g->own) / sizeof(g->own[0]))) {
        g->own_shard[g->nown] = (int8_t)shard;
        g->own[g->nown++] = p;
    }
}
static void own(rfx_groups_t *g, void *p) { own_on(g, p, 0); }

/* ---- one pass of a group-by (aggregates [a0, a0 + na) of the query; the first pass also makes the key columns / first rows), step by step:
 * gb_setup -> { gb_scope -> gb_size -> gb_passes } (once more under the exact scope when the sampled one did not hold) -> gb_null_slot ->
 * gb_prove_tuples -> gb_emit_small | gb_emit.  Every step answers RFX_OK or an error (x->err says which); group_by_pass owns the cleanup. ---- */
#define GB_AGAIN 2 /* gb_passes: a key outside the sampled scope -- the scope again, exactly, then the passes again */

/* the shards' views of the query, a mask selection gathered, xbar keys bucketed; whether the scope may be sampled */
static int gb_setup(gq_t *G) {
    rfx_exec_t *x = G->x;
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int S = G->S;
    const int na = G->na;
    int rc = RFX_OK;
    for (int s = 0; s < S && rc == RFX_OK; s++) {
        rc = shard_view(q, S, s, G->a0, na, &sh[s]);
        rfx_exec_split(q->nrows, S, s, &sh[s].row0, &sh[s].nrows);
    }
    if (rc != RFX_OK) { snprintf(x->err, sizeof(x->err), "rfx_exec: a column of the query has no per-shard address"); return rc; }
    for (int a = 0; a < na; a++) G->need_first_values |= sh[0].aggs[a].kind == RFX_AGG_FIRST;
    G->all_rank = G->need_first_values && (S > 1 || G->exch);
    G->multi = S > 1 || G->exch;
    /* the tail, sharded: one slice of the groups per device (its lead ranks, emits and -- rfx_exec_groups_fetch_all -- reads it back); FIRST
     * values live with the rows and keep the every-shard emit + SUM merge on the lead */
    G->nsl = 1;
    G->slown[0] = 0;
    G->slice_all = 0;
    /* (decided for the QUERY, not for this pass's chunk of the aggregates: every pass of one query leaves its columns the same way) */
    int any_first = 0;
    for (int a = 0; a < q->nagg; a++) any_first |= q->aggs[a].kind == RFX_AGG_FIRST;
    if ((q->flags & RFX_Q_SLICED) && !any_first && !q->d_mask && S > 1) {
        if (x->slice_shards) {
            G->nsl = S;
            for (int s = 0; s < S; s++) G->slown[s] = s;
            G->slice_all = S > x->ndev;
        } else if (x->ndev > 1) {
            G->nsl = x->ndev;
            for (int d = 0; d < x->ndev; d++) G->slown[d] = x->devlead[d];
        }
    }
    for (int s = 0; s < S; s++) G->slidx[s] = -1;
    for (int i = 0; i < G->nsl; i++) G->slidx[G->slown[i]] = i;
    const int multi = G->multi;
    if (q->d_mask) {
        if (multi || q->npred) { snprintf(x->err, sizeof(x->err), "rfx_exec: a mask selection runs on one shard, without comparisons beside it"); rc = RFX_ELIMIT; return rc; }
        if ((rc = gather_selected(x, &sh[0], na, q->nkeys)) != RFX_OK) { snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error()); return rc; }
        G->npred = 0;
        G->total_rows = sh[0].nrows; /* first rows rank among the SELECTED rows; translated back at the end */
    }
    for (int k = 0; k < q->nkeys; k++) G->any_xbar |= q->kxbar && q->kxbar[k] > 0;
    if (G->any_xbar && (rc = run_shards(x, ph_xbar, G)) != RFX_OK) return rc;
    /* ---- the scope ---- */
    G->spec_id = q->d_keys[0];
    /* (one process only: whether to sample must be decided alike by every process, and row counts / remembered misses are local) */
    G->spec_ok = !(q->flags & RFX_Q_NO_SAMPLED_SCOPE) && !q->d_mask && !G->exch && q->nrows >= ((int64_t)1 << 24) && !getenv("RFX_NO_SAMPLED_SCOPE") &&
                  !spec_known_bad(x, G->spec_id, q->nrows);
    G->retried = 0;
    return RFX_OK;
}

/* the scope: sampled / remembered / exact, one key or several (composite plan, row hash); decides dense vs. hashed */
static int gb_scope(gq_t *G) {
    rfx_exec_t *x = G->x;
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int S = G->S;
    const int multi = G->multi;
    const int any_xbar = G->any_xbar;
    rfx_groups_t *out = G->out;
    int rc = RFX_OK;
    for (;;) {
        G->spec = G->rowhash = G->fused_keys = G->sparse_sampled = 0;
        for (int s = 0; s < S; s++) sh[s].key = sh[s].keys[0];
        if (G->nkeys == 1) {
            int have = 0;
            /* (a DENSE key range the sample found wider than the LDS forms last time takes the scope pass at once: that pass samples for itself) */
            if (G->spec_ok && !(any_xbar == 0 && spec_known_wide(x, G->spec_id, q->nrows))) {
                if ((rc = run_shards(x, ph_scope_sample, G)) != RFX_OK) return rc;
                if ((rc = fold_scope(G, 0, &G->kmin, &G->kmax, &G->seen)) != RFX_OK) return rc;
                G->spec = G->seen > 0 && G->kmin != NULL_I64 && G->kmax >= G->kmin && (uint64_t)(G->kmax - G->kmin) < RFX_SCOPE_SAMPLE_MAX_RANGE;
                have = G->spec;
                /* SPARSE keys by the sample alone: a sampled range can only be too small, so one that already exceeds the row count decides
                 * "range > rows" -- open addressing (core/index.c:2013) -- without index_scope_i64's pass over the column (8 GB per 1e9 rows, a
                 * twentieth of such a query): the tables are sized by the row count as the reference sizes them, and a null key the sample did
                 * not see shows in the tables' own null slot afterwards */
                if (!have && G->seen > 0 && G->kmin != NULL_I64 && G->kmax >= G->kmin && (uint64_t)G->kmax - (uint64_t)G->kmin + 1 > (uint64_t)q->nrows) {
                    G->seen = q->nrows;
                    G->sparse_sampled = 1;
                    have = 1;
                }
                if (!have && !any_xbar && G->kmin != NULL_I64) spec_remember_wide(x, G->spec_id, q->nrows); /* neither LDS-sized nor sparse: dense and wide */
            }
            if (!have && q->key_scope && !any_xbar && !q->d_mask && !G->exch && q->key_scope[0] != NULL_I64 && q->key_scope[1] >= q->key_scope[0] &&
                (uint64_t)(q->key_scope[1] - q->key_scope[0]) < RFX_SCOPE_SAMPLE_MAX_RANGE) {
                /* the caller's remembered whole-column scope: LDS-sized, a superset of any selection's -- `seen` = every row (an upper bound
                 * that only sizes tables; an empty selection comes out as zero groups) */
                G->kmin = q->key_scope[0];
                G->kmax = q->key_scope[1];
                G->seen = q->nrows;
                have = 1;
                x->stat[RFX_XSTAT_SCOPE_REMEMBERED]++;
            }
            if (!have) {
                if ((rc = run_shards(x, ph_scope_group, G)) != RFX_OK) return rc;
                if ((rc = fold_scope(G, 0, &G->kmin, &G->kmax, &G->seen)) != RFX_OK) return rc;
            }
        } else {
            /* scopes of every key column, then the reference's multiplier plan (core/index.c:2340-2383) */
            int planned = 0;
            if (G->spec_ok) {
                if ((rc = run_shards(x, ph_scope_sample, G)) != RFX_OK) return rc;
                int64_t prod = 1;
                G->spec = 1;
                for (int k = 0; k < G->nkeys && G->spec; k++) {
                    if ((rc = fold_scope(G, k, &G->kmins[k], &G->kmaxs[k], &G->seen)) != RFX_OK) return rc;
                    if (G->seen <= 0 || G->kmins[k] == NULL_I64 || G->kmaxs[k] < G->kmins[k] || (uint64_t)(G->kmaxs[k] - G->kmins[k]) >= RFX_SCOPE_SAMPLE_MAX_RANGE) G->spec = 0;
                    else prod *= G->kmaxs[k] - G->kmins[k] + 1;
                    if (prod > RFX_SCOPE_SAMPLE_MAX_RANGE) G->spec = 0;
                }
                planned = G->spec;
            }
            if (!planned) {
                G->scope_filtered = G->npred > 0;
                for (int k = 0; k < G->nkeys; k++) {
                    G->phase_key = k;
                    if ((rc = run_shards(x, ph_scope_col, G)) != RFX_OK) return rc;
                    if ((rc = fold_scope(G, k, &G->kmins[k], &G->kmaxs[k], &G->seen)) != RFX_OK) return rc;
                }
            }
            G->kmin = 0;
            G->kmax = -1;
            if (G->seen > 0) {
                if (rfx_composite_plan(G->kmins, G->kmaxs, G->nkeys, G->kmults, &G->comp_max) != RFX_OK) {
                    /* ranges beyond 64 bits / a null key: the reference's row-hash path (index_group_list, core/index.c:2731-2790) -- grouped on
                     * the reference's own row hash; its tuple comparison on every probe is made once, afterwards (below) */
                    /* over several shards / processes the tuple proof is made by aggregates (rfx_exec_group_by: a MIN and a MAX per key column ride
                     * through the same merge; one hash = one tuple iff they agree) -- which skip nulls: a null key rides as max + 1 there and comes back as the null */
                    for (int k = 0; k < G->nkeys; k++) {
                        x->rh_kmin[k] = G->kmins[k];
                        x->rh_kmax[k] = G->kmaxs[k];
                        if (multi && G->kmins[k] == NULL_I64 && G->kmaxs[k] == INT64_MAX) { /* no value left to stand in for the null */
                            snprintf(x->err, sizeof(x->err), "rfx_exec: key tuples with a null key beside INT64_MAX run on one shard");
                            rc = RFX_ELIMIT;
                            return rc;
                        }
                    }
                    if ((rc = run_shards(x, ph_row_hash, G)) != RFX_OK) return rc;
                    G->rowhash = 1;
                    G->spec = 0;
                    G->phase_key = -1;
                    G->scope_filtered = G->npred > 0;
                    if ((rc = run_shards(x, ph_scope_col, G)) != RFX_OK) return rc;
                    if ((rc = fold_scope(G, 0, &G->kmin, &G->kmax, &G->seen)) != RFX_OK) return rc;
                } else {
                    G->kmax = G->comp_max; /* forced scope {0, max}, core/index.c:2421 */
                    if ((uint64_t)G->comp_max + 1 > (uint64_t)G->seen) {
                        if ((rc = run_shards(x, ph_composite, G)) != RFX_OK) return rc;
                    } else G->fused_keys = 1;
                }
            }
        }
        if (G->seen > 0 && G->nkeys == 1 && G->kmin == NULL_I64 && (q->flags & RFX_Q_REFUSE_NULL_KEY)) { rc = RFX_EXEC_NULL_KEY; return rc; }
        if (G->spec) x->stat[RFX_XSTAT_SCOPE_SAMPLED]++;
        out->nkeys = G->nkeys;
        if (G->seen <= 0) return RFX_OK; /* nothing selected: zero groups (group_by_pass says so) */
        /* dense "perfect hash" iff range <= rows (core/index.c:2013), like the reference; else open addressing */
        G->range = (uint64_t)G->kmax - (uint64_t)G->kmin + 1;
        G->dense = G->range != 0 && G->range <= (uint64_t)G->seen && G->kmin != NULL_I64 && !G->rowhash;
        if (G->rowhash) G->dense = 0;
        if (G->spec && !G->dense) { /* (not a miss of the sample: nothing to remember) */
            G->spec_ok = 0;
            continue;
        }
        return RFX_OK;
    }
}

/* table sizes, the small-range form, global row ids under an exchange */
static void gb_size(gq_t *G) {
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int S = G->S, na = G->na, multi = G->multi;
    rfx_hip_group_table_arrays(sh[0].aggs, na, &G->narr);
    G->cap = G->cap_max = 16;
    if (!G->dense) {
        /* the reference sizes its table by the row count (ht_oa_create(len), core/index.c:1805); the distinct keys are usually far fewer:
         * start at 4 M slots and take the reference's size when a pass reports the table full */
        while (G->cap_max < 2 * G->seen) G->cap_max <<= 1;
        G->cap = G->cap_max < (1 << 22) ? G->cap_max : (1 << 22);
        if (G->cap_hint > G->cap && G->cap_hint <= G->cap_max) G->cap = G->cap_hint;
        G->narr += 1;
        G->fused_keys = 0;
    }
    G->small = G->dense && G->nkeys == 1 && G->range <= RFX_RANK_SMALL && !multi && !(q->flags & RFX_Q_NO_SMALL) && !q->d_mask;
    if (G->exch) /* global row ids: this process' rows come after the lower ranks' (known since the scope exchange) */
        for (int s = 0; s < S; s++) {
            int64_t r0;
            rfx_exec_split(q->nrows, S, s, &r0, NULL);
            sh[s].row0 = G->proc_row0 + r0;
        }
}

/* the passes over the shards and the merge of their tables; a full hashed table grows (every shard and process together) and runs again */
static int gb_passes(gq_t *G) {
    rfx_exec_t *x = G->x;
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int S = G->S;
    const int multi = G->multi;
    int rc = RFX_OK;
    for (;;) {
        {
            T_BEGIN(x);
            rc = run_shards(x, ph_pass, G);
            if (rc == RFX_OK && x->timing && !multi) rc = rfx_hip_ctx_sync(x->ctx[0]);
            T_END(x, RFX_XSTAT_NS_PASS);
            if (rc != RFX_OK) return rc;
        }
        {
            int flag = 0, any = 0;
            for (int s = 0; s < S; s++) flag |= sh[s].flag;
            /* (a dense pass under an exact scope has nothing to report: no exchange for it) */
            if ((rc = xp_any(x, (G->exch && (G->spec || !G->dense)) ? G->world : 0, flag, &any)) != RFX_OK) return rc;
            if (any && G->dense) { /* the sampled scope did not hold somewhere: the exact scope, and the pass again */
                G->spec_ok = 0;
                if (!G->retried) {
                    G->retried = 1;
                    x->stat[RFX_XSTAT_SCOPE_RETRIED]++;
                    spec_remember_bad(x, G->spec_id, q->nrows);
                }
                return GB_AGAIN;
            }
            int full = any;
            if (!full && multi) {
                T_BEGIN(x);
                rc = merge_tables(G, &full);
                T_END(x, RFX_XSTAT_NS_MERGE);
                if (rc != RFX_OK) return rc;
            }
            if (full) { /* table full (a pass gives up at 3/4 load, early): every shard and process grows together */
                if (G->cap >= G->cap_max) { snprintf(x->err, sizeof(x->err), "rfx_exec: the hashed group table is full at the reference's own size"); rc = RFX_ELIMIT; return rc; }
                G->cap = G->cap_max;
                x->stat[RFX_XSTAT_HASH_GROWN]++;
                continue;
            }
        }
        return RFX_OK;
    }
}

/* sparse keys routed by the sample alone: did a null key come by after all?  its slot is the tables' last */
static int gb_null_slot(gq_t *G) {
    rfx_exec_t *x = G->x;
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int S = G->S;
    int rc = RFX_OK;
    if (G->sparse_sampled && !G->dense && (q->flags & RFX_Q_REFUSE_NULL_KEY)) {
        int null_seen = 0;
        for (int s = 0; s < S && rc == RFX_OK; s++) {
            int64_t f = INF_I64;
            if (S > 1) rfx_hip_ctx_bind_thread(x->ctx[s]);
            rc = rfx_hip_d2h(x->ctx[s], &f, sh[s].ht.d_first + G->cap, 8);
            null_seen |= f != INF_I64;
        }
        if (S > 1) rfx_hip_ctx_bind_thread(x->ctx[0]);
        if (rc != RFX_OK) { snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error()); return rc; }
        if (null_seen) { rc = RFX_EXEC_NULL_KEY; return rc; }
    }
    return RFX_OK;
}

/* ---- one hash = one tuple?  Every row's group-first row (the join probe against the group-by's own table), then per key column:
 * the column gathered at those rows must equal the column itself (K1 counts the rows where it does not) ---- */
static int gb_prove_tuples(gq_t *G) {
    rfx_exec_t *x = G->x;
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int multi = G->multi;
    rfx_groups_t *out = G->out;
    int rc = RFX_OK;
    if (!G->rowhash && !(q->flags & RFX_Q_PROBE_FIRST)) return RFX_OK;
    shard_t *h = &sh[0];
    rfx_ctx_t *c = x->ctx[0];
    if (G->dense || multi) { if (q->flags & RFX_Q_PROBE_FIRST) { rc = RFX_ESTATE; snprintf(x->err, sizeof(x->err), "rfx_exec: a first-row probe needs the hashed path on one shard"); return rc; } }
    else {
        void *ids = NULL, *chk = NULL;
        rc = rfx_hip_malloc(c, &ids, (size_t)(h->nrows ? h->nrows : 1) * 8);
        if (rc == RFX_OK) rc = rfx_hip_join_probe_hash(c, (const int64_t *)h->key, h->nrows, &h->ht, (int64_t *)ids);
        int collision = 0;
        if (rc == RFX_OK && G->rowhash) rc = rfx_hip_malloc(c, &chk, (size_t)(h->nrows ? h->nrows : 1) * 8);
        for (int k = 0; k < G->nkeys && rc == RFX_OK && G->rowhash && !collision; k++) {
            rfx_pred_t ne;
            rfx_value_t cv;
            int64_t differ = 0;
            memset(&ne, 0, sizeof(ne));
            ne.d_col = chk;
            ne.col_type = RFX_I64;
            ne.op = RFX_NE;
            ne.d_rhs_col = h->keys[k];
            ne.rhs_type = RFX_I64;
            rc = rfx_hip_gather_or(c, h->keys[k], h->keys[k], (const int64_t *)ids, h->nrows, 0, chk);
            if (rc == RFX_OK) rc = rfx_hip_filter_aggr_host(c, &ne, 1, RFX_AND, NULL, 0, h->nrows, &cv, &differ);
            if (rc == RFX_OK && differ) collision = 1;
        }
        if (chk) rfx_hip_free(c, chk);
        if ((q->flags & RFX_Q_PROBE_FIRST) && rc == RFX_OK && !collision && G->first_pass) {
            out->d_probe = (int64_t *)ids;
            own(out, ids);
        } else if (ids) rfx_hip_free(c, ids);
        if (rc == RFX_OK && collision) {
            snprintf(x->err, sizeof(x->err), "row-hash collision between two key tuples");
            rc = RFX_ESTATE;
        }
        if (rc != RFX_OK) { if (!x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error()); return rc; }
    }
    return RFX_OK;
}

/* few slots: rank + emit are ONE launch and the result block comes back in one copy -- the only host round trip after the pass */
static int gb_emit_small(gq_t *G) {
    rfx_exec_t *x = G->x;
    shard_t *sh = G->sh;
    const int na = G->na;
    rfx_groups_t *out = G->out;
    int rc = RFX_OK;
    shard_t *h = &sh[0];
    rfx_ctx_t *c = x->ctx[0];
    const size_t bcells = 1 + (size_t)(2 + na) * (size_t)G->range;
    void *blk = NULL;
    int64_t *mirror = (int64_t *)malloc(bcells * 8);
    rc = mirror ? rfx_hip_malloc(c, &blk, bcells * 8) : RFX_ENOMEM;
    if (rc == RFX_OK) rc = rfx_hip_group_rank_emit_small(c, h->aggs, &h->gt, 0, 0, (int64_t *)blk);
    if (rc == RFX_OK) rc = rfx_hip_d2h(c, mirror, blk, bcells * 8);
    if (rc != RFX_OK) {
        free(mirror);
        if (blk) rfx_hip_free(c, blk);
        snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
        return rc;
    }
    if (G->first_pass) {
        out->groups = mirror[0];
        out->path = RFX_PATH_DENSE_SMALL;
        out->d_block = (const char *)blk;
        out->h_block = (const char *)mirror;
        out->block_bytes = bcells * 8;
        out->d_keys = (int64_t *)blk + 1;
        out->d_first = (int64_t *)blk + 1 + G->range;
        own(out, blk);
    }
    if (!G->first_pass) { /* a later pass of a long output list: its own block, no mirror (fetched through the device) */
        free(mirror);
        own(out, blk);
    }
    for (int a = 0; a < na; a++) out->d_results[G->a0 + a] = (int64_t *)blk + 1 + (size_t)(2 + a) * (size_t)G->range;
    return RFX_OK;
}

/* rank + emit on every shard, first values merged across them, the result's key columns */
static int gb_emit(gq_t *G) {
    rfx_exec_t *x = G->x;
    const rfx_query_t *q = G->q;
    shard_t *sh = G->sh;
    const int na = G->na;
    rfx_groups_t *out = G->out;
    int rc = RFX_OK;
    const int64_t t_emit0 = x->timing ? now_ns() : 0;
    if ((rc = run_shards(x, ph_rank_emit, G)) != RFX_OK) return rc;
    if (x->timing) { /* rank = to the last shard's ranking done; emit = the rest of the phase (+ FIRST values, key columns below) */
        int64_t tr = t_emit0;
        for (int s = 0; s < G->S; s++)
            if ((G->slidx[s] >= 0 || G->all_rank) && sh[s].t_rank > tr) tr = sh[s].t_rank;
        x->stat[RFX_XSTAT_NS_RANK] += tr - t_emit0;
        x->stat[RFX_XSTAT_NS_EMIT] += now_ns() - tr;
    }
    T_BEGIN(x);
    if (G->all_rank && (rc = merge_first_values(G)) != RFX_OK) return rc;
    rfx_hip_ctx_bind_thread(x->ctx[0]);
    if (G->nsl > 1) { /* a sliced result: every owner's pieces, in group order */
        const int64_t g = sh[G->slown[0]].groups;
        for (int i = 1; i < G->nsl; i++)
            if (sh[G->slown[i]].groups != g) { snprintf(x->err, sizeof(x->err), "rfx_exec: the devices disagree on the groups of the merged tables"); return RFX_ESTATE; }
        G->groups = g;
        if (G->first_pass) out->groups = g;
        else if (out->groups != g) { snprintf(x->err, sizeof(x->err), "rfx_exec: two passes of one query disagree on the groups"); return RFX_ESTATE; }
        if (G->first_pass) out->nslices = G->nsl;
        for (int i = 0; i < G->nsl && g > 0; i++) {
            const int s = G->slown[i];
            shard_t *h = &sh[s];
            struct rfx_gslice *sl = &out->slice[i];
            sl->shard = s;
            sl->g0 = h->g0;
            sl->n = h->gn;
            for (int a = 0; a < na; a++) sl->d_results[G->a0 + a] = h->gn ? (int64_t *)h->dout + (size_t)(a + 1) * (size_t)h->gstride : NULL;
            own_on(out, h->dout, s);
            if (G->first_pass) {
                sl->d_keys = (int64_t *)h->dout;
                sl->d_first = (int64_t *)h->dfirst;
                own_on(out, h->dfirst, s);
                for (int k = 0; k < G->nkeys && G->nkeys > 1; k++) {
                    sl->d_keycols[k] = (int64_t *)h->kc[k];
                    own_on(out, h->kc[k], s);
                    h->kc[k] = NULL;
                }
            } else if (h->dfirst) rfx_hip_free(x->ctx[s], h->dfirst);
            h->dout = h->dfirst = NULL; /* the result owns them now */
        }
        if (g > 0) { /* the column pointers a caller names columns by: slice 0's */
            for (int a = 0; a < na; a++) out->d_results[G->a0 + a] = out->slice[0].d_results[G->a0 + a];
            if (G->first_pass) {
                out->d_keys = out->slice[0].d_keys;
                out->d_first = out->slice[0].d_first;
                for (int k = 0; k < G->nkeys && G->nkeys > 1; k++) out->d_keycols[k] = out->slice[0].d_keycols[k];
            }
            if (G->first_pass) x->stat[RFX_XSTAT_SLICED]++;
        }
        T_END(x, RFX_XSTAT_NS_EMIT);
        return RFX_OK;
    }
    {
        shard_t *h = &sh[0];
        rfx_ctx_t *c = x->ctx[0];
        const int64_t g = h->groups;
        G->groups = g;
        if (G->first_pass) out->groups = g;
        else if (out->groups != g) { snprintf(x->err, sizeof(x->err), "rfx_exec: two passes of one query disagree on the groups"); rc = RFX_ESTATE; return rc; }
        if (g > 0) {
            for (int a = 0; a < na; a++) out->d_results[G->a0 + a] = (int64_t *)h->dout + (size_t)(a + 1) * (size_t)h->gstride;
            own(out, h->dout);
            if (G->first_pass) {
                out->d_keys = (int64_t *)h->dout;
                /* several keys: the result's key columns -- decoded from the composite key (key_i = min_i + (composite / mult_i) % range_i
                 * = key_i[first row], core/query.c:110-135) or, on the row-hash path, gathered at the groups' first rows */
                /* (row hash over several shards: the proof passes of rfx_exec_group_by bring the key columns -- no shard holds every first row) */
                for (int k = 0; k < G->nkeys && G->nkeys > 1 && !(G->rowhash && G->multi) && rc == RFX_OK; k++) {
                    void *cell = NULL;
                    rc = rfx_hip_malloc(c, &cell, (size_t)g * 8);
                    if (rc != RFX_OK) break;
                    own(out, cell);
                    out->d_keycols[k] = (int64_t *)cell;
                    rc = G->rowhash ? rfx_hip_gather(c, h->keys[k], (const int64_t *)h->dfirst, g, cell)
                                    : rfx_hip_composite_decode(c, (const int64_t *)h->dout, g, G->kmins[k], G->kmults[k], G->kmaxs[k] - G->kmins[k] + 1, (int64_t *)cell);
                }
                if (rc == RFX_OK && q->d_mask && h->dfirst) { /* first rows among the SELECTED rows -> rows of the table */
                    void *tr = NULL;
                    rc = rfx_hip_malloc(c, &tr, (size_t)g * 8);
                    if (rc == RFX_OK) rc = rfx_hip_gather(c, h->sel_ids, (const int64_t *)h->dfirst, g, tr);
                    if (rc == RFX_OK) rc = rfx_hip_ctx_sync(c); /* (the old block goes back to the pool) */
                    if (rc == RFX_OK) {
                        rfx_hip_free(c, h->dfirst);
                        h->dfirst = tr;
                    } else if (tr) rfx_hip_free(c, tr);
                }
                out->d_first = (int64_t *)h->dfirst;
                own(out, h->dfirst);
                if (rc == RFX_OK) rc = rfx_hip_ctx_sync(c);
                if (rc != RFX_OK) { snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error()); h->dout = h->dfirst = NULL; return rc; }
            } else if (h->dfirst) rfx_hip_free(c, h->dfirst);
            h->dout = h->dfirst = NULL; /* the result owns them now */
        }
    }
    T_END(x, RFX_XSTAT_NS_EMIT);
    return RFX_OK;
}

static int group_by_pass(rfx_exec_t *x, const rfx_query_t *q, int a0, int na, int first_pass, int64_t cap_hint, rfx_groups_t *out) {
    const int S = x->nshards;
    gq_t *G = (gq_t *)calloc(1, sizeof(gq_t));
    shard_t *sh = (shard_t *)calloc((size_t)S, sizeof(shard_t));
    if (!G || !sh) { free(G); free(sh); return RFX_ENOMEM; }
    G->x = x;
    G->q = q;
    G->sh = sh;
    G->S = S;
    G->exch = world_rank(x, &G->world, &G->rank);
    G->a0 = a0;
    G->na = na;
    G->first_pass = first_pass;
    G->cap_hint = cap_hint;
    G->out = out;
    G->npred = q->npred;
    G->nkeys = q->nkeys;
    G->total_rows = q->nrows;
    G->want_first = (q->flags & RFX_Q_WANT_FIRST) != 0;
    int rc = gb_setup(G);
    while (rc == RFX_OK) {
        {
            T_BEGIN(x);
            rc = gb_scope(G);
            T_END(x, RFX_XSTAT_NS_SCOPE);
        }
        if (rc != RFX_OK || G->seen <= 0) break;
        gb_size(G);
        rc = gb_passes(G);
        if (rc != GB_AGAIN) break;
        rc = RFX_OK;
    }
    if (rc == RFX_OK && G->seen <= 0) out->groups = 0;
    else if (rc == RFX_OK) {
        rfx_hip_ctx_bind_thread(x->ctx[0]);
        rc = gb_null_slot(G);
        if (rc == RFX_OK) {
            out->path = G->rowhash ? RFX_PATH_ROWHASH : (G->dense ? RFX_PATH_DENSE : RFX_PATH_HASH);
            out->capacity = G->dense ? 0 : G->cap;
            rc = gb_prove_tuples(G);
        }
        if (rc == RFX_OK) rc = G->small ? gb_emit_small(G) : gb_emit(G);
        for (int a = 0; a < na && rc == RFX_OK; a++) {
            const rfx_agg_t *g = &sh[0].aggs[a];
            out->result_type[a0 + a] = g->kind == RFX_AGG_AVG ? RFX_F64 : (g->kind == RFX_AGG_COUNT ? RFX_I64 : rfx_agg_input_type(g));
        }
    }
    for (int s = 0; s < S; s++) {
        if (S > 1) rfx_hip_ctx_bind_thread(x->ctx[s]);
        sh_release(x, &sh[s], s);
    }
    if (S > 1) rfx_hip_ctx_bind_thread(x->ctx[0]);
    free(sh);
    free(G);
    return rc;
}

static int world_is_multi(rfx_exec_t *x) {
    int w, r;
    return world_rank(x, &w, &r);
}
/* Key tuples grouped on their row hash over SEVERAL shards: one hash = one tuple?  On one shard every row is compared with its group's first
 * row (gb_prove_tuples); across shards no row id leaves its shard, so every key column rides through the group-by once more as a (MIN, MAX)
 * pair -- the same hashed tables, the same merge, the same group order and slices -- and a group whose rows agree on every key column
 * (min == max) is exactly one tuple; the maxima ARE the result's key columns.  A disagreement is a 64-bit hash collision between two
 * tuples (probability ~ groups^2 / 2^65): RFX_ESTATE "collision", nothing is answered.  (index_group_list's __index_list_cmp_row, made
 * once per group instead of on every probe: core/index.c:2731-2790.) */
static int rowhash_proof_passes(rfx_exec_t *x, const rfx_query_t *q, int64_t cap, rfx_groups_t *out) {
    int rc = RFX_OK;
    const int S = x->nshards;
    for (int k0 = 0; k0 < q->nkeys && rc == RFX_OK; k0 += RFX_MAX_AGGS / 2) {
        const int nk = q->nkeys - k0 < RFX_MAX_AGGS / 2 ? q->nkeys - k0 : RFX_MAX_AGGS / 2;
        rfx_agg_t pa[RFX_MAX_AGGS];
        /* MIN / MAX skip nulls: a key column with nulls rides as a copy whose nulls read max + 1 (no key has it), shard by shard */
        void *tmp[RFX_MAX_AGGS / 2][RFX_MAX_SHARDS];
        int64_t repl[RFX_MAX_AGGS / 2];
        int nnull = 0;
        rfx_qcol_t *cols2 = NULL;
        memset(pa, 0, sizeof(pa));
        memset(tmp, 0, sizeof(tmp));
        for (int j = 0; j < nk; j++) nnull += x->rh_kmin[k0 + j] == NULL_I64;
        if (nnull) {
            cols2 = (rfx_qcol_t *)calloc((size_t)(q->ncols + nk), sizeof(*cols2));
            if (!cols2) return RFX_ENOMEM;
            if (q->ncols) memcpy(cols2, q->cols, (size_t)q->ncols * sizeof(*cols2));
        }
        int nc2 = q->ncols;
        for (int j = 0; j < nk; j++) {
            const void *kcol = q->d_keys[k0 + j];
            if (x->rh_kmin[k0 + j] == NULL_I64) {
                repl[j] = x->rh_kmax[k0 + j] == NULL_I64 ? 0 : x->rh_kmax[k0 + j] + 1;
                for (int s = 0; s < S && rc == RFX_OK; s++) {
                    int64_t r0, len;
                    int bad = 0;
                    rfx_exec_split(q->nrows, S, s, &r0, &len);
                    const void *src = xlate(q, s, kcol, &bad);
                    if (bad) { rc = RFX_EINVAL; break; }
                    rfx_hip_ctx_bind_thread(x->ctx[s]);
                    rc = rfx_hip_malloc(x->ctx[s], &tmp[j][s], (size_t)(len > 0 ? len : 1) * 8);
                    if (rc == RFX_OK) rc = rfx_hip_replace_i64(x->ctx[s], (const int64_t *)src, len, NULL_I64, repl[j], (int64_t *)tmp[j][s]);
                }
                rfx_hip_ctx_bind_thread(x->ctx[0]);
                if (rc != RFX_OK) break;
                for (int s = 0; s < S; s++) cols2[nc2].d[s] = tmp[j][s];
                nc2++;
                kcol = tmp[j][0];
            }
            pa[2 * j].kind = RFX_AGG_MIN;
            pa[2 * j + 1].kind = RFX_AGG_MAX;
            pa[2 * j].d_col = pa[2 * j + 1].d_col = kcol;
            pa[2 * j].col_type = pa[2 * j + 1].col_type = RFX_I64;
        }
        rfx_query_t q2 = *q;
        q2.aggs = pa;
        q2.nagg = 2 * nk;
        if (out->nslices <= 1) q2.flags &= ~RFX_Q_SLICED; /* a whole result (FIRST values among its columns): whole key columns beside it */
        if (cols2) {
            q2.cols = cols2;
            q2.ncols = nc2;
        }
        rfx_groups_t *P = rc == RFX_OK ? (rfx_groups_t *)calloc(1, sizeof(*P)) : NULL;
        if (!P && rc == RFX_OK) rc = RFX_ENOMEM;
        if (P) {
            P->groups = out->groups;
            P->nslices = out->nslices;
            P->nkeys = out->nkeys;
            rc = group_by_pass(x, &q2, 0, 2 * nk, 0, cap, P);
        }
        const int nsl = P && P->nslices > 1 ? P->nslices : 1;
        for (int i = 0; P && i < nsl && rc == RFX_OK; i++) {
            const int s = P->nslices > 1 ? P->slice[i].shard : 0;
            const int64_t n = P->nslices > 1 ? P->slice[i].n : P->groups;
            if (n == 0) continue;
            rfx_hip_ctx_bind_thread(x->ctx[s]);
            for (int j = 0; j < nk && rc == RFX_OK; j++) {
                const void *mn = P->nslices > 1 ? P->slice[i].d_results[2 * j] : P->d_results[2 * j], *mx = P->nslices > 1 ? P->slice[i].d_results[2 * j + 1] : P->d_results[2 * j + 1];
                rfx_pred_t ne;
                rfx_value_t cv;
                int64_t differ = 0;
                memset(&ne, 0, sizeof(ne));
                ne.d_col = mn;
                ne.col_type = RFX_I64;
                ne.op = RFX_NE;
                ne.d_rhs_col = mx;
                ne.rhs_type = RFX_I64;
                rc = rfx_hip_filter_aggr_host(x->ctx[s], &ne, 1, RFX_AND, NULL, 0, n, &cv, &differ);
                if (rc == RFX_OK && differ) {
                    snprintf(x->err, sizeof(x->err), "row-hash collision between two key tuples");
                    rc = RFX_ESTATE;
                }
                if (rc == RFX_OK && tmp[j][0]) rc = rfx_hip_replace_i64(x->ctx[s], (const int64_t *)mx, n, repl[j], NULL_I64, (int64_t *)mx);
                if (rc == RFX_OK) {
                    if (P->nslices > 1) out->slice[i].d_keycols[k0 + j] = (int64_t *)mx;
                    if (i == 0) out->d_keycols[k0 + j] = (int64_t *)mx;
                }
            }
        }
        for (int j = 0; j < nk; j++)
            for (int s = 0; s < S; s++)
                if (tmp[j][s]) { /* (stream-ordered: the passes that read it are enqueued before the free) */
                    rfx_hip_ctx_bind_thread(x->ctx[s]);
                    rfx_hip_free(x->ctx[s], tmp[j][s]);
                }
        rfx_hip_ctx_bind_thread(x->ctx[0]);
        if (rc != RFX_OK && !x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
        for (int i = 0; P && i < P->nown; i++) own_on(out, P->own[i], P->own_shard[i]); /* the key columns live in the proof passes' blocks */
        free(P);
        free(cols2);
    }
    return rc;
}
int rfx_exec_group_by(rfx_exec_t *x, const rfx_query_t *q, rfx_groups_t *out) {
    if (!x || !q || !out || q->nkeys < 1 || q->nkeys > RFX_MAX_KEYS || !q->d_keys || q->nagg < 0 || q->nagg > RFX_EXEC_MAX_AGGS || q->npred < 0 || q->npred > RFX_MAX_PREDS)
        return RFX_EINVAL;
    rfx_hip_ctx_bind_thread(x->ctx[0]);
    x->stat[RFX_XSTAT_QUERIES]++;
    x->err[0] = 0;
    memset(out, 0, sizeof(*out));
    T_BEGIN(x);
    int rc = RFX_OK, first = 1;
    int64_t cap = 0;
    for (int a0 = 0; rc == RFX_OK && (a0 < q->nagg || first);) { /* more outputs than one table set carries: several passes, same groups, same order */
        const int na = q->nagg ? agg_chunk(q, a0) : 0;
        if (na < 0) { snprintf(x->err, sizeof(x->err), "rfx_exec: aggregate %d: nxnodes outside 0..%d or xnodes NULL", a0, RFX_MAX_XNODES); rc = RFX_EINVAL; break; }
        rc = group_by_pass(x, q, a0, na, first, cap, out);
        cap = out->capacity;
        first = 0;
        a0 += na;
        if (q->nagg == 0 || out->groups == 0) break;
    }
    out->nagg = q->nagg;
    if (rc == RFX_OK && out->path == RFX_PATH_ROWHASH && out->groups > 0 && (x->nshards > 1 || world_is_multi(x))) rc = rowhash_proof_passes(x, q, cap, out);
    if (rc == RFX_OK && out->nslices <= 1) { /* the whole result on shard 0: one slice, so that every reader walks slices */
        out->nslices = 1;
        out->slice[0].shard = 0;
        out->slice[0].g0 = 0;
        out->slice[0].n = out->groups;
        out->slice[0].d_keys = out->d_keys;
        out->slice[0].d_first = out->d_first;
        for (int k = 0; k < RFX_MAX_KEYS; k++) out->slice[0].d_keycols[k] = out->d_keycols[k];
        for (int a = 0; a < RFX_EXEC_MAX_AGGS; a++) out->slice[0].d_results[a] = out->d_results[a];
    }
    T_END(x, RFX_XSTAT_NS_TOTAL);
    if (rc != RFX_OK) rfx_exec_groups_free(x, out);
    return rc;
}
/* ---- the result to the host ---- */
typedef struct {
    rfx_exec_t *x;
    const rfx_groups_t *g;
    int n;
    const void *const *srcs;
    void *const *dsts;
} fetch_t;
/* the piece of column `src0` (a column pointer of slice 0) that slice i holds */
static const void *slice_col(const rfx_groups_t *g, int i, const void *src0) {
    const struct rfx_gslice *a = &g->slice[0], *b = &g->slice[i];
    if (!src0) return NULL;
    if (src0 == a->d_keys) return b->d_keys;
    if (src0 == a->d_first) return b->d_first;
    for (int k = 0; k < RFX_MAX_KEYS; k++)
        if (src0 == a->d_keycols[k]) return b->d_keycols[k];
    for (int r = 0; r < RFX_EXEC_MAX_AGGS; r++)
        if (src0 == a->d_results[r]) return b->d_results[r];
    return NULL;
}
static int ph_fetch(void *arg, int s) {
    fetch_t *F = (fetch_t *)arg;
    const rfx_groups_t *g = F->g;
    rfx_ctx_t *c = F->x->ctx[s];
    int any = 0, rc = RFX_OK;
    for (int i = 0; i < g->nslices && rc == RFX_OK; i++) {
        if (g->slice[i].shard != s || g->slice[i].n == 0) continue;
        for (int j = 0; j < F->n && rc == RFX_OK; j++) {
            const void *p = slice_col(g, i, F->srcs[j]);
            if (!p) { rfx_hip_ctx_sync(c); return RFX_EINVAL; }
            const size_t bytes = (size_t)g->slice[i].n * 8;
            if (bytes >= ((size_t)64 << 20) && !F->x->no_d2h_pipeline) rc = rfx_hip_d2h_pipelined(c, (char *)F->dsts[j] + (size_t)g->slice[i].g0 * 8, p, bytes); /* (a large column: pinned staging, parallel first touch) */
            else {
                rc = rfx_hip_d2h_async(c, (char *)F->dsts[j] + (size_t)g->slice[i].g0 * 8, p, bytes);
                any = 1;
            }
        }
    }
    if (any) { /* ONE wait for all of this shard's copies */
        const int src = rfx_hip_ctx_sync(c);
        if (rc == RFX_OK) rc = src;
    }
    return rc;
}
int rfx_exec_groups_fetch_all(rfx_exec_t *x, const rfx_groups_t *g, int n, const void *const *d_srcs, void *const *dsts) {
    if (!x || !g || n < 0 || (n && (!d_srcs || !dsts))) return RFX_EINVAL;
    if (n == 0 || g->groups == 0) return RFX_OK;
    T_BEGIN(x);
    int rc = RFX_OK;
    if (g->h_block) { /* small dense tables: the block is mirrored on the host already */
        for (int j = 0; j < n && rc == RFX_OK; j++) {
            const char *p = (const char *)d_srcs[j];
            if (p >= g->d_block && p + (size_t)g->groups * 8 <= g->d_block + g->block_bytes) memcpy(dsts[j], g->h_block + (p - g->d_block), (size_t)g->groups * 8);
            else rc = rfx_hip_d2h(x->ctx[0], dsts[j], p, (size_t)g->groups * 8);
        }
    } else {
        fetch_t F = {x, g, n, d_srcs, dsts};
        if (g->nslices <= 1) {
            rfx_hip_ctx_bind_thread(x->ctx[0]);
            if (g->nslices == 1 && g->slice[0].d_keys == NULL && g->slice[0].n == 0) rc = RFX_OK; /* (an empty result) */
            else rc = ph_fetch(&F, g->nslices == 1 ? g->slice[0].shard : 0);
        } else rc = run_shards(x, ph_fetch, &F);
        if (rc != RFX_OK && !x->err[0]) snprintf(x->err, sizeof(x->err), "rfx_exec: result read-back: %s", rc == RFX_EINVAL ? "a column that is not the result's" : rfx_hip_last_error());
    }
    if (x->timing) {
        const int64_t dt = now_ns() - t0_;
        x->stat[RFX_XSTAT_NS_FETCH] += dt;
        x->stat[RFX_XSTAT_NS_TOTAL] += dt;
    }
    return rc;
}
int rfx_exec_groups_fetch(rfx_exec_t *x, const rfx_groups_t *g, void *dst, const void *d_src, size_t bytes) {
    if (!x || !g || (!dst && bytes)) return RFX_EINVAL;
    if (g->h_block && (const char *)d_src >= g->d_block && (const char *)d_src + bytes <= g->d_block + g->block_bytes) {
        memcpy(dst, g->h_block + ((const char *)d_src - g->d_block), bytes);
        return RFX_OK;
    }
    if (g->nslices > 1) { /* a sliced column: whole or not at all */
        if (bytes != (size_t)g->groups * 8) return RFX_EINVAL;
        const void *srcs[1] = {d_src};
        void *dsts[1] = {dst};
        return rfx_exec_groups_fetch_all(x, g, 1, srcs, dsts);
    }
    T_BEGIN(x);
    const int rc = rfx_hip_d2h(x->ctx[0], dst, d_src, bytes);
    if (x->timing) {
        const int64_t dt = now_ns() - t0_;
        x->stat[RFX_XSTAT_NS_FETCH] += dt;
        x->stat[RFX_XSTAT_NS_TOTAL] += dt;
    }
    return rc;
}
void rfx_exec_groups_free(rfx_exec_t *x, rfx_groups_t *g) {
    if (!x || !g) return;
    for (int i = 0; i < g->nown; i++) {
        const int s = g->own_shard[i] >= 0 && g->own_shard[i] < x->nshards ? g->own_shard[i] : 0;
        rfx_hip_free(x->ctx[s], g->own[i]);
    }
    free((void *)g->h_block);
    memset(g, 0, sizeof(*g));
}

/* ------------------------------------------------------------------------------------------------ join index (one shard)
 * BUILD = the group-by's first-occurrence table over the right keys with zero aggregates (dense while the key range stays within
 * 4 x the right rows or 16 M slots, else hashed), PROBE = one pass over the left keys.  Several keys: ranges over BOTH sides that
 * multiply into 64 bits make one injective composite key per side (exact); wider tuples probe on the reference's row hash and every
 * matched row's key columns are compared afterwards (__index_list_cmp_row, done once). */
int rfx_exec_join_index(rfx_exec_t *x, const void *const *dlk, const void *const *drk, int nk, int64_t nl, int64_t nr, int64_t *d_ids, int *collision) {
    if (!x || !dlk || !drk || nk < 1 || nk > RFX_MAX_KEYS || !d_ids || nl < 0 || nr < 0) return RFX_EINVAL;
    if (x->nshards > 1) { snprintf(x->err, sizeof(x->err), "rfx_exec: joins run on one shard"); return RFX_ELIMIT; }
    rfx_ctx_t *c = x->ctx[0];
    rfx_hip_ctx_bind_thread(c);
    x->stat[RFX_XSTAT_QUERIES]++;
    x->err[0] = 0;
    if (collision) *collision = 0;
    void *tmp[8];
    int ntmp = 0, rc = RFX_OK, exact = 1;
#define JT(ptr, bytes) do { ptr = NULL; if ((rc = rfx_hip_malloc(c, &ptr, (bytes))) != RFX_OK) goto out; tmp[ntmp++] = ptr; } while (0)
    const void *lkey = dlk[0], *rkey = drk[0];
    if (nl == 0) return RFX_OK;
    if (nk > 1) {
        int64_t mins[RFX_MAX_KEYS], maxs[RFX_MAX_KEYS], mults[RFX_MAX_KEYS], tmax = 0, seen = 0;
        for (int i = 0; i < nk; i++) {
            int64_t a0, a1, b0, b1;
            if ((rc = rfx_hip_scope_i64(c, (const int64_t *)dlk[i], NULL, 0, RFX_AND, nl, &a0, &a1, &seen)) != RFX_OK ||
                (rc = rfx_hip_scope_i64(c, (const int64_t *)drk[i], NULL, 0, RFX_AND, nr, &b0, &b1, &seen)) != RFX_OK) goto out;
            mins[i] = a0 < b0 ? a0 : b0;
            maxs[i] = a1 > b1 ? a1 : b1;
        }
        void *lc, *rcc;
        JT(lc, (size_t)nl * 8);
        JT(rcc, (size_t)(nr ? nr : 1) * 8);
        if (rfx_composite_plan(mins, maxs, nk, mults, &tmax) == RFX_OK) {
            if ((rc = rfx_hip_composite_key(c, dlk, mins, mults, nk, nl, (int64_t *)lc)) != RFX_OK || (rc = rfx_hip_composite_key(c, drk, mins, mults, nk, nr, (int64_t *)rcc)) != RFX_OK) goto out;
        } else {
            if ((rc = rfx_hip_row_hash(c, dlk, nk, nl, 0, (int64_t *)lc)) != RFX_OK || (rc = rfx_hip_row_hash(c, drk, nk, nr, 0, (int64_t *)rcc)) != RFX_OK) goto out;
            exact = 0;
        }
        lkey = lc;
        rkey = rcc;
    }
    {
        int64_t kmin = 0, kmax = -1, seen = 0;
        if (nr > 0 && (rc = rfx_hip_scope_i64(c, (const int64_t *)rkey, NULL, 0, RFX_AND, nr, &kmin, &kmax, &seen)) != RFX_OK) goto out;
        const uint64_t range = nr > 0 ? (uint64_t)kmax - (uint64_t)kmin + 1 : 0;
        rfx_agg_t none;
        memset(&none, 0, sizeof(none));
        uint64_t lim = 4 * (uint64_t)nr > (1u << 24) ? 4 * (uint64_t)nr : (1u << 24);
        if ((uint64_t)seen > lim) lim = (uint64_t)seen;
        if (nr == 0) {
            /* no right row: every id is null -- a probe of an empty dense table */
            void *first;
            JT(first, 8);
            rfx_group_tables_t gt;
            memset(&gt, 0, sizeof(gt));
            gt.kmin = 0; gt.range = 1; gt.d_first = (int64_t *)first;
            if ((rc = rfx_hip_group_tables_init(c, &none, &gt)) != RFX_OK || (rc = rfx_hip_join_probe_dense(c, (const int64_t *)lkey, nl, INF_I64, 1, (const int64_t *)first, d_ids)) != RFX_OK) goto out;
        } else if (range != 0 && range <= lim && range <= (1ull << 29) && kmin != NULL_I64) {
            void *first;
            JT(first, (size_t)range * 8);
            rfx_group_tables_t gt;
            memset(&gt, 0, sizeof(gt));
            gt.kmin = kmin; gt.range = (int64_t)range; gt.d_first = (int64_t *)first;
            if ((rc = rfx_hip_group_tables_init(c, &none, &gt)) != RFX_OK || (rc = rfx_hip_group_dense_accumulate(c, (const int64_t *)rkey, NULL, 0, RFX_AND, &none, nr, 0, &gt)) != RFX_OK ||
                (rc = rfx_hip_join_probe_dense(c, (const int64_t *)lkey, nl, kmin, (int64_t)range, (const int64_t *)first, d_ids)) != RFX_OK) goto out;
        } else {
            int64_t cap_max = 16, cap;
            while (cap_max < 2 * nr) cap_max <<= 1;
            cap = cap_max < (1 << 22) ? cap_max : (1 << 22);
            for (;;) {
                void *store = NULL;
                if ((rc = rfx_hip_malloc(c, &store, (size_t)2 * (size_t)(cap + 1) * 8)) != RFX_OK) goto out;
                rfx_hash_tables_t ht;
                memset(&ht, 0, sizeof(ht));
                ht.capacity = cap; ht.d_keys = (int64_t *)store; ht.d_first = (int64_t *)store + (cap + 1);
                int arc = rfx_hip_hash_tables_init(c, &none, &ht);
                if (arc == RFX_OK) arc = rfx_hip_group_hash_accumulate(c, (const int64_t *)rkey, NULL, 0, RFX_AND, &none, nr, 0, &ht);
                if (arc == RFX_OK) arc = rfx_hip_join_probe_hash(c, (const int64_t *)lkey, nl, &ht, d_ids);
                if (arc == RFX_OK) arc = rfx_hip_ctx_sync(c); /* the probe has read the table before it is freed */
                rfx_hip_free(c, store);
                if (arc == RFX_OK) break;
                if (arc == RFX_ELIMIT && cap < cap_max) { cap = (cap << 4) < cap_max ? (cap << 4) : cap_max; x->stat[RFX_XSTAT_HASH_GROWN]++; continue; }
                rc = arc;
                goto out;
            }
        }
    }
    if (!exact) {
        void *chk;
        JT(chk, (size_t)nl * 8);
        for (int i = 0; i < nk; i++) {
            rfx_pred_t p;
            memset(&p, 0, sizeof(p));
            p.d_col = chk; p.col_type = RFX_I64; p.op = RFX_NE; p.d_rhs_col = dlk[i]; p.rhs_type = RFX_I64;
            rfx_value_t dummy[1];
            int64_t differ = 0;
            if ((rc = rfx_hip_gather_or(c, drk[i], dlk[i], d_ids, nl, 0, chk)) != RFX_OK || (rc = rfx_hip_filter_aggr_host(c, &p, 1, RFX_AND, NULL, 0, nl, dummy, &differ)) != RFX_OK) goto out;
            if (differ) {
                if (collision) *collision = 1;
                snprintf(x->err, sizeof(x->err), "row-hash collision between two key tuples");
                rc = RFX_ESTATE;
                goto out_quiet;
            }
        }
    }
    rc = rfx_hip_ctx_sync(c);
out:
    if (rc != RFX_OK) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
out_quiet:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(c, tmp[i]);
    return rc;
#undef JT
}
