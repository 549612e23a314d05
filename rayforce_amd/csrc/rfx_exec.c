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
    int own_overflow;    /* a result block found no room in rfx_groups_t.own[] (freed on the spot): the query fails */
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

/* The planner by concern (round 6: one 2 150-line file before).  ONE translation unit, read in this order: */
#include "rfx_exec_threads.c"
#include "rfx_exec_shard.c"
#include "rfx_exec_scalar.c"
#include "rfx_exec_group_phases.c"
#include "rfx_exec_merge.c"
#include "rfx_exec_groupby.c"
#include "rfx_exec_result.c"
