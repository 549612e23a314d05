/* rfx_exec_group_phases.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  group-by: the query's state (gq_t) and the per-shard phases (xbar, scopes, row hash, composite key, the pass, local merges). */
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

static int by_rows_env(void) { /* RFX_EMIT_BY_ROWS: 0 = the slot-ranking tail everywhere (A/B), 2 = the tail by rows wherever the probe arrays exist (tests) */
    static int v = -1;
    if (v < 0) v = getenv("RFX_EMIT_BY_ROWS") ? atoi(getenv("RFX_EMIT_BY_ROWS")) : 1;
    return v;
}
/* the hashed pass on one shard that will want every row's slot and group-first row afterwards (the row-hash route's proof; RFX_Q_PROBE_FIRST) */
static int want_row_slots(const gq_t *G, const shard_t *h) {
    return !G->dense && !G->multi && (G->rowhash || (G->q->flags & RFX_Q_PROBE_FIRST)) && h->nrows > 0 && !getenv("RFX_NO_INSERT_SLOTS");
}
/* ... and whose table is so large against the rows that the tail will walk the ROWS (ph_rank_emit): nothing but the insert pass, the first-row look-up and
 * the emit by rows ever reads that table -- it is laid out PACKED (one line per insert instead of one per field).  At least 2^24 slots: an estimate of 8M
 * groups and more, where the partitioned forms of the insert decline anyway */
static int want_packed(const gq_t *G, const shard_t *h) {
    if (!want_row_slots(G, h) || !by_rows_env() || G->nsl != 1 || getenv("RFX_NO_PACKED_TABLE") || G->narr < 2 || G->narr > 2 + 2 * RFX_MAX_AGGS) return 0;
    for (int a = 0; a < G->na; a++)
        if (h->aggs[a].nxnodes > 0) return 0; /* (expression trees: the packed insert does not carry them) */
    return by_rows_env() == 2 || (G->cap >= ((int64_t)1 << 24) && h->nrows <= 4 * (G->cap + 1));
}
/* tables of one shard: one block, arrays of `cells` 8-byte cells (what the exchanges and the merge kernel walk) -- or, packed, entries of narr cells */
static int tables_layout(gq_t *G, int s, int packed) {
    shard_t *h = &G->sh[s];
    rfx_ctx_t *c = G->x->ctx[s];
    const int64_t cells = G->dense ? (int64_t)G->range : G->cap + 1;
    const int64_t step = packed ? 1 : cells;
    int64_t *base = (int64_t *)h->store;
    int k = 0;
    h->packed = packed ? G->narr : 0;
    memset(&h->ht, 0, sizeof(h->ht));
    h->ht.capacity = G->cap;
    h->ht.nagg = G->na;
    h->ht.d_keys = base + (k++) * step;
    h->ht.d_first = base + (k++) * step;
    for (int a = 0; a < G->na; a++) {
        h->ht.d_acc[a] = base + (k++) * step;
        h->ht.d_cnt[a] = has_cnt(&h->aggs[a]) ? base + (k++) * step : NULL;
    }
    return packed ? rfx_hip_hash_tables_init_packed(c, h->aggs, &h->ht, G->narr) : rfx_hip_hash_tables_init(c, h->aggs, &h->ht);
}
static int tables_alloc(gq_t *G, int s) {
    shard_t *h = &G->sh[s];
    rfx_ctx_t *c = G->x->ctx[s];
    const int64_t cells = G->dense ? (int64_t)G->range : G->cap + 1;
    if (h->store) rfx_hip_free(c, h->store);
    h->store = NULL;
    h->packed = 0;
    int rc = rfx_hip_malloc(c, &h->store, (size_t)G->narr * (size_t)cells * 8);
    if (rc != RFX_OK) return rc;
    if (!G->dense) return tables_layout(G, s, want_packed(G, h));
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
        /* the row-hash route / a first-row probe on one shard will want every row's slot and group-first row: a table that takes the rows directly
         * (about as many groups as rows) says the slots while it inserts -- the probe pass over all rows again is saved (gb_prove_tuples) */
        const int want_slots = want_row_slots(G, h);
        h->slots_recorded = 0;
        if (want_slots && !h->probe_slots) {
            void *p = NULL;
            if ((rc = rfx_hip_malloc(c, &p, (size_t)h->nrows * 8)) != RFX_OK) return rc;
            h->probe_slots = (int64_t *)p;
        }
        if (h->packed) { /* (tables_alloc laid the table out packed: the rows go straight in, their scaled slots come back) */
            rc = rfx_hip_group_hash_accumulate_packed(c, (const int64_t *)h->key, h->preds, G->npred, G->q->logic, h->aggs, h->nrows, h->row0, &h->ht, h->packed, h->probe_slots);
            if (rc == RFX_OK) h->slots_recorded = 1;
            else if (rc == RFX_ESTATE && (rc = tables_layout(G, s, 0)) != RFX_OK) return rc; /* a query the packed insert does not carry: field by field after all */
        }
        if (!h->packed)
            rc = h->nrows == 0 ? RFX_OK
                               : rfx_hip_group_hash_accumulate_slots(c, (const int64_t *)h->key, h->preds, G->npred, G->q->logic, h->aggs, h->nrows, h->row0, &h->ht,
                                                                     want_slots ? h->probe_slots : NULL, want_slots ? &h->slots_recorded : NULL);
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
