/* rfx_exec_groupby.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  group-by: the planner proper -- scope memo, gb_setup .. gb_emit, one pass, the row-hash proof passes, rfx_exec_group_by. */
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
/* the result takes a device block over (released by rfx_exec_groups_free).  own[] is sized for the worst case (RFX_GROUPS_OWN); should a block
 * still find no room it is freed HERE -- never dropped silently -- and the query fails (x->own_overflow, checked where the result is handed out) */
static void own_on(rfx_exec_t *x, rfx_groups_t *g, void *p, int shard) {
    if (!p) return;
    if (g->nown < (int)(sizeof(g->own) / sizeof(g->own[0]))) {
        g->own_shard[g->nown] = (int8_t)shard;
        g->own[g->nown++] = p;
        return;
    }
    rfx_hip_free(x->ctx[shard >= 0 && shard < x->nshards ? shard : 0], p);
    x->own_overflow = 1;
}
static void own(rfx_exec_t *x, rfx_groups_t *g, void *p) { own_on(x, g, p, 0); }

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
        /* every row's group-first row AND table slot in one probe; the tuple proof as ONE pass over the key columns with one counter back (round 5: a
         * gather, a compare pass and a sync per key column); both arrays stay with the shard: a table that is large against the rows is emitted by
         * ROWS (ph_rank_emit) instead of by ranking its slots */
        void *ids = NULL, *slots = NULL;
        const int recorded = h->slots_recorded && h->probe_slots; /* (the insert pass said every row's slot: only the groups' first rows are looked up) */
        if (h->probe_slots && !recorded) { rfx_hip_free(c, h->probe_slots); h->probe_slots = NULL; }
        rc = rfx_hip_malloc(c, &ids, (size_t)(h->nrows ? h->nrows : 1) * 8);
        if (recorded) {
            slots = h->probe_slots;
            h->probe_slots = NULL;
            if (rc == RFX_OK) rc = rfx_hip_hash_slot_first(c, &h->ht, (const int64_t *)slots, h->nrows, (int64_t *)ids);
        } else {
            if (rc == RFX_OK) rc = rfx_hip_malloc(c, &slots, (size_t)(h->nrows ? h->nrows : 1) * 8);
            if (rc == RFX_OK) rc = rfx_hip_join_probe_hash_slots(c, (const int64_t *)h->key, h->nrows, &h->ht, (int64_t *)ids, (int64_t *)slots);
        }
        int collision = 0;
        if (rc == RFX_OK && G->rowhash) {
            int64_t differ = 0;
            rc = rfx_hip_tuple_check(c, h->keys, G->nkeys, (const int64_t *)ids, h->nrows, &differ);
            if (rc == RFX_OK && differ) collision = 1;
        }
        if (rc == RFX_OK && !collision) {
            h->probe_ids = (int64_t *)ids;
            h->probe_slots = (int64_t *)slots;
            ids = slots = NULL;
            if ((q->flags & RFX_Q_PROBE_FIRST) && G->first_pass) {
                out->d_probe = h->probe_ids;
                own(x, out, h->probe_ids);
                h->probe_owned_by_result = 1;
            }
        }
        if (ids) rfx_hip_free(c, ids);
        if (slots) rfx_hip_free(c, slots);
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
        own(x, out, blk);
    }
    if (!G->first_pass) { /* a later pass of a long output list: its own block, no mirror (fetched through the device) */
        free(mirror);
        own(x, out, blk);
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
            own_on(x, out, h->dout, s);
            if (G->first_pass) {
                sl->d_keys = (int64_t *)h->dout;
                sl->d_first = (int64_t *)h->dfirst;
                own_on(x, out, h->dfirst, s);
                for (int k = 0; k < G->nkeys && G->nkeys > 1; k++) {
                    sl->d_keycols[k] = (int64_t *)h->kc[k];
                    own_on(x, out, h->kc[k], s);
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
            own(x, out, h->dout);
            if (G->first_pass) {
                out->d_keys = (int64_t *)h->dout;
                /* several keys: the result's key columns -- decoded from the composite key (key_i = min_i + (composite / mult_i) % range_i
                 * = key_i[first row], core/query.c:110-135) or, on the row-hash path, gathered at the groups' first rows */
                /* (row hash over several shards: the proof passes of rfx_exec_group_by bring the key columns -- no shard holds every first row) */
                for (int k = 0; k < G->nkeys && G->nkeys > 1 && !(G->rowhash && G->multi) && rc == RFX_OK; k++) {
                    void *cell = NULL;
                    rc = rfx_hip_malloc(c, &cell, (size_t)g * 8);
                    if (rc != RFX_OK) break;
                    own(x, out, cell);
                    out->d_keycols[k] = (int64_t *)cell;
                    if (!G->rowhash) rc = rfx_hip_composite_decode(c, (const int64_t *)h->dout, g, G->kmins[k], G->kmults[k], G->kmaxs[k] - G->kmins[k] + 1, (int64_t *)cell);
                }
                if (rc == RFX_OK && G->nkeys > 1 && G->rowhash && !G->multi) /* (every key column at the groups' first rows: ONE launch, the first rows read once) */
                    rc = rfx_hip_gather_many(c, (const void *const *)h->keys, G->nkeys, (const int64_t *)h->dfirst, g, (void *const *)out->d_keycols);
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
                own(x, out, h->dfirst);
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
        for (int i = 0; P && i < P->nown; i++) own_on(x, out, P->own[i], P->own_shard[i]); /* the key columns live in the proof passes' blocks */
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
    if (rc == RFX_OK && x->own_overflow) {
        snprintf(x->err, sizeof(x->err), "rfx_exec: the result registers more device blocks than rfx_groups_t.own[] holds (%d)", (int)RFX_GROUPS_OWN);
        rc = RFX_ELIMIT;
    }
    x->own_overflow = 0;
    if (rc != RFX_OK) rfx_exec_groups_free(x, out);
    return rc;
}
