/* rfx_exec_result.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  results to the host (sliced fetch over every owner's stream), release, the join index. */
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
int rfx_exec_groups_window(const rfx_groups_t *g, int64_t g0, int64_t n, rfx_groups_t *out) {
    if (!g || !out || g0 < 0 || n < 0 || g0 + n > g->groups || g->nslices > 1) return RFX_EINVAL;
    *out = *g;
    out->nown = 0; /* (a view: the blocks stay the original's) */
    out->groups = n;
    out->d_probe = NULL; /* (per row, not per group) */
#define WIN(p) do { if (p) p = (void *)((char *)(p) + (size_t)g0 * 8); } while (0)
    WIN(out->d_keys);
    WIN(out->d_first);
    for (int k = 0; k < RFX_MAX_KEYS; k++) WIN(out->d_keycols[k]);
    for (int a = 0; a < RFX_EXEC_MAX_AGGS; a++) WIN(out->d_results[a]);
    if (out->nslices == 1) {
        struct rfx_gslice *sl = &out->slice[0];
        WIN(sl->d_keys);
        WIN(sl->d_first);
        for (int k = 0; k < RFX_MAX_KEYS; k++) WIN(sl->d_keycols[k]);
        for (int a = 0; a < RFX_EXEC_MAX_AGGS; a++) WIN(sl->d_results[a]);
        sl->g0 = 0;
        sl->n = n;
    }
#undef WIN
    return RFX_OK;
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
static int join_index_on(rfx_exec_t *x, rfx_ctx_t *c, char *err, size_t errsz, const void *const *dlk, const void *const *drk, int nk, int64_t nl, int64_t nr, int64_t *d_ids, int *collision);
int rfx_exec_join_index(rfx_exec_t *x, const void *const *dlk, const void *const *drk, int nk, int64_t nl, int64_t nr, int64_t *d_ids, int *collision) {
    if (!x || !dlk || !drk || nk < 1 || nk > RFX_MAX_KEYS || !d_ids || nl < 0 || nr < 0) return RFX_EINVAL;
    if (x->nshards > 1) { snprintf(x->err, sizeof(x->err), "rfx_exec: rfx_exec_join_index runs on one shard (several: rfx_exec_join_index_shard per shard)"); return RFX_ELIMIT; }
    rfx_hip_ctx_bind_thread(x->ctx[0]);
    x->stat[RFX_XSTAT_QUERIES]++;
    x->err[0] = 0;
    return join_index_on(x, x->ctx[0], x->err, sizeof(x->err), dlk, drk, nk, nl, nr, d_ids, collision);
}
/* Over SHARDS (round 6): a broadcast join -- the BUILD side (the right table's key columns) WHOLE on the shard's device, the PROBE side this shard's
 * rows of the left table; every shard builds the same first-occurrence table and probes its own rows: no exchange, right row ids are global.  Called on
 * the shard's own thread (rfx_exec_run), its context bound; the error text goes to the shard's slot. */
int rfx_exec_join_index_shard(rfx_exec_t *x, int shard, const void *const *dlk, const void *const *drk, int nk, int64_t nl, int64_t nr, int64_t *d_ids, int *collision) {
    if (!x || shard < 0 || shard >= x->nshards || !dlk || !drk || nk < 1 || nk > RFX_MAX_KEYS || (!d_ids && nl > 0) || nl < 0 || nr < 0) return RFX_EINVAL;
    x->errs[shard][0] = 0;
    return join_index_on(x, x->ctx[shard], x->errs[shard], sizeof(x->errs[shard]), dlk, drk, nk, nl, nr, d_ids, collision);
}
static int join_index_on(rfx_exec_t *x, rfx_ctx_t *c, char *err, size_t errsz, const void *const *dlk, const void *const *drk, int nk, int64_t nl, int64_t nr, int64_t *d_ids, int *collision) {
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
                if (arc == RFX_ELIMIT && cap < cap_max) { cap = (cap << 4) < cap_max ? (cap << 4) : cap_max; __atomic_fetch_add(&x->stat[RFX_XSTAT_HASH_GROWN], 1, __ATOMIC_RELAXED); continue; }
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
                snprintf(err, errsz, "row-hash collision between two key tuples");
                rc = RFX_ESTATE;
                goto out_quiet;
            }
        }
    }
    rc = rfx_hip_ctx_sync(c);
out:
    if (rc != RFX_OK) snprintf(err, errsz, "%s", rfx_hip_last_error());
out_quiet:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(c, tmp[i]);
    return rc;
#undef JT
}
