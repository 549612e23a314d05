/* rfx_exec_scalar.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  scalar aggregates (rfx_exec_filter_aggr) and where (rfx_exec_where) over the shards. */
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
