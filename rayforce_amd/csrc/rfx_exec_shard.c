/* rfx_exec_shard.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  one shard's view of a query (its row range of every column, its scratch), mask queries. */
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
    /* hashed path on one shard: per row the first row of its group and the table slot of its key (gb_prove_tuples' probe), kept for an emit by rows */
    int64_t *probe_ids, *probe_slots;
    int slots_recorded; /* probe_slots was written by the insert pass itself (rfx_hip_group_hash_accumulate_slots): no probe needed, only the first rows */
    int packed;         /* > 0: the hashed table is PACKED (rfx_hip.h), this many cells per entry; probe_slots holds scaled slots */
    int probe_owned_by_result; /* (RFX_Q_PROBE_FIRST: the result took probe_ids over) */
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
    if (h->probe_ids && !h->probe_owned_by_result) rfx_hip_free(x->ctx[s], h->probe_ids);
    if (h->probe_slots) rfx_hip_free(x->ctx[s], h->probe_slots);
    h->probe_ids = h->probe_slots = NULL;
    h->slots_recorded = 0;
    h->packed = 0;
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
