/* rfx_ops_update.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * rfx_update. */
/* ------------------------------------------------------------------------------------------------ update
 * (update {col: mapping ... from: t [where: p] [by: k]}) -- ray_update, core/update.c:936-1106.  The reference turns `where:` into
 * row ids (ray_where), evaluates every mapping over the filtered / grouped table and writes: under a filter, value i goes to row
 * ids[i]; under `by:`, each group's aggregate goes to all of that group's selected rows; a name the table does not have becomes a
 * new column that is null elsewhere (__update_table).  Covered here: `from:` a table VALUE (the quoted-symbol form updates the
 * host's global in place: the host's own job), flat or nested `where:`, mappings that are an i64 / f64 atom, a column, an
 * element-wise expression (+ - * div, nested) -- and, with `by:` one 8-byte integer key column, (aggr column) for sum / avg / min /
 * max / count / first under a flat `where:`.  Value and column types must agree (the reference also casts f64 into i64 columns:
 * delegated).  Anything else is the host's ray_update. */
static obj_p delegate_update(obj_p dict, const char *why) {
    if (H.bound == 1 && H.f[F_UPDATE]) return HOST_CALL(((rfx_unary_f)H.f[F_UPDATE])(dict));
    char b[300];
    snprintf(b, sizeof(b), "rfx_update: shape not covered by the MI355X path (%s) and no host ray_update to delegate to", why);
    return fail(b);
}
/* ---- update, step by step: upd_where -> upd_by -> per mapping { upd_value -> upd_column } -> upd_result.  Every step answers UPD_GO, UPD_BACK
 * (the shape is the host's: u->why says which) or UPD_STOP (u->res is the error object); update_impl owns the cleanup. ---- */
enum { UPD_GO = 0, UPD_BACK = 1, UPD_STOP = 2 };
typedef struct {
    obj_p tab;
    int64_t nrows;
    void *tmp[3 * RFX_MAX_AGGS + 8]; /* device blocks of this call */
    int ntmp;
    const char *why;
    obj_p res;
    wplan_t wp;      /* where: as comparisons (when flat) */
    int64_t *d_ids;  /* ... as row ids (NULL: every row) */
    int64_t m;       /* rows written */
    const void *dk;  /* by: the key column on the device, its scope through the predicates */
    int64_t kmin, kmax, seen;
} upd_t;
typedef struct { /* what one mapping writes */
    int vtype;             /* RFX_I64 | RFX_F64: element type of the values */
    const void *dvals_col; /* a full-length value column (no by:), or */
    uint64_t atom_bits;    /* ... one value for every row, or */
    rfx_agg_t agg;         /* ... (by:) an aggregate per group */
} upd_val_t;
static int upd_back(upd_t *u, const char *why) { u->why = why; return UPD_BACK; }
static int upd_stop(upd_t *u, obj_p err) { u->res = err; return UPD_STOP; }

/* where: -> row ids (ray_where) */
static int upd_where(upd_t *u, obj_p where, obj_p by) {
    int flat = 1;
    const int prc = plan_where(u->tab, where, &u->wp);
    if (prc == -2) return upd_stop(u, fail_hip("column upload"));
    if (prc < 0) flat = 0;
    if (by && !flat) return upd_back(u, "by: with a nested where: tree");
    const int wrc = where_ids(u->tab, where, &u->wp, flat, u->nrows, &u->d_ids, &u->m);
    if (wrc == -2) return upd_stop(u, fail_hip("where"));
    if (wrc < 0) return upd_back(u, "where: shape");
    if (u->d_ids) u->tmp[u->ntmp++] = u->d_ids;
    return UPD_GO;
}
/* by: one 8-byte integer key column with a dense scope */
static int upd_by(upd_t *u, obj_p by) {
    if (by->type != -RFX_TYPE_SYMBOL) return upd_back(u, "by: is not one column");
    obj_p kc = table_col(u->tab, by->i64);
    if (!kc || !(kc->type == RFX_TYPE_I64 || kc->type == RFX_TYPE_SYMBOL || kc->type == RFX_TYPE_TIMESTAMP)) return upd_back(u, "by: key is not an 8-byte integer column");
    if (resident(kc, 0, &u->dk) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
    if (rfx_hip_scope_i64(g_ctx, (const int64_t *)u->dk, u->wp.preds, u->wp.npred, u->wp.logic, u->nrows, &u->kmin, &u->kmax, &u->seen) != RFX_OK) return upd_stop(u, fail_hip("scope"));
    const uint64_t range = u->seen > 0 ? (uint64_t)u->kmax - (uint64_t)u->kmin + 1 : 0;
    if (u->seen > 0 && !(range != 0 && range <= (uint64_t)u->seen && u->kmin != RFX_NULL_I64 && range <= (1ull << 31))) return upd_back(u, "by: sparse or null keys");
    return UPD_GO;
}
/* the values of one mapping: (aggr column) under by:, else an atom, a column or an element-wise expression (evaluated into a device column) */
static int upd_value(upd_t *u, obj_p e, obj_p by, upd_val_t *v) {
    obj_p tab = u->tab;
    memset(v, 0, sizeof(*v));
    if (by) {
        if (e->type != RFX_TYPE_LIST || e->len != 2) return upd_back(u, "by: mapping is not (aggr column)");
        const int f = fn_id(RFX_AS_LIST(e)[0]);
        static const int KIND[] = {RFX_AGG_SUM, RFX_AGG_AVG, RFX_AGG_MIN, RFX_AGG_MAX, RFX_AGG_COUNT, RFX_AGG_FIRST};
        obj_p a = RFX_AS_LIST(e)[1];
        if (f < F_SUM || f > F_FIRST || a->type != -RFX_TYPE_SYMBOL) return upd_back(u, "by: mapping is not (aggr column)");
        obj_p c = table_col(tab, a->i64);
        if (!c || !col_ctype(c) || c->type == RFX_TYPE_SYMBOL) return upd_back(u, "aggregate column type");
        const void *d;
        if (resident(c, 0, &d) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
        v->agg.d_col = d;
        v->agg.col_type = col_ctype(c);
        v->agg.kind = KIND[f - F_SUM];
        v->vtype = (f == F_AVG) ? RFX_F64 : (f == F_COUNT) ? RFX_I64 : col_ctype(c);
    } else if (e->type == -RFX_TYPE_I64) { v->vtype = RFX_I64; v->atom_bits = (uint64_t)e->i64; }
    else if (e->type == -RFX_TYPE_F64) { v->vtype = RFX_F64; memcpy(&v->atom_bits, &e->f64, 8); }
    else if (e->type == -RFX_TYPE_SYMBOL) {
        obj_p c = table_col(tab, e->i64);
        if (!c || !(c->type == RFX_TYPE_I64 || c->type == RFX_TYPE_F64)) return upd_back(u, "mapping column type");
        if (resident(c, 0, &v->dvals_col) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
        v->vtype = col_ctype(c);
    } else if (e->type == RFX_TYPE_LIST && e->len == 3) {
        rfx_xnode_t nodes[RFX_MAX_XNODES];
        int nn = 0, ncols = 0;
        const char *why = NULL;
        const int top = build_xnodes(tab, e, nodes, &nn, &ncols, &why);
        if (top == -2) return upd_stop(u, fail_hip("column upload"));
        if (top < 0) return upd_back(u, why);
        if (ncols == 0) return upd_back(u, "expression without a column");
        rfx_agg_t xa;
        memset(&xa, 0, sizeof(xa));
        xa.kind = RFX_AGG_SUM;
        xa.col_type = RFX_I64;
        xa.nxnodes = nn;
        xa.xnodes = nodes;
        void *dx = NULL;
        if (rfx_hip_malloc(g_ctx, &dx, (size_t)u->nrows * 8) != RFX_OK) return upd_stop(u, fail_hip("expression column"));
        u->tmp[u->ntmp++] = dx;
        int32_t ot = RFX_I64;
        if (rfx_hip_eval_expr(g_ctx, &xa, u->nrows, dx, &ot) != RFX_OK) return upd_stop(u, fail_hip("eval_expr"));
        v->dvals_col = dx;
        v->vtype = ot;
    } else return upd_back(u, "mapping is neither an atom, a column nor an element-wise expression");
    return UPD_GO;
}
/* a device copy of the column `tc` (or nulls for a new one), the writes of one mapping into it, and the host vector it becomes */
static int upd_column(upd_t *u, obj_p tc, obj_p by, const upd_val_t *v, obj_p *newcol) {
    const int64_t nrows = u->nrows;
    /* __suitable_types (update.c:81-107): same type, or an i64 column taking f64 values by conversion -- the latter is the host's */
    int8_t out_type = v->vtype == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64;
    if (tc) {
        if (!(tc->type == RFX_TYPE_I64 || tc->type == RFX_TYPE_F64)) return upd_back(u, "updated column is not i64 / f64");
        if (col_ctype(tc) != v->vtype) return upd_back(u, "value type differs from the column's (the reference converts; delegated)");
        out_type = tc->type;
    }
    void *dcol = NULL;
    if (rfx_hip_malloc(g_ctx, &dcol, (size_t)nrows * 8) != RFX_OK) return upd_stop(u, fail_hip("column copy"));
    u->tmp[u->ntmp++] = dcol;
    if (tc) {
        const void *dold;
        if (resident(tc, 0, &dold) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
        if (rfx_hip_update_set(g_ctx, dcol, NULL, nrows, dold, 0) != RFX_OK) return upd_stop(u, fail_hip("column copy")); /* copy: ids NULL, vals = old */
    } else if (rfx_hip_update_set(g_ctx, dcol, NULL, nrows, NULL, v->vtype == RFX_F64 ? 0x7FF8000000000000ull : 0x8000000000000000ull) != RFX_OK)
        return upd_stop(u, fail_hip("column fill"));
    if (u->m > 0) {
        if (!by) {
            if (rfx_hip_update_set(g_ctx, dcol, u->d_ids, u->m, v->dvals_col, v->atom_bits) != RFX_OK) return upd_stop(u, fail_hip("update_set"));
        } else if (u->seen > 0) {
            const int64_t range = (int64_t)((uint64_t)u->kmax - (uint64_t)u->kmin + 1);
            int narr = 0;
            rfx_hip_group_table_arrays(&v->agg, 1, &narr);
            void *store = NULL;
            if (rfx_hip_malloc(g_ctx, &store, (size_t)narr * (size_t)range * 8) != RFX_OK) return upd_stop(u, fail_hip("group tables"));
            u->tmp[u->ntmp++] = store;
            int64_t *base = (int64_t *)store;
            rfx_group_tables_t gt;
            memset(&gt, 0, sizeof(gt));
            gt.kmin = u->kmin;
            gt.range = range;
            gt.nagg = 1;
            gt.d_first = base;
            gt.d_acc[0] = base + range;
            gt.d_cnt[0] = narr > 2 ? base + 2 * range : NULL;
            if (rfx_hip_group_tables_init(g_ctx, &v->agg, &gt) != RFX_OK ||
                rfx_hip_group_dense_accumulate(g_ctx, (const int64_t *)u->dk, u->wp.preds, u->wp.npred, u->wp.logic, &v->agg, nrows, 0, &gt) != RFX_OK ||
                rfx_hip_update_group(g_ctx, dcol, (const int64_t *)u->dk, u->d_ids, u->m, &v->agg, &gt) != RFX_OK)
                return upd_stop(u, fail_hip("grouped update"));
        }
    }
    *newcol = H.vector(out_type, nrows);
    if (rfx_hip_d2h(g_ctx, RFX_AS_RAW(*newcol), dcol, (size_t)nrows * 8) != RFX_OK) return upd_stop(u, fail_hip("read-back"));
    return UPD_GO;
}
/* the result table: the old columns (shared), replaced or extended by the updated ones (which it takes over) */
static obj_p upd_result(obj_p tab, const int64_t *mnames, obj_p *newcols, int nmap) {
    obj_p tnames = RFX_AS_LIST(tab)[0], tcols = RFX_AS_LIST(tab)[1];
    int64_t nnew = 0;
    for (int i = 0; i < nmap; i++)
        if (!table_col(tab, mnames[i])) {
            int dup = 0;
            for (int j = 0; j < i; j++) dup |= mnames[j] == mnames[i];
            if (!dup) nnew++;
        }
    obj_p rk = H.vector(RFX_TYPE_SYMBOL, tnames->len + nnew), rv = H.vector(RFX_TYPE_LIST, tnames->len + nnew);
    for (int64_t c = 0; c < tnames->len; c++) {
        RFX_AS_I64(rk)[c] = RFX_AS_I64(tnames)[c];
        obj_p col = NULL;
        for (int i = nmap - 1; i >= 0 && !col; i--)
            if (mnames[i] == RFX_AS_I64(tnames)[c] && newcols[i]) { col = newcols[i]; newcols[i] = NULL; }
        RFX_AS_LIST(rv)[c] = col ? col : H.clone(RFX_AS_LIST(tcols)[c]);
    }
    int64_t at = tnames->len;
    for (int i = 0; i < nmap; i++)
        if (newcols[i] && !table_col(tab, mnames[i])) {
            RFX_AS_I64(rk)[at] = mnames[i];
            RFX_AS_LIST(rv)[at++] = newcols[i];
            newcols[i] = NULL;
        }
    return H.table(rk, rv);
}

/* ---- over SHARDS (round 6).  An update is row-local once every row knows its value: every shard writes ITS rows of the new column in one pass --
 *   new[i] = selected(i) ? value(i) : old(i)            (rfx_hip_update_select)
 * with `selected` = the where: tree as a 0 / 1 column per shard (mask_column_sharded: the comparisons' operands are resident shard by shard), and `value`
 * an atom, the shard's piece of a column, an element-wise expression evaluated over the shard's rows (expr_operand) or, under by:, the row's group
 * aggregate: the planner's sharded group-by over the same selection (every shard scatters its rows, the tables merge: rfx_exec_group_by) gives (key,
 * aggregate) per group, which becomes a value table over the key range that every shard looks its rows' keys up in (rfx_hip_group_ids_table).  The
 * reference runs the same three steps over its pool: ray_where -> the mappings over the MAPFILTER / MAPGROUP columns -> set_ids (core/update.c:1001,
 * 1048-1080, 781-850). ---- */
typedef int (*piece_fn)(void *arg, int s, int64_t r0, int64_t n, void *d_out);
static int map_shards(obj_p out, size_t esz, piece_fn fn, void *arg);
static const void *shard_piece(const void *p, int s);
static int mask_column_sharded(obj_p tab, obj_p where, int64_t nrows, const void **d_col);
typedef struct {
    const void *d_old, *d_mask, *d_vals; /* shard 0's addresses (pieces through shard_piece); NULL: none */
    uint64_t null_bits, atom_bits;
    /* by: */
    const void *d_key;
    int64_t kmin, range;
    void *table[RFX_MAX_SHARDS], *looked[RFX_MAX_SHARDS];
} usel_t;
static int usel_piece(void *arg, int s, int64_t r0, int64_t n, void *d_out) {
    (void)r0;
    usel_t *U = (usel_t *)arg;
    const void *vals = shard_piece(U->d_vals, s);
    if (U->d_key) { /* by: every row's group aggregate through the value table */
        int rc = rfx_hip_malloc(g_ctxs[s], &U->looked[s], (size_t)n * 8);
        if (rc == RFX_OK) rc = rfx_hip_group_ids_table(g_ctxs[s], (const int64_t *)shard_piece(U->d_key, s), n, U->kmin, U->range, (const int64_t *)U->table[s], (int64_t *)U->looked[s]);
        if (rc != RFX_OK) return rc;
        vals = U->looked[s];
    }
    return rfx_hip_update_select(g_ctxs[s], d_out, shard_piece(U->d_old, s), U->null_bits, (const int64_t *)shard_piece(U->d_mask, s), vals, U->atom_bits, n);
}
/* UPD_GO: newcols[] filled; UPD_BACK / UPD_STOP as the one-device steps */
static int upd_sharded(upd_t *u, obj_p where, obj_p by, int nmap, const int64_t *mnames, obj_p *mexpr, obj_p *newcols) {
    obj_p tab = u->tab;
    const int64_t nrows = u->nrows;
    const void *dmask = NULL, *dk = NULL;
    if (where) {
        const int mrc = mask_column_sharded(tab, where, nrows, &dmask);
        if (mrc == -2) return upd_stop(u, fail_hip("where"));
        if (mrc != 0) return upd_back(u, "where: shape");
    }
    if (by) {
        if (by->type != -RFX_TYPE_SYMBOL) return upd_back(u, "by: is not one column");
        obj_p kc = table_col(tab, by->i64);
        if (!kc || !(kc->type == RFX_TYPE_I64 || kc->type == RFX_TYPE_SYMBOL || kc->type == RFX_TYPE_TIMESTAMP)) return upd_back(u, "by: key is not an 8-byte integer column");
        if (resident(kc, 0, &dk) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
    }
    for (int i = 0; i < nmap; i++) {
        obj_p e = mexpr[i], tc = table_col(tab, mnames[i]);
        usel_t U;
        memset(&U, 0, sizeof(U));
        int vtype = 0, st = UPD_GO;
        rfx_groups_t R;
        int have_R = 0;
        obj_p gk = NULL, gv = NULL;
        if (by) {
            if (e->type != RFX_TYPE_LIST || e->len != 2) return upd_back(u, "by: mapping is not (aggr column)");
            const int f = fn_id(RFX_AS_LIST(e)[0]);
            static const int KIND[] = {RFX_AGG_SUM, RFX_AGG_AVG, RFX_AGG_MIN, RFX_AGG_MAX, RFX_AGG_COUNT, RFX_AGG_FIRST};
            obj_p a = RFX_AS_LIST(e)[1];
            if (f < F_SUM || f > F_FIRST || a->type != -RFX_TYPE_SYMBOL) return upd_back(u, "by: mapping is not (aggr column)");
            obj_p c = table_col(tab, a->i64);
            if (!c || !col_ctype(c) || c->type == RFX_TYPE_SYMBOL) return upd_back(u, "aggregate column type");
            const void *dc;
            if (resident(c, 0, &dc) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
            vtype = (f == F_AVG) ? RFX_F64 : (f == F_COUNT) ? RFX_I64 : col_ctype(c);
            /* the groups over the SAME selection, through the planner: scatter on every shard, merged tables, groups in first-occurrence order */
            rfx_agg_t ag;
            memset(&ag, 0, sizeof(ag));
            ag.d_col = dc;
            ag.col_type = col_ctype(c);
            ag.kind = KIND[f - F_SUM];
            rfx_pred_t pm;
            memset(&pm, 0, sizeof(pm));
            pm.d_col = dmask; pm.col_type = RFX_I64; pm.op = RFX_NE; pm.rhs_type = RFX_I64; pm.rhs_i = 0;
            const void *dkeys[1] = {dk};
            rfx_query_t Q;
            memset(&Q, 0, sizeof(Q));
            Q.preds = dmask ? &pm : NULL;
            Q.npred = dmask ? 1 : 0;
            Q.logic = RFX_AND;
            Q.aggs = &ag;
            Q.nagg = 1;
            Q.nkeys = 1;
            Q.d_keys = dkeys;
            Q.nrows = nrows;
            Q.cols = g_qcols;
            Q.ncols = g_nqcols;
            Q.flags = RFX_Q_REFUSE_NULL_KEY | RFX_Q_SLICED;
            const int grc = rfx_exec_group_by(g_x, &Q, &R);
            if (grc == RFX_EXEC_NULL_KEY) return upd_back(u, "by: sparse or null keys");
            if (grc != RFX_OK) return upd_stop(u, fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error()));
            have_R = 1;
            if (!(R.path == RFX_PATH_DENSE || R.path == RFX_PATH_DENSE_SMALL)) { rfx_exec_groups_free(g_x, &R); return upd_back(u, "by: sparse or null keys"); }
            U.d_key = dk;
            if (R.groups > 0) {
                gk = H.vector(RFX_TYPE_I64, R.groups);
                gv = H.vector(RFX_TYPE_I64, R.groups);
                const void *srcs[2] = {R.d_keys, R.d_results[0]};
                void *dsts[2] = {RFX_AS_RAW(gk), RFX_AS_RAW(gv)};
                if (rfx_exec_groups_fetch_all(g_x, &R, 2, srcs, dsts) != RFX_OK) st = upd_stop(u, fail_hip("group results"));
                int64_t lo = RFX_INF_I64, hi = RFX_NULL_I64;
                for (int64_t g = 0; g < R.groups && st == UPD_GO; g++) {
                    const int64_t kk = RFX_AS_I64(gk)[g];
                    lo = kk < lo ? kk : lo;
                    hi = kk > hi ? kk : hi;
                }
                const uint64_t range = st == UPD_GO ? (uint64_t)hi - (uint64_t)lo + 1 : 0;
                if (st == UPD_GO && !(range != 0 && range <= (1ull << 28))) st = upd_back(u, "by: sparse or null keys");
                if (st == UPD_GO) {
                    obj_p tbl = H.vector(RFX_TYPE_I64, (int64_t)range);
                    for (uint64_t k = 0; k < range; k++) RFX_AS_I64(tbl)[k] = 0;
                    for (int64_t g = 0; g < R.groups; g++) RFX_AS_I64(tbl)[RFX_AS_I64(gk)[g] - lo] = RFX_AS_I64(gv)[g];
                    U.kmin = lo;
                    U.range = (int64_t)range;
                    for (int sh = 0; sh < g_nshards && st == UPD_GO; sh++) { /* (a table of the key range on every shard: small against the rows) */
                        rfx_hip_ctx_bind_thread(g_ctxs[sh]);
                        if (rfx_hip_malloc(g_ctxs[sh], &U.table[sh], (size_t)range * 8) != RFX_OK || rfx_hip_h2d_pipelined(g_ctxs[sh], U.table[sh], RFX_AS_RAW(tbl), (size_t)range * 8) != RFX_OK)
                            st = upd_stop(u, fail_hip("value table"));
                    }
                    rfx_hip_ctx_bind_thread(g_ctx);
                    H.drop(tbl);
                }
            } else { /* nothing selected: no row is written */
                U.d_key = NULL;
                U.d_vals = NULL;
            }
        } else if (e->type == -RFX_TYPE_I64) { vtype = RFX_I64; U.atom_bits = (uint64_t)e->i64; }
        else if (e->type == -RFX_TYPE_F64) { vtype = RFX_F64; memcpy(&U.atom_bits, &e->f64, 8); }
        else if (e->type == -RFX_TYPE_SYMBOL) {
            obj_p c = table_col(tab, e->i64);
            if (!c || !(c->type == RFX_TYPE_I64 || c->type == RFX_TYPE_F64)) return upd_back(u, "mapping column type");
            if (resident(c, 0, &U.d_vals) != RFX_OK) return upd_stop(u, fail_hip("column upload"));
            vtype = col_ctype(c);
        } else if (e->type == RFX_TYPE_LIST && e->len == 3) {
            int ct = RFX_I64;
            const int xrc = expr_operand(tab, e, &U.d_vals, &ct); /* (every shard evaluates its rows into a scratch column of the call) */
            if (xrc == -2) return upd_stop(u, fail_hip("eval_expr"));
            if (xrc != 0) return upd_back(u, "mapping is neither an atom, a column nor an element-wise expression");
            vtype = ct;
        } else return upd_back(u, "mapping is neither an atom, a column nor an element-wise expression");
        int8_t out_type = vtype == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64;
        if (st == UPD_GO && tc) {
            if (!(tc->type == RFX_TYPE_I64 || tc->type == RFX_TYPE_F64)) st = upd_back(u, "updated column is not i64 / f64");
            else if (col_ctype(tc) != vtype) st = upd_back(u, "value type differs from the column's (the reference converts; delegated)");
            else {
                out_type = tc->type;
                if (resident(tc, 0, &U.d_old) != RFX_OK) st = upd_stop(u, fail_hip("column upload"));
            }
        }
        if (st == UPD_GO) {
            U.null_bits = vtype == RFX_F64 ? 0x7FF8000000000000ull : 0x8000000000000000ull;
            U.d_mask = (by && !U.d_key) ? NULL : dmask;
            if (by && !U.d_key) { /* (by: with an empty selection: the old column, or nulls for a new one) */
                U.d_vals = U.d_old;
                U.atom_bits = U.null_bits;
            }
            newcols[i] = H.vector(out_type, nrows);
            if (map_shards(newcols[i], 8, usel_piece, &U) != RFX_OK) st = upd_stop(u, fail_hip("update over the shards"));
        }
        for (int sh = 0; sh < g_nshards; sh++) {
            if (!U.table[sh] && !U.looked[sh]) continue;
            rfx_hip_ctx_bind_thread(g_ctxs[sh]);
            if (U.table[sh]) rfx_hip_free(g_ctxs[sh], U.table[sh]);
            if (U.looked[sh]) rfx_hip_free(g_ctxs[sh], U.looked[sh]);
        }
        rfx_hip_ctx_bind_thread(g_ctx);
        if (gk) H.drop(gk);
        if (gv) H.drop(gv);
        if (have_R) rfx_exec_groups_free(g_x, &R);
        if (st != UPD_GO) return st;
    }
    return UPD_GO;
}

static obj_p update_impl(obj_p dict) {
    rfx_host_bind();
    if (!dict || dict->type != RFX_TYPE_DICT || RFX_AS_LIST(dict)[0]->type != RFX_TYPE_SYMBOL) return fail("update: expected a dict");
    obj_p from = dict_get(dict, "from");
    if (!from) return fail("'update' expects 'from' param");
    /* `from: 't` parses as (quote t): the in-place form on a global -- evaluated by the host only */
    if (from->type == RFX_TYPE_LIST) return delegate_update(dict, "from: is an expression (in-place update of a global)");
    obj_p tab = HOST_CALL(H.eval(from)); /* (not under our lock: the host may fan the evaluation out) */
    if (!tab || tab->type == RFX_TYPE_ERR) return tab;
    if (tab->type != RFX_TYPE_TABLE) {
        H.drop(tab);
        return delegate_update(dict, "from: does not evaluate to a table value");
    }
    upd_t u;
    memset(&u, 0, sizeof(u));
    u.tab = tab;
    u.wp.logic = RFX_AND;
    u.kmax = -1;
    obj_p where = dict_get(dict, "where"), by = dict_get(dict, "by");
    obj_p dkeys = RFX_AS_LIST(dict)[0], dvals = RFX_AS_LIST(dict)[1];
    const int64_t s_from = H.intern("from", 4), s_where = H.intern("where", 5), s_by = H.intern("by", 2);
    obj_p tcols = RFX_AS_LIST(tab)[1];
    u.nrows = u.m = tcols->len ? RFX_AS_LIST(tcols)[0]->len : 0;
    obj_p newcols[RFX_MAX_AGGS] = {0};
    int64_t mnames[RFX_MAX_AGGS];
    obj_p mexpr[RFX_MAX_AGGS];
    int nmap = 0, st = UPD_GO;
    for (int64_t i = 0; i < dkeys->len && st == UPD_GO; i++) {
        const int64_t k = RFX_AS_I64(dkeys)[i];
        if (k == s_from || k == s_where || k == s_by) continue;
        if (nmap >= RFX_MAX_AGGS) { st = upd_back(&u, "more than 8 mappings"); break; }
        mnames[nmap] = k;
        mexpr[nmap++] = RFX_AS_LIST(dvals)[i];
    }
    if (st == UPD_GO && nmap == 0) st = upd_back(&u, "no mapping");
    if (st == UPD_GO && u.nrows == 0) st = upd_back(&u, "empty table");
    for (int64_t i = 0; i < tcols->len && st == UPD_GO; i++)
        if (RFX_AS_LIST(tcols)[i]->len != u.nrows) st = upd_back(&u, "ragged table");
    if (st == UPD_GO && ensure_ctx() != RFX_OK) st = upd_stop(&u, fail_hip("no usable MI355X"));
    if (st == UPD_GO && g_nshards > 1) {
        st = upd_sharded(&u, where, by, nmap, mnames, mexpr, newcols);
        if (st == UPD_BACK) g_refused_sharded = 1;
        nmap = st == UPD_GO ? nmap : nmap; /* (newcols[] are dropped below unless the result takes them) */
    } else {
    if (st == UPD_GO && where) st = upd_where(&u, where, by);
    if (st == UPD_GO && by) st = upd_by(&u, by);
    }
    for (int i = 0; i < nmap && st == UPD_GO && g_nshards == 1; i++) {
        upd_val_t v;
        st = upd_value(&u, mexpr[i], by, &v);
        if (st == UPD_GO) st = upd_column(&u, table_col(tab, mnames[i]), by, &v, &newcols[i]);
    }
    obj_p res;
    if (st == UPD_GO) {
        res = upd_result(tab, mnames, newcols, nmap);
        g_last_gpu = 1;
    } else if (st == UPD_STOP) res = u.res;
    else { /* the host's: nothing of ours is left behind */
        for (int i = 0; i < u.ntmp; i++) rfx_hip_free(g_ctx, u.tmp[i]);
        u.ntmp = 0;
        qtmp_release();
        res = delegate_update(dict, u.why ? u.why : "unsupported");
    }
    for (int i = 0; i < nmap && i < RFX_MAX_AGGS; i++)
        if (newcols[i]) H.drop(newcols[i]);
    for (int i = 0; i < u.ntmp; i++) rfx_hip_free(g_ctx, u.tmp[i]);
    qtmp_release();
    H.drop(tab);
    return res;
}
rfx_obj_p rfx_update(rfx_obj_p dict) {
    op_begin();
    g_last_gpu = 0;
    obj_p r = update_impl(dict);
    op_end();
    return r;
}
