/* rfx_ops_folds.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * at, aggregates over vectors / lazy MAPFILTER / MAPGROUP pairs, rfx_group. */
/* ---- grouped aggregates over a lazy MAPGROUP (val, index) pair (core/group.c:26-46; aggr_sum(val, index) ... core/aggr.c:1078-2063) ----
 * `index` is the reference's 7-slot group index (index_group_build, core/index.c:1696-1699):
 *   [0] type  [1] group count  [2] group ids  [3] shift  [4] source column  [5] filter ids  [6] first ids
 * INDEX_TYPE_IDS:   row i (= position in the filter, if any) belongs to group [2][i]; value row x = filter ? filter[i] : i
 * INDEX_TYPE_SHIFT: [2] is the key TABLE (slot -> group id) and the group of row x is [2][source[x] - shift]
 * (AGGR_ITER, core/aggr.c:73-161).  On the device both are a dense group-by: IDS keyed by the id column over [0, groups), SHIFT keyed
 * by the source column over [shift, shift + table length) -- whose first-occurrence ranking reproduces the table's ids, so the table
 * itself is not even read.  Under a filter the value (and source) column is gathered by the filter ids first.  The parted /
 * window index flavours (core/aggr.c:126-159) go back to the host's own aggregate. */
static obj_p fold_mapgroup(int f, int kind, obj_p x) {
    obj_p val = RFX_AS_LIST(x)[0], index = RFX_AS_LIST(x)[1];
    const char *why = NULL;
    void *tmp[8];
    int ntmp = 0;
    obj_p res = NULL;
    if (!index || index->type != RFX_TYPE_LIST || index->len != 7) return fail("aggregate: malformed group index");
    obj_p *ix = RFX_AS_LIST(index);
    const int64_t itype = ix[0]->i64, groups = ix[1]->i64;
    obj_p gids = ix[2], source = ix[4], filter = ix[5];
    if (!(val->type > 0 && col_ctype(val) && val->type != RFX_TYPE_SYMBOL)) { why = "value column type"; goto out; }
    if (itype != RFX_INDEX_TYPE_IDS && itype != RFX_INDEX_TYPE_SHIFT) { why = "parted / window index"; goto out; }
    if (!gids || gids->type != RFX_TYPE_I64 || groups < 0) { why = "group ids"; goto out; }
    if (itype == RFX_INDEX_TYPE_SHIFT && !(source && source->type > 0 && col_ctype(source) == RFX_I64)) { why = "source column"; goto out; }
    const int filtered = filter && filter->type == RFX_TYPE_I64;
    const int64_t n = filtered ? filter->len : (itype == RFX_INDEX_TYPE_IDS ? gids->len : source->len);
    if (itype == RFX_INDEX_TYPE_IDS && gids->len != n) return fail("aggregate: group ids / filter length mismatch");
    if (!filtered && val->len != n) return fail("length");
    const int out_f64 = kind == RFX_AGG_AVG || (kind != RFX_AGG_COUNT && col_ctype(val) == RFX_F64);
    if (groups == 0 || n == 0) return H.vector(out_f64 ? RFX_TYPE_F64 : RFX_TYPE_I64, 0);
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1) {
        /* over the shards: a dense group-by keyed by the index's id column (IDS) or its source column (SHIFT), planned like any by: -- every
         * shard scatters its rows, the tables merge, the groups come out in first-occurrence order = the index's group ids.  A filtered index
         * aligns its ids with filter positions, not rows: the host's own aggregate. */
        if (filtered) { g_refused_sharded = 1; return refused1(f, x); }
        const void *dvs = NULL, *dks = NULL;
        int rc = resident(val, 0, &dvs);
        if (rc == RFX_OK) rc = itype == RFX_INDEX_TYPE_IDS ? transient_sharded(gids, &dks) : resident(source, 0, &dks);
        if (rc != RFX_OK) { qtmp_release(); return fail_hip("column upload"); }
        rfx_agg_t as;
        memset(&as, 0, sizeof(as));
        as.d_col = dvs;
        as.col_type = col_ctype(val);
        as.kind = kind;
        rfx_query_t Q;
        memset(&Q, 0, sizeof(Q));
        const void *dkeys[1] = {dks};
        Q.aggs = &as;
        Q.nagg = 1;
        Q.logic = RFX_AND;
        Q.nkeys = 1;
        Q.d_keys = dkeys;
        Q.nrows = n;
        Q.cols = g_qcols;
        Q.ncols = g_nqcols;
        Q.flags = RFX_Q_SLICED;
        rfx_groups_t R;
        rc = rfx_exec_group_by(g_x, &Q, &R);
        if (rc != RFX_OK) { qtmp_release(); return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error()); }
        if (R.groups != groups) {
            rfx_exec_groups_free(g_x, &R);
            qtmp_release();
            why = "group count of the index does not match its rows";
            goto out;
        }
        obj_p outv = H.vector(out_f64 ? RFX_TYPE_F64 : RFX_TYPE_I64, groups);
        const void *srcs[1] = {R.d_results[0]};
        void *dsts[1] = {RFX_AS_RAW(outv)};
        rc = rfx_exec_groups_fetch_all(g_x, &R, 1, srcs, dsts);
        rfx_exec_groups_free(g_x, &R);
        qtmp_release();
        if (rc != RFX_OK) { H.drop(outv); return fail_hip("group emit"); }
        return outv;
    }
    {
        const void *dv = NULL, *dk = NULL, *dfl = NULL;
        if (resident(val, 0, &dv) != RFX_OK) { res = fail_hip("column upload"); goto done; }
        if (filtered) {
            if (transient(filter, &dfl) != RFX_OK) { res = fail_hip("filter upload"); goto done; }
            void *g = NULL;
            if (rfx_hip_malloc(g_ctx, &g, (size_t)n * 8) != RFX_OK) { res = fail_hip("scratch"); goto done; }
            tmp[ntmp++] = g;
            if (rfx_hip_gather_checked(g_ctx, dv, val->len, col_ctype(val), (const int64_t *)dfl, n, g) != RFX_OK) { res = fail_hip("gather"); goto done; }
            dv = g;
        }
        int64_t kmin = 0, range = groups;
        if (itype == RFX_INDEX_TYPE_IDS) {
            if (transient(gids, &dk) != RFX_OK) { res = fail_hip("group ids upload"); goto done; }
        } else {
            kmin = ix[3]->i64;
            range = gids->len;
            if (resident(source, 0, &dk) != RFX_OK) { res = fail_hip("column upload"); goto done; }
            if (filtered) {
                void *g = NULL;
                if (rfx_hip_malloc(g_ctx, &g, (size_t)n * 8) != RFX_OK) { res = fail_hip("scratch"); goto done; }
                tmp[ntmp++] = g;
                if (rfx_hip_gather_checked(g_ctx, dk, source->len, RFX_I64, (const int64_t *)dfl, n, g) != RFX_OK) { res = fail_hip("gather"); goto done; }
                dk = g;
            }
        }
        if (range <= 0) { why = "empty key table"; goto out; }
        rfx_agg_t a;
        memset(&a, 0, sizeof(a));
        a.d_col = dv;
        a.col_type = col_ctype(val);
        a.kind = kind;
        int narr = 0;
        rfx_hip_group_table_arrays(&a, 1, &narr);
        void *store = NULL;
        if (rfx_hip_malloc(g_ctx, &store, (size_t)narr * (size_t)range * 8) != RFX_OK) { res = fail_hip("group tables"); goto done; }
        tmp[ntmp++] = store;
        int64_t *base = (int64_t *)store;
        rfx_group_tables_t gt;
        memset(&gt, 0, sizeof(gt));
        gt.kmin = kmin;
        gt.range = range;
        gt.nagg = 1;
        gt.d_first = base;
        gt.d_acc[0] = base + range;
        gt.d_cnt[0] = narr > 2 ? base + 2 * range : NULL;
        int64_t ng = 0;
        if (rfx_hip_group_tables_init(g_ctx, &a, &gt) != RFX_OK || rfx_hip_group_dense_accumulate(g_ctx, (const int64_t *)dk, NULL, 0, RFX_AND, &a, n, 0, &gt) != RFX_OK ||
            rfx_hip_group_rank(g_ctx, &gt, n, &ng) != RFX_OK) { res = fail_hip("group-by over the index"); goto done; }
        if (ng != groups) { why = "group count of the index does not match its rows"; goto out; }
        void *dout = NULL;
        if (rfx_hip_malloc(g_ctx, &dout, (size_t)groups * 8) != RFX_OK) { res = fail_hip("result"); goto done; }
        tmp[ntmp++] = dout;
        void *ptrs[1] = {dout};
        obj_p out = H.vector(out_f64 ? RFX_TYPE_F64 : RFX_TYPE_I64, groups);
        if (rfx_hip_group_emit(g_ctx, &a, &gt, NULL, NULL, ptrs) != RFX_OK || rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dout, (size_t)groups * 8) != RFX_OK) {
            H.drop(out);
            res = fail_hip("group emit");
            goto done;
        }
        res = out;
        goto done;
    }
out:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    ntmp = 0;
    if (H.bound == 1 && H.f[f]) res = HOST_CALL(((rfx_unary_f)H.f[f])(x));
    else {
        char b[256];
        snprintf(b, sizeof(b), "aggregate over a MAPGROUP pair: not covered by the MI355X path (%s) and no host function to delegate to", why ? why : "unsupported");
        res = fail(b);
    }
done:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    return res;
}

/* (rfx_group keys): the reference's group index of an I64 key column (index_group_i64_scoped, core/index.c:2002-2092) built on the
 * device: first-occurrence table (K7), rank (K8), then either the key table (INDEX_TYPE_SHIFT, range <= INDEX_SCOPE_LIMIT = 524 288)
 * or the per-row id vector (INDEX_TYPE_IDS).  Slots as index_group_build lays them out; sparse keys (range > rows) are the host's. */
static obj_p group_impl(obj_p keys) {
    rfx_host_bind();
    if (!keys || keys->type <= 0 || col_ctype(keys) != RFX_I64) return fail("group: expected an i64-like vector");
    const int64_t n = keys->len;
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1 && n > 0) {
        /* over the shards: the planner's group-by without aggregates gives the groups' keys and first rows in first-occurrence order; the key-table form
         * of the index (INDEX_TYPE_SHIFT, range <= INDEX_SCOPE_LIMIT) is then a table of at most 524 288 cells built from them; the per-row form
         * (INDEX_TYPE_IDS) maps every row through that table where the rows live.  Sparse or null keys: the host's own `group` */
        const void *dks = NULL;
        if (resident(keys, 0, &dks) != RFX_OK) return fail_hip("column upload");
        const void *dkeys[1] = {dks};
        rfx_query_t Q;
        memset(&Q, 0, sizeof(Q));
        Q.logic = RFX_AND;
        Q.nkeys = 1;
        Q.d_keys = dkeys;
        Q.nrows = n;
        Q.cols = g_qcols;
        Q.ncols = g_nqcols;
        Q.flags = RFX_Q_WANT_FIRST | RFX_Q_REFUSE_NULL_KEY;
        rfx_groups_t R;
        const int grc = rfx_exec_group_by(g_x, &Q, &R);
        if (grc != RFX_OK && grc != RFX_EXEC_NULL_KEY) return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
        obj_p res_s = NULL;
        if (grc == RFX_OK && (R.path == RFX_PATH_DENSE || R.path == RFX_PATH_DENSE_SMALL) && R.groups > 0) {
            obj_p gk = H.vector(RFX_TYPE_I64, R.groups), firsts_s = H.vector(RFX_TYPE_I64, R.groups);
            const void *srcs[2] = {R.d_keys, R.d_first};
            void *dsts[2] = {RFX_AS_RAW(gk), RFX_AS_RAW(firsts_s)};
            int ok = rfx_exec_groups_fetch_all(g_x, &R, 2, srcs, dsts) == RFX_OK;
            int64_t lo = RFX_INF_I64, hi = RFX_NULL_I64;
            for (int64_t g = 0; g < R.groups && ok; g++) {
                const int64_t kk = RFX_AS_I64(gk)[g];
                lo = kk < lo ? kk : lo;
                hi = kk > hi ? kk : hi;
            }
            const uint64_t range_s = ok ? (uint64_t)hi - (uint64_t)lo + 1 : 0;
            if (ok && range_s != 0 && range_s <= (uint64_t)n) {
                obj_p table = H.vector(RFX_TYPE_I64, (int64_t)range_s);
                for (uint64_t i = 0; i < range_s; i++) RFX_AS_I64(table)[i] = RFX_NULL_I64;
                for (int64_t g = 0; g < R.groups; g++) RFX_AS_I64(table)[RFX_AS_I64(gk)[g] - lo] = g;
                const int shift_s = range_s <= RFX_INDEX_SCOPE_LIMIT;
                obj_p ids_s = NULL;
                if (!shift_s) { /* INDEX_TYPE_IDS: every row's group id, computed where the rows live through the table (uploaded to every shard) */
                    ids_s = H.vector(RFX_TYPE_I64, n);
                    void *dt[RFX_MAX_SHARDS] = {0}, *dg[RFX_MAX_SHARDS] = {0};
                    int rc2 = RFX_OK;
                    for (int sh = 0; sh < g_nshards && rc2 == RFX_OK; sh++) {
                        int64_t r0, len;
                        rfx_exec_split(n, g_nshards, sh, &r0, &len);
                        if (len <= 0) continue;
                        const void *piece = dks;
                        for (int k = 0; k < g_nqcols && sh > 0; k++)
                            if (g_qcols[k].d[0] == dks) piece = g_qcols[k].d[sh];
                        rfx_hip_ctx_bind_thread(g_ctxs[sh]);
                        rc2 = rfx_hip_malloc(g_ctxs[sh], &dt[sh], (size_t)range_s * 8);
                        if (rc2 == RFX_OK) rc2 = rfx_hip_malloc(g_ctxs[sh], &dg[sh], (size_t)len * 8);
                        if (rc2 == RFX_OK) rc2 = rfx_hip_h2d_pipelined(g_ctxs[sh], dt[sh], RFX_AS_RAW(table), (size_t)range_s * 8);
                        if (rc2 == RFX_OK) rc2 = rfx_hip_group_ids_table(g_ctxs[sh], (const int64_t *)piece, len, lo, (int64_t)range_s, (const int64_t *)dt[sh], (int64_t *)dg[sh]);
                        if (rc2 == RFX_OK) rc2 = rfx_hip_d2h_pipelined(g_ctxs[sh], (char *)RFX_AS_RAW(ids_s) + (size_t)r0 * 8, dg[sh], (size_t)len * 8);
                    }
                    for (int sh = 0; sh < g_nshards; sh++) {
                        if (!dt[sh] && !dg[sh]) continue;
                        rfx_hip_ctx_bind_thread(g_ctxs[sh]);
                        rfx_hip_ctx_sync(g_ctxs[sh]);
                        if (dt[sh]) rfx_hip_free(g_ctxs[sh], dt[sh]);
                        if (dg[sh]) rfx_hip_free(g_ctxs[sh], dg[sh]);
                    }
                    rfx_hip_ctx_bind_thread(g_ctx);
                    H.drop(table);
                    table = NULL;
                    if (rc2 != RFX_OK) { H.drop(ids_s); ids_s = NULL; }
                }
                if (shift_s || ids_s) {
                    res_s = H.vector(RFX_TYPE_LIST, 7);
                    obj_p *ixs = RFX_AS_LIST(res_s);
                    ixs[0] = H.i64(shift_s ? RFX_INDEX_TYPE_SHIFT : RFX_INDEX_TYPE_IDS);
                    ixs[1] = H.i64(R.groups);
                    ixs[2] = shift_s ? table : ids_s;
                    ixs[3] = H.i64(shift_s ? lo : RFX_NULL_I64);
                    ixs[4] = shift_s ? H.clone(keys) : H.null_obj;
                    ixs[5] = H.null_obj;
                    ixs[6] = firsts_s;
                    firsts_s = NULL;
                }
            }
            H.drop(gk);
            if (firsts_s) H.drop(firsts_s);
        }
        if (grc == RFX_OK) rfx_exec_groups_free(g_x, &R);
        if (res_s) return res_s;
        g_refused_sharded = 1; /* sparse or null keys: the host's own */
        return (H.bound == 1 && g_host_group) ? HOST_CALL(((rfx_unary_f)g_host_group)(keys)) : fail_ctx();
    }
    const void *dk = NULL;
    if (n && resident(keys, 0, &dk) != RFX_OK) return fail_hip("column upload");
    int64_t kmin = 0, kmax = -1, seen = 0;
    if (n && rfx_hip_scope_i64(g_ctx, (const int64_t *)dk, NULL, 0, RFX_AND, n, &kmin, &kmax, &seen) != RFX_OK) return fail_hip("scope");
    const uint64_t range = n ? (uint64_t)kmax - (uint64_t)kmin + 1 : 0;
    if (n && kmin == RFX_NULL_I64) return fail("group: null keys are not built on the MI355X path"); /* (every null row its own group: core/index.c:1808-1816) */
    if (n && !(range != 0 && range <= (uint64_t)n)) {
        /* SPARSE keys (round 6): index_group_i64_unscoped (core/index.c:1959-1977) -> index_group_distribute (:1777-1911): the IDS flavour with NO first
         * rows -- [IDS, groups, per-row ids, null, null, filter, null].  The planner's hashed group-by (K9) without aggregates gives the groups in
         * first-occurrence order (= the reference with one executor, what the goldens pin; with several its ids follow its chunks' table order) and,
         * per row, its group's first row; rfx_hip_group_ids_first turns those into the per-row ids. */
        const void *dkeys[1] = {dk};
        rfx_query_t Q;
        memset(&Q, 0, sizeof(Q));
        Q.logic = RFX_AND;
        Q.nkeys = 1;
        Q.d_keys = dkeys;
        Q.nrows = n;
        Q.cols = g_qcols;
        Q.ncols = g_nqcols;
        Q.flags = RFX_Q_WANT_FIRST | RFX_Q_REFUSE_NULL_KEY | RFX_Q_PROBE_FIRST;
        rfx_groups_t R;
        const int grc = rfx_exec_group_by(g_x, &Q, &R);
        if (grc != RFX_OK) return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
        obj_p res_h = NULL;
        void *dg = NULL;
        if (R.path == RFX_PATH_HASH && R.groups > 0 && R.d_probe && R.d_first && rfx_hip_malloc(g_ctx, &dg, (size_t)n * 8) == RFX_OK &&
            rfx_hip_group_ids_first(g_ctx, R.d_probe, n, R.d_first, R.groups, (int64_t *)dg) == RFX_OK) {
            obj_p ids_h = H.vector(RFX_TYPE_I64, n);
            if (rfx_hip_d2h_pipelined(g_ctx, RFX_AS_RAW(ids_h), dg, (size_t)n * 8) == RFX_OK) {
                res_h = H.vector(RFX_TYPE_LIST, 7);
                obj_p *ixh = RFX_AS_LIST(res_h);
                ixh[0] = H.i64(RFX_INDEX_TYPE_IDS);
                ixh[1] = H.i64(R.groups);
                ixh[2] = ids_h;
                ixh[3] = H.i64(RFX_NULL_I64);
                ixh[4] = H.null_obj;
                ixh[5] = H.null_obj;
                ixh[6] = H.null_obj;
            } else H.drop(ids_h);
        }
        if (dg) rfx_hip_free(g_ctx, dg);
        rfx_exec_groups_free(g_x, &R);
        return res_h ? res_h : fail_hip("group index over sparse keys");
    }
    void *store = NULL, *dfirst = NULL, *dids = NULL;
    obj_p res = NULL, gids = NULL, firsts = NULL;
    int64_t groups = 0;
    const int shift_form = range <= RFX_INDEX_SCOPE_LIMIT;
    if (n) {
        rfx_agg_t none;
        memset(&none, 0, sizeof(none));
        rfx_group_tables_t gt;
        memset(&gt, 0, sizeof(gt));
        if (rfx_hip_malloc(g_ctx, &store, (size_t)range * 8) != RFX_OK) return fail_hip("group tables");
        gt.kmin = kmin;
        gt.range = (int64_t)range;
        gt.nagg = 0;
        gt.d_first = (int64_t *)store;
        int ok = rfx_hip_group_tables_init(g_ctx, &none, &gt) == RFX_OK &&
                 rfx_hip_group_dense_accumulate(g_ctx, (const int64_t *)dk, NULL, 0, RFX_AND, &none, n, 0, &gt) == RFX_OK &&
                 rfx_hip_group_rank(g_ctx, &gt, n, &groups) == RFX_OK;
        ok = ok && rfx_hip_malloc(g_ctx, &dfirst, (size_t)(groups ? groups : 1) * 8) == RFX_OK &&
             rfx_hip_group_emit(g_ctx, &none, &gt, NULL, (int64_t *)dfirst, NULL) == RFX_OK;
        const int64_t nid = shift_form ? (int64_t)range : n;
        ok = ok && rfx_hip_malloc(g_ctx, &dids, (size_t)nid * 8) == RFX_OK &&
             (shift_form ? rfx_hip_group_slot_ids(g_ctx, &gt, (int64_t *)dids) : rfx_hip_group_ids_dense(g_ctx, (const int64_t *)dk, n, &gt, (int64_t *)dids)) == RFX_OK;
        if (ok) {
            gids = H.vector(RFX_TYPE_I64, nid);
            firsts = H.vector(RFX_TYPE_I64, groups);
            ok = rfx_hip_d2h(g_ctx, RFX_AS_RAW(gids), dids, (size_t)nid * 8) == RFX_OK && (groups == 0 || rfx_hip_d2h(g_ctx, RFX_AS_RAW(firsts), dfirst, (size_t)groups * 8) == RFX_OK);
        }
        if (store) rfx_hip_free(g_ctx, store);
        if (dfirst) rfx_hip_free(g_ctx, dfirst);
        if (dids) rfx_hip_free(g_ctx, dids);
        if (!ok) {
            if (gids) H.drop(gids);
            if (firsts) H.drop(firsts);
            return fail_hip("group index");
        }
    } else {
        gids = H.vector(RFX_TYPE_I64, 0);
        firsts = H.vector(RFX_TYPE_I64, 0);
    }
    res = H.vector(RFX_TYPE_LIST, 7);
    obj_p *ix = RFX_AS_LIST(res);
    ix[0] = H.i64(shift_form && n ? RFX_INDEX_TYPE_SHIFT : RFX_INDEX_TYPE_IDS);
    ix[1] = H.i64(groups);
    ix[2] = gids;
    ix[3] = H.i64(shift_form && n ? kmin : RFX_NULL_I64);
    ix[4] = shift_form && n ? H.clone(keys) : H.null_obj; /* NULL_OBJ is the host's static null, passed as index_group_build passes it */
    ix[5] = H.null_obj;
    ix[6] = firsts;
    return res;
}
rfx_obj_p rfx_group(rfx_obj_p keys) {
    op_begin();
    obj_p r = group_impl(keys);
    op_end();
    return r;
}

/* scalar aggregates of a vector or of a lazy MAPFILTER (val, ids) pair (core/filter.c:29-49, core/math.c:1874-1890) */
static obj_p fold_impl(int f, int kind, obj_p x) {
    rfx_host_bind();
    if (!x) return fail("aggregate: null argument");
    if (x->type == RFX_TYPE_MAPGROUP) return fold_mapgroup(f, kind, x);
    if (x->type == RFX_TYPE_MAPFILTER) {
        /* the lazy (val, ids) pair an FN_AGGR built-in receives (core/eval.c:723-728): gather on the device, fold there --
         * the filtered vector the reference would materialise (filter_collect) never exists on the host */
        obj_p val = RFX_AS_LIST(x)[0], ids = RFX_AS_LIST(x)[1];
        if (!(val->type > 0 && col_ctype(val) && val->type != RFX_TYPE_SYMBOL) || ids->type != RFX_TYPE_I64) {
            if (H.bound == 1 && H.f[f]) return HOST_CALL(((rfx_unary_f)H.f[f])(x));
            return fail("aggregate: only (i64/f64 vector, i64 ids) MAPFILTER pairs run on the MI355X path");
        }
        if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
        if (g_nshards > 1) { /* every shard gathers and folds the ids inside its rows; partials folded in shard order (rfx_exec_filter_aggr) */
            const void *dvs;
            const int64_t *dsel[RFX_MAX_SHARDS];
            int64_t nsel[RFX_MAX_SHARDS];
            rfx_agg_t as;
            rfx_value_t vs;
            if (resident(val, 0, &dvs) != RFX_OK) return fail_hip("column upload");
            int rc = sel_ids_sharded(ids, val->len, dsel, nsel);
            if (rc == 1) { /* ids that do not ascend through the shards' row ranges: not a filter's -- the host's own aggregate */
                qtmp_release();
                g_refused_sharded = 1;
                return refused1(f, x);
            }
            memset(&as, 0, sizeof(as));
            as.d_col = dvs;
            as.col_type = col_ctype(val);
            as.kind = kind;
            if (rc == RFX_OK) rc = fold_sharded(&as, val->len, dsel, nsel, &vs);
            qtmp_release();
            if (rc != RFX_OK) return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
            return value_atom(&vs);
        }
        const void *dv, *di;
        if (resident(val, 0, &dv) != RFX_OK || transient(ids, &di) != RFX_OK) return fail_hip("column upload");
        void *dg = NULL;
        rfx_agg_t a;
        memset(&a, 0, sizeof(a));
        a.col_type = col_ctype(val);
        a.kind = kind;
        rfx_value_t v;
        int ok = rfx_hip_malloc(g_ctx, &dg, (size_t)(ids->len ? ids->len : 1) * 8) == RFX_OK &&
                 rfx_hip_gather_checked(g_ctx, dv, val->len, col_ctype(val), (const int64_t *)di, ids->len, dg) == RFX_OK;
        a.d_col = dg;
        ok = ok && rfx_hip_filter_aggr_host(g_ctx, NULL, 0, RFX_AND, &a, 1, ids->len, &v, NULL) == RFX_OK;
        if (dg) rfx_hip_free(g_ctx, dg);
        if (!ok) return fail_hip("filter_aggr over a MAPFILTER");
        return value_atom(&v);
    }
    if (!(x->type > 0 && col_ctype(x) && x->type != RFX_TYPE_SYMBOL)) {
        if (H.bound == 1 && H.f[f]) return HOST_CALL(((rfx_unary_f)H.f[f])(x));
        return fail("aggregate: only i64/f64 vectors run on the MI355X path");
    }
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    const void *d;
    if (resident(x, 0, &d) != RFX_OK) return fail_hip("column upload");
    rfx_agg_t a;
    memset(&a, 0, sizeof(a));
    a.d_col = d;
    a.col_type = col_ctype(x);
    a.kind = kind;
    rfx_value_t v;
    if (g_nshards > 1) { /* the fold on every shard, the partials in shard order (unop_fold's two levels, core/math.c:2176-2231) */
        if (fold_sharded(&a, x->len, NULL, NULL, &v) != RFX_OK) return fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
        return value_atom(&v);
    }
    if (rfx_hip_filter_aggr_host(g_ctx, NULL, 0, RFX_AND, &a, 1, x->len, &v, NULL) != RFX_OK) return fail_hip("filter_aggr");
    return value_atom(&v);
}
static obj_p fold_op(int f, int kind, obj_p x) {
    op_begin();
    obj_p r = fold_impl(f, kind, x);
    op_end();
    return r;
}
rfx_obj_p rfx_sum(rfx_obj_p x) { return fold_op(F_SUM, RFX_AGG_SUM, x); }
rfx_obj_p rfx_avg(rfx_obj_p x) { return fold_op(F_AVG, RFX_AGG_AVG, x); }
rfx_obj_p rfx_min(rfx_obj_p x) { return fold_op(F_MIN, RFX_AGG_MIN, x); }
rfx_obj_p rfx_max(rfx_obj_p x) { return fold_op(F_MAX, RFX_AGG_MAX, x); }
rfx_obj_p rfx_count(rfx_obj_p x) { return fold_op(F_COUNT, RFX_AGG_COUNT, x); }
rfx_obj_p rfx_first(rfx_obj_p x) { return fold_op(F_FIRST, RFX_AGG_FIRST, x); }
