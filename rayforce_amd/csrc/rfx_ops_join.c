/* rfx_ops_join.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * equi-joins. */
/* ------------------------------------------------------------------------------------------------ equi-joins
 * (left-join [keys] x y) / (inner-join [keys] x y): ray_left_join / ray_inner_join, core/join.c:158-298 -- vary_f over (key symbols,
 * left table, right table).  Index = per left row the first right row with an equal key tuple (index_left_join_obj,
 * core/index.c:2886-2928): the group-by's first-occurrence table over the right keys (zero aggregates), probed with the left keys. */
/* ---- over SHARDS (round 6): a broadcast join.  The left table stays where its rows are (row ranges, one per shard); the right table's key columns and
 * every right column the result carries are kept WHOLE on every shard's device (resident_ex(whole): cached like any column, uploaded through all the
 * PCIe links at once); every shard builds the same first-occurrence table over the right keys, probes ITS left rows (rfx_exec_join_index_shard: global
 * right row ids), gathers the right columns at those ids and writes its rows of the result -- no exchange between the shards, the result's row order is the
 * left table's (index_left_join_obj / index_inner_join_obj, core/index.c:2886-2990; the reference runs the same probe over its pool, core/join.c:158-298). */
typedef struct {
    int inner, nk, collision[RFX_MAX_SHARDS];
    int64_t nl, nr;
    const void *dlk[RFX_MAX_SHARDS][RFX_MAX_KEYS], *drk[RFX_MAX_SHARDS][RFX_MAX_KEYS];
    void *ids[RFX_MAX_SHARDS], *lids[RFX_MAX_SHARDS], *rids[RFX_MAX_SHARDS], *dcol[RFX_MAX_SHARDS];
    int64_t nout[RFX_MAX_SHARDS], off[RFX_MAX_SHARDS];
    /* the column being assembled */
    const void *src[RFX_MAX_SHARDS], *left[RFX_MAX_SHARDS];
    int src_is_right;
    uint64_t nullbits;
    char *out;
} jsh_t;
static int jsh_index(void *arg, int s) {
    jsh_t *J = (jsh_t *)arg;
    int64_t r0, n;
    rfx_exec_split(J->nl, g_nshards, s, &r0, &n);
    J->nout[s] = J->inner ? 0 : n;
    if (n <= 0) return RFX_OK;
    rfx_ctx_t *c = g_ctxs[s];
    int rc = rfx_hip_malloc(c, &J->ids[s], (size_t)n * 8);
    if (rc == RFX_OK) rc = rfx_exec_join_index_shard(g_x, s, J->dlk[s], J->drk[s], J->nk, n, J->nr, (int64_t *)J->ids[s], &J->collision[s]);
    if (rc == RFX_OK && J->inner) { /* this shard's matched left rows (local positions, in order) and their right rows */
        rfx_pred_t p;
        memset(&p, 0, sizeof(p));
        p.d_col = J->ids[s]; p.col_type = RFX_I64; p.op = RFX_NE; p.rhs_type = RFX_I64; p.rhs_i = RFX_NULL_I64;
        int64_t m = 0;
        rc = rfx_hip_where_begin(c, &p, 1, RFX_AND, NULL, n, &m);
        if (rc == RFX_OK) rc = rfx_hip_malloc(c, &J->lids[s], (size_t)(m ? m : 1) * 8);
        if (rc == RFX_OK) rc = rfx_hip_malloc(c, &J->rids[s], (size_t)(m ? m : 1) * 8);
        if (rc == RFX_OK) rc = rfx_hip_where_emit(c, 0, (int64_t *)J->lids[s]);
        if (rc == RFX_OK && m) rc = rfx_hip_gather(c, J->ids[s], (const int64_t *)J->lids[s], m, J->rids[s]);
        J->nout[s] = m;
    }
    if (rc == RFX_OK) rc = rfx_hip_malloc(c, &J->dcol[s], (size_t)(J->nout[s] ? J->nout[s] : 1) * 8);
    return rc;
}
static int jsh_column(void *arg, int s) {
    jsh_t *J = (jsh_t *)arg;
    const int64_t m = J->nout[s];
    if (m <= 0) return RFX_OK;
    rfx_ctx_t *c = g_ctxs[s];
    int rc;
    if (J->inner) rc = rfx_hip_gather(c, J->src[s], (const int64_t *)(J->src_is_right ? J->rids[s] : J->lids[s]), m, J->dcol[s]);
    else rc = rfx_hip_gather_or(c, J->src[s], J->left[s], (const int64_t *)J->ids[s], m, J->nullbits, J->dcol[s]);
    if (rc == RFX_OK) rc = rfx_hip_d2h(c, J->out + (size_t)J->off[s] * 8, J->dcol[s], (size_t)m * 8);
    return rc;
}
static int jsh_release(void *arg, int s) {
    jsh_t *J = (jsh_t *)arg;
    void *p[4] = {J->ids[s], J->lids[s], J->rids[s], J->dcol[s]};
    for (int i = 0; i < 4; i++)
        if (p[i]) rfx_hip_free(g_ctxs[s], p[i]);
    J->ids[s] = J->lids[s] = J->rids[s] = J->dcol[s] = NULL;
    return RFX_OK;
}
/* 1: answered (*res), 0: not this path's (the caller hands the join to the host), -1: failed (*res = the error object) */
static int join_sharded(int inner, obj_p ksyms, obj_p lt, obj_p rt, obj_p *lk, obj_p *rk, int nk, int64_t nl, int64_t nr, obj_p *res) {
    jsh_t *J = (jsh_t *)calloc(1, sizeof(jsh_t));
    if (!J) return 0;
    J->inner = inner; J->nk = nk; J->nl = nl; J->nr = nr;
    obj_p lnames = RFX_AS_LIST(lt)[0], rnames = RFX_AS_LIST(rt)[0];
    int rcode = 0;
    void *devs[RFX_MAX_SHARDS];
    const void *d0;
    for (int i = 0; i < nk; i++) {
        if (resident_ex(lk[i], 0, 0, &d0, devs) != RFX_OK) goto done; /* (a device handle without per-shard pieces, a parted column: not this path's) */
        for (int sh = 0; sh < g_nshards; sh++) J->dlk[sh][i] = devs[sh];
        if (resident_ex(rk[i], 0, 1, &d0, devs) != RFX_OK) goto done;
        for (int sh = 0; sh < g_nshards; sh++) J->drk[sh][i] = devs[sh];
    }
    if (rfx_exec_run(g_x, jsh_index, J) != RFX_OK) {
        int coll = 0;
        for (int sh = 0; sh < g_nshards; sh++) coll |= J->collision[sh];
        if (!coll) { *res = fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error()); rcode = -1; }
        goto done; /* (a row-hash collision between two key tuples: the host's own join) */
    }
    {
        int64_t names[64], nout = 0;
        int ncol = 0;
        for (int sh = 0; sh < g_nshards; sh++) {
            int64_t r0;
            rfx_exec_split(nl, g_nshards, sh, &r0, NULL);
            J->off[sh] = inner ? nout : r0;
            nout += J->nout[sh];
        }
        for (int i = 0; i < nk; i++) names[ncol++] = RFX_AS_I64(ksyms)[i];
        for (int pass = 0; pass < 2; pass++) {
            obj_p nm = pass ? rnames : lnames;
            for (int64_t i = 0; i < nm->len && ncol < 64; i++) {
                int64_t sy = RFX_AS_I64(nm)[i];
                int dup = 0;
                for (int j = 0; j < ncol; j++) dup |= names[j] == sy;
                if (!dup) names[ncol++] = sy;
            }
        }
        if (ncol >= 64) goto done;
        obj_p rk_ = H.vector(RFX_TYPE_SYMBOL, ncol), rv = H.vector(RFX_TYPE_LIST, ncol);
        int ok = 1;
        for (int c = 0; c < ncol; c++) { /* (every slot of rv is filled, also after a failure: the list is dropped as a whole) */
            RFX_AS_I64(rk_)[c] = names[c];
            obj_p lc = table_col(lt, names[c]), rc = table_col(rt, names[c]);
            obj_p o = NULL;
            if (!ok) o = NULL;
            else if (!inner && (c < nk || !rc)) o = H.clone(lc); /* left join: key columns and left-only columns are the left table's own */
            else {
                obj_p src = inner ? (rc ? rc : lc) : rc;
                o = H.vector(src->type, nout);
                J->src_is_right = src == rc;
                J->out = (char *)RFX_AS_RAW(o);
                J->nullbits = col_ctype(src) == RFX_F64 ? 0x7FF8000000000000ull : 0x8000000000000000ull;
                ok = resident_ex(src, 0, J->src_is_right, &d0, devs) == RFX_OK;
                for (int sh = 0; sh < g_nshards && ok; sh++) J->src[sh] = devs[sh];
                for (int sh = 0; sh < g_nshards; sh++) J->left[sh] = NULL;
                if (ok && !inner && lc) { /* a column of both tables: the left value where no right row matches */
                    ok = resident_ex(lc, 0, 0, &d0, devs) == RFX_OK;
                    for (int sh = 0; sh < g_nshards && ok; sh++) J->left[sh] = devs[sh];
                }
                if (ok && nout) ok = rfx_exec_run(g_x, jsh_column, J) == RFX_OK;
            }
            RFX_AS_LIST(rv)[c] = o ? o : H.null_obj;
        }
        if (!ok) {
            H.drop(rk_);
            H.drop(rv);
            *res = fail(rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error());
            rcode = -1;
            goto done;
        }
        *res = H.table(rk_, rv);
        rcode = 1;
    }
done:
    rfx_exec_run(g_x, jsh_release, J);
    rfx_hip_ctx_bind_thread(g_ctx);
    free(J);
    return rcode;
}

static obj_p join_impl(int inner, obj_p *x, int64_t n) {
    rfx_host_bind();
    const int fidx = inner ? F_IJ : F_LJ;
    if (n != 3 || !x[0] || !x[1] || !x[2]) return fail("join: expected (keys, left table, right table)");
    if (x[0]->type != RFX_TYPE_SYMBOL || x[1]->type != RFX_TYPE_TABLE || x[2]->type != RFX_TYPE_TABLE) return fail("join: expected (symbol vector, table, table)");
    obj_p ksyms = x[0], lt = x[1], rt = x[2];
    obj_p lnames = RFX_AS_LIST(lt)[0], lcols = RFX_AS_LIST(lt)[1], rnames = RFX_AS_LIST(rt)[0], rcols = RFX_AS_LIST(rt)[1];
    const int64_t nl = lcols->len ? RFX_AS_LIST(lcols)[0]->len : 0, nr = rcols->len ? RFX_AS_LIST(rcols)[0]->len : 0;
    const int nk = (int)ksyms->len;
    const char *why = NULL;
    void *tmp[4 * RFX_MAX_KEYS + 8];
    int ntmp = 0;
    obj_p res = NULL;
    if (nl == 0 || nr == 0) return H.clone(lt); /* core/join.c:171-172 */
    if (nk < 1 || nk > RFX_MAX_KEYS) { why = "1..8 key columns"; goto out; }
    obj_p lk[RFX_MAX_KEYS], rk[RFX_MAX_KEYS];
    const void *dlk[RFX_MAX_KEYS], *drk[RFX_MAX_KEYS];
    for (int i = 0; i < nk; i++) {
        lk[i] = table_col(lt, RFX_AS_I64(ksyms)[i]);
        rk[i] = table_col(rt, RFX_AS_I64(ksyms)[i]);
        if (!lk[i] || !rk[i] || col_ctype(lk[i]) != RFX_I64 || col_ctype(rk[i]) != RFX_I64 || lk[i]->type != rk[i]->type) { why = "join key is not an 8-byte integer column of both tables"; goto out; }
    }
    for (int64_t i = 0; i < lcols->len; i++) if (!col_ctype(RFX_AS_LIST(lcols)[i])) { why = "non-8-byte column"; goto out; }
    for (int64_t i = 0; i < rcols->len; i++) {
        obj_p rc = RFX_AS_LIST(rcols)[i], lc = table_col(lt, RFX_AS_I64(rnames)[i]);
        if (!col_ctype(rc)) { why = "non-8-byte column"; goto out; }
        if (lc && lc->type != rc->type) return fail("join: a column has different types in the two tables"); /* err_type, core/join.c:50-51 */
    }
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1) {
        obj_p r = NULL;
        const int jr = join_sharded(inner, ksyms, lt, rt, lk, rk, nk, nl, nr, &r);
        if (jr > 0) g_last_gpu = 1;
        if (jr != 0) return r;
        g_refused_sharded = 1;
        return refusedn(fidx, x, n);
    }
    for (int i = 0; i < nk; i++)
        if (resident(lk[i], 0, &dlk[i]) != RFX_OK || resident(rk[i], 0, &drk[i]) != RFX_OK) { res = fail_hip("column upload"); goto done; }
#define JOIN_TMP(ptr, bytes) do { ptr = NULL; if (rfx_hip_malloc(g_ctx, &ptr, (bytes)) != RFX_OK) { res = fail_hip("join scratch"); goto done; } tmp[ntmp++] = ptr; } while (0)
    /* the join index -- per left row the first right row with an equal key tuple, or null -- is the planner's (rfx_exec_join_index: dense
     * first-occurrence table or the hashed one, composite key or the reference's row hash + the tuple check) */
    void *ids = NULL;
    JOIN_TMP(ids, (size_t)nl * 8);
    {
        int collision = 0;
        const int jrc = rfx_exec_join_index(g_x, dlk, drk, nk, nl, nr, (int64_t *)ids, &collision);
        if (jrc != RFX_OK && collision) { why = "row-hash collision between two key tuples"; goto out; }
        if (jrc != RFX_OK) { res = fail(rfx_exec_last_error(g_x)); goto done; }
    }
    /* result columns: keys, then the other left columns, then the right-only ones (ray_union / ray_except order, core/join.c:83-156) */
    {
        int64_t names[64];
        int ncol = 0;
        for (int i = 0; i < nk; i++) names[ncol++] = RFX_AS_I64(ksyms)[i];
        for (int pass = 0; pass < 2; pass++) {
            obj_p nm = pass ? rnames : lnames;
            for (int64_t i = 0; i < nm->len && ncol < 64; i++) {
                int64_t sy = RFX_AS_I64(nm)[i];
                int dup = 0;
                for (int j = 0; j < ncol; j++) dup |= names[j] == sy;
                if (!dup) names[ncol++] = sy;
            }
        }
        if (ncol >= 64) { why = "too many columns"; goto out; }
        void *lids = NULL, *rids = NULL, *dcol = NULL;
        int64_t nout = nl;
        if (inner) { /* matched left rows in order, paired with their right rows (index_inner_join_obj) */
            rfx_pred_t p;
            memset(&p, 0, sizeof(p));
            p.d_col = ids; p.col_type = RFX_I64; p.op = RFX_NE; p.rhs_type = RFX_I64; p.rhs_i = RFX_NULL_I64;
            if (rfx_hip_where_begin(g_ctx, &p, 1, RFX_AND, NULL, nl, &nout) != RFX_OK) { res = fail_hip("join where"); goto done; }
            JOIN_TMP(lids, (size_t)(nout ? nout : 1) * 8);
            JOIN_TMP(rids, (size_t)(nout ? nout : 1) * 8);
            if (rfx_hip_where_emit(g_ctx, 0, (int64_t *)lids) != RFX_OK || (nout && rfx_hip_gather(g_ctx, ids, (const int64_t *)lids, nout, rids) != RFX_OK)) { res = fail_hip("join where"); goto done; }
        }
        JOIN_TMP(dcol, (size_t)(nout ? nout : 1) * 8);
        obj_p rk_ = H.vector(RFX_TYPE_SYMBOL, ncol), rv = H.vector(RFX_TYPE_LIST, ncol);
        int ok = 1;
        for (int c = 0; c < ncol; c++) {
            RFX_AS_I64(rk_)[c] = names[c];
            obj_p lc = table_col(lt, names[c]), rc = table_col(rt, names[c]);
            const int iskey = c < nk;
            obj_p o = NULL;
            if (!inner && (iskey || !rc)) o = H.clone(lc); /* left join: key columns and left-only columns are the left table's own */
            else {
                obj_p src = (inner ? (rc ? rc : lc) : rc);
                o = H.vector(src->type, nout);
                const void *dsrc, *dleft = NULL;
                ok = ok && resident(src, 0, &dsrc) == RFX_OK;
                if (ok && !inner && lc) ok = resident(lc, 0, &dleft) == RFX_OK;
                if (ok && nout) {
                    if (inner) ok = rfx_hip_gather(g_ctx, dsrc, (const int64_t *)(rc ? rids : lids), nout, dcol) == RFX_OK;
                    else ok = rfx_hip_gather_or(g_ctx, dsrc, dleft, (const int64_t *)ids, nout, col_ctype(src) == RFX_F64 ? 0x7FF8000000000000ull : 0x8000000000000000ull, dcol) == RFX_OK;
                    ok = ok && rfx_hip_d2h(g_ctx, RFX_AS_RAW(o), dcol, (size_t)nout * 8) == RFX_OK;
                }
            }
            RFX_AS_LIST(rv)[c] = o;
        }
        if (!ok) { H.drop(rk_); H.drop(rv); res = fail_hip("join columns"); goto done; }
        res = H.table(rk_, rv);
        g_last_gpu = 1;
        goto done;
    }
out:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    ntmp = 0;
    if (H.bound == 1 && H.f[fidx]) res = HOST_CALL(((rfx_vary_f)H.f[fidx])(x, n));
    else {
        char b[320];
        snprintf(b, sizeof(b), "join: shape not covered by the MI355X path (%s) and no host function to delegate to", why ? why : "unsupported");
        res = fail(b);
    }
done:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    return res;
#undef JOIN_TMP
}
static obj_p join_op(int inner, obj_p *x, int64_t n) {
    op_begin();
    g_last_gpu = 0;
    obj_p r = join_impl(inner, x, n);
    g_stat[g_last_gpu ? ST_JOIN_GPU : ST_JOIN_DELEGATED]++;
    op_end();
    return r;
}
rfx_obj_p rfx_left_join(rfx_obj_p *x, int64_t n) { return join_op(0, x, n); }
rfx_obj_p rfx_inner_join(rfx_obj_p *x, int64_t n) { return join_op(1, x, n); }

static obj_p at_impl(obj_p col, obj_p ids) {
    rfx_host_bind();
    if (!col || !ids || !col_ctype(col) || ids->type != RFX_TYPE_I64) return fail("at: expected (i64|f64 column, I64 ids)");
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1) {
        /* over the shards: ids that ascend through the shards' row ranges (a filter's: ops_where) are cut at the boundaries, every shard gathers ITS
         * ids from its piece and its values go to the result at the ids' positions; ids in any other order (or null / out of range: they read as
         * nulls, at_vec_*_by_i64 core/items.c:53-72) need the column whole: the host's own `at` */
        const void *dcs;
        const int64_t *dsel[RFX_MAX_SHARDS];
        int64_t nsel[RFX_MAX_SHARDS];
        if (resident(col, 0, &dcs) != RFX_OK) return fail_hip("upload");
        int rc = ids->len ? sel_ids_sharded(ids, col->len, dsel, nsel) : RFX_OK;
        if (rc == 1) {
            qtmp_release();
            g_refused_sharded = 1;
            return (H.bound == 1 && g_host_at) ? HOST_CALL(((rfx_binary_f)g_host_at)(col, ids)) : fail_ctx();
        }
        obj_p outs = H.vector(col->type, ids->len);
        void *dg[RFX_MAX_SHARDS] = {0};
        int64_t at = 0;
        for (int sh = 0; sh < g_nshards && rc == RFX_OK && ids->len; sh++) {
            int64_t r0;
            rfx_exec_split(col->len, g_nshards, sh, &r0, NULL);
            if (nsel[sh]) {
                const void *piece = dcs;
                for (int k = 0; k < g_nqcols && sh > 0; k++)
                    if (g_qcols[k].d[0] == dcs) piece = g_qcols[k].d[sh];
                rfx_hip_ctx_bind_thread(g_ctxs[sh]);
                rc = rfx_hip_malloc(g_ctxs[sh], &dg[sh], (size_t)nsel[sh] * 8);
                if (rc == RFX_OK) rc = rfx_hip_gather(g_ctxs[sh], (const char *)piece - (size_t)r0 * 8, dsel[sh], nsel[sh], dg[sh]); /* (global ids: the piece's base moved back) */
                if (rc == RFX_OK) rc = rfx_hip_d2h_async(g_ctxs[sh], (char *)RFX_AS_RAW(outs) + (size_t)at * 8, dg[sh], (size_t)nsel[sh] * 8);
            }
            at += nsel[sh];
        }
        for (int sh = 0; sh < g_nshards; sh++) {
            if (!dg[sh]) continue;
            rfx_hip_ctx_bind_thread(g_ctxs[sh]);
            const int src = rfx_hip_ctx_sync(g_ctxs[sh]);
            if (rc == RFX_OK) rc = src;
            rfx_hip_free(g_ctxs[sh], dg[sh]);
        }
        rfx_hip_ctx_bind_thread(g_ctx);
        qtmp_release();
        if (rc != RFX_OK) { H.drop(outs); return fail_hip("gather"); }
        return outs;
    }
    const void *dc, *di;
    if (resident(col, 0, &dc) != RFX_OK || transient(ids, &di) != RFX_OK) return fail_hip("upload");
    obj_p out = H.vector(col->type, ids->len);
    void *dout = NULL;
    /* ids come from the caller: null / negative / out-of-range ids read as the typed null (at_vec_*_by_i64, core/items.c:53-72) */
    int ok = rfx_hip_malloc(g_ctx, &dout, (size_t)ids->len * 8 + 8) == RFX_OK &&
             rfx_hip_gather_checked(g_ctx, dc, col->len, col_ctype(col), (const int64_t *)di, ids->len, dout) == RFX_OK &&
             rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dout, (size_t)ids->len * 8) == RFX_OK;
    if (dout) rfx_hip_free(g_ctx, dout);
    if (!ok) { H.drop(out); return fail_hip("gather"); }
    return out;
}
rfx_obj_p rfx_at(rfx_obj_p col, rfx_obj_p ids) {
    op_begin();
    obj_p r = at_impl(col, ids);
    op_end();
    return r;
}
