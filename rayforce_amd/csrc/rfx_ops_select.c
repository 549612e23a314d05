/* rfx_ops_select.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * rfx_select: PLAN -> RUN (rfx_exec) -> BUILD (the host result table, one read-back of every column). */
/* ------------------------------------------------------------------------------------------------ select */
/* RFX_TRACE=2: where a select's wall time goes (microseconds between marks), one line per query on stderr */
static double g_tm[12];
static int g_ntm;
static void tm_mark(void) {
    if (g_ntm < 12) {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        g_tm[g_ntm++] = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    }
}
static void tm_print(void) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("RFX_TRACE"); on = e && atoi(e) >= 2; }
    if (on && g_ntm > 1) {
        fprintf(stderr, "[rfx] select us:");
        for (int i = 1; i < g_ntm; i++) fprintf(stderr, " %.0f", g_tm[i] - g_tm[i - 1]);
        fprintf(stderr, "  (plan | scope | tables+pass+rank | emit | fetch | build+free) total %.0f\n", g_tm[g_ntm - 1] - g_tm[0]);
    }
}
/* ---- rfx_select, piece by piece.  Every piece answers SEL_GO (carry on), SEL_OUT (*why says which shape the host must answer) or
 * SEL_DONE (an error / a finished result: the caller stops). ---- */
enum { SEL_GO = 0, SEL_OUT = 1, SEL_DONE = 2 };
/* the output mappings {name: (aggr column | expression)} of a select dict (everything but from: where: by: take:) as aggregate
 * descriptors over resident device columns */
#define SEL_AGGS (2 * RFX_MAX_AGGS + 1) /* the query's own + the hidden ones of the reproducible sums: a second limb per rewritten aggregate, one COUNT */
typedef struct {
    rfx_agg_t aggs[SEL_AGGS];
    rfx_xnode_t xnodes[RFX_MAX_AGGS][RFX_MAX_XNODES];
    int64_t names[RFX_MAX_AGGS];
    int outtype[RFX_MAX_AGGS];
    int nagg;
    /* reproducible grouped f64 sums (det_rewrite): aggregate a runs as an i64 SUM over its argument scaled by 2^det_k[a]; with two limbs a second, hidden
     * i64 SUM (aggregate det_lo[a], behind the query's own) adds up what the first limb's cells rounded away, scaled by 2^det_m[a] more; an average also
     * needs the groups' row counts: ONE hidden COUNT aggregate (det_cnt) */
    unsigned char det_on[RFX_MAX_AGGS], det_avg[RFX_MAX_AGGS];
    signed char det_lo[RFX_MAX_AGGS];
    int det_k[RFX_MAX_AGGS], det_m[RFX_MAX_AGGS];
    int nhidden, det_cnt;
} sel_maps_t;
/* result cells of an aggregate over a widened 4-byte column, back in the column's own width: the i64 null and the i64 identities of an
 * all-null group (core/aggr.c:1246) become the 4-byte ones */
static void sel_narrow_i32(obj_p col, const int64_t *cells, int64_t n, int kind) {
    int32_t *o = (int32_t *)RFX_AS_RAW(col);
    if (kind == RFX_AGG_SUM) { /* FOLD_ADDI32's result IS the low half of the 64-bit sum, whatever that sum is (no null / identity to translate) */
        for (int64_t i = 0; i < n; i++) o[i] = (int32_t)(uint32_t)(uint64_t)cells[i];
        return;
    }
    for (int64_t i = 0; i < n; i++) o[i] = cells[i] == RFX_NULL_I64 ? INT32_MIN : (cells[i] == INT64_MAX ? INT32_MAX : (int32_t)cells[i]);
}
static int sel_mappings(obj_p tab, obj_p dkeys, obj_p dvals, int grouped, sel_maps_t *M, const char **why) {
    const int64_t s_from = H.intern("from", 4), s_where = H.intern("where", 5), s_by = H.intern("by", 2), s_take = H.intern("take", 4);
    M->nagg = 0;
    M->nhidden = 0;
    M->det_cnt = -1;
    memset(M->det_lo, -1, sizeof(M->det_lo));
    memset(M->det_on, 0, sizeof(M->det_on));
    memset(M->det_avg, 0, sizeof(M->det_avg));
    for (int64_t i = 0; i < dkeys->len; i++) {
        int64_t k = RFX_AS_I64(dkeys)[i];
        if (k == s_from || k == s_where || k == s_by || k == s_take) continue;
        obj_p e = RFX_AS_LIST(dvals)[i];
        const int n = M->nagg;
        if (n >= RFX_MAX_AGGS || e->type != RFX_TYPE_LIST || e->len != 2) { *why = "mapping shape"; return SEL_OUT; }
        int f = fn_id(RFX_AS_LIST(e)[0]);
        obj_p a = RFX_AS_LIST(e)[1];
        static const int KIND[] = {RFX_AGG_SUM, RFX_AGG_AVG, RFX_AGG_MIN, RFX_AGG_MAX, RFX_AGG_COUNT, RFX_AGG_FIRST};
        if (f < F_SUM || f > F_FIRST) { *why = "mapping is not (aggr ...)"; return SEL_OUT; }
        memset(&M->aggs[n], 0, sizeof(M->aggs[n]));
        M->aggs[n].kind = KIND[f - F_SUM];
        if (a->type == RFX_TYPE_LIST && a->len == 3) {
            /* (aggr expr), expr = (op x y) over columns, atoms and nested expressions: folded on the device.  (count expr)
             * answers the number of groups in the reference and (first expr) under by: is a `length` error there: the host's */
            if (f == F_COUNT || f == F_FIRST) { *why = "count / first of an expression"; return SEL_OUT; }
            int nn = 0, ncols = 0;
            int top = build_xnodes(tab, a, M->xnodes[n], &nn, &ncols, why);
            if (top == -2) return SEL_DONE;
            if (top < 0) return SEL_OUT;
            if (ncols == 0) { *why = "expression without a column"; return SEL_OUT; }
            M->aggs[n].nxnodes = nn;
            M->aggs[n].xnodes = M->xnodes[n];
            M->aggs[n].col_type = RFX_I64;
            M->outtype[n] = (f == F_AVG || rfx_agg_input_type(&M->aggs[n]) == RFX_F64) ? RFX_TYPE_F64 : RFX_TYPE_I64;
            M->names[M->nagg++] = k;
            continue;
        }
        if (a->type != -RFX_TYPE_SYMBOL) { *why = "mapping is not (aggr column)"; return SEL_OUT; }
        obj_p c = table_col(tab, a->i64);
        /* a 4-byte integer column (I32 / DATE / TIME): min / max / first / count / sum fold its widened device copy and the result cells
         * are narrowed back (sel_narrow_i32); avg and the sum of dates are the host's */
        const int narrow = c && IS_I32_FAMILY(c->type) && !(g_npx && proxy_of(c)) && !(grouped && c->type == RFX_TYPE_I32) && /* (any grouped aggregate over an I32 column is a `type` error in the reference: its to say) */
                           ((f == F_MIN || f == F_MAX || f == F_FIRST || f == F_COUNT) ||
                            /* sums of I32 / TIME columns wrap in 32 bits there (FOLD_ADDI32 / ADDI32, core/math.c:1864-1871, core/aggr.c:1095-1100):
                             * the low 32 bits of the 64-bit sum of the widened column are that sum */
                            (f == F_SUM && !grouped && (c->type == RFX_TYPE_I32 || c->type == RFX_TYPE_TIME))); /* (grouped: a `type` error there) */
        if (!c || (!narrow && (!col_ctype(c) || c->type == RFX_TYPE_SYMBOL))) { *why = "aggregate column type"; return SEL_OUT; }
        const void *d;
        if (resident(c, 0, &d) != RFX_OK) return SEL_DONE;
        M->aggs[n].d_col = d;
        M->aggs[n].col_type = narrow ? RFX_I64 : col_ctype(c);
        M->outtype[n] = (f == F_AVG) ? RFX_TYPE_F64 : (f == F_COUNT) ? RFX_TYPE_I64 : c->type;
        M->names[M->nagg++] = k;
    }
    return SEL_GO;
}

/* the key column(s) of a group-by result, read back in group order from what the planner emitted: one key as its cells (the virtual Date
 * column of a parted table narrowed to 4-byte days, ENUM indices decoded through the enum's domain -- aggr_first, core/aggr.c:515-546);
 * several keys as the planner's key columns (decoded from the composite key = key_i[first row], core/query.c:110-135, or gathered at the
 * groups' first rows on the row-hash path).  *ok carries the device-call status on; SEL_OUT: an enum whose domain does not resolve. */
typedef struct {
    int nkeys;
    int8_t key_out_type;
    obj_p kenum;
    obj_p *kcs;
} sel_keys_t;
/* (two steps around ONE read-back of every result column -- rfx_exec_groups_fetch_all, each slice over its own device's link:
 * sel_key_columns_plan makes the vectors and names (device column, host destination) pairs, sel_key_columns_finish narrows / decodes) */
typedef struct {
    int n;
    const void *src[RFX_MAX_KEYS + SEL_AGGS];
    void *dst[RFX_MAX_KEYS + SEL_AGGS];
    void *tmp[RFX_MAX_KEYS + SEL_AGGS]; /* 8-byte staging of a column whose vector is 4 bytes wide (freed by the caller) */
    int ntmp;
} sel_fetch_t;
static int sel_fetch_add(sel_fetch_t *F, const void *src, void *dst) {
    if (F->n >= (int)(sizeof(F->src) / sizeof(F->src[0]))) return 0;
    F->src[F->n] = src;
    F->dst[F->n++] = dst;
    return 1;
}
static void *sel_fetch_tmp(sel_fetch_t *F, int64_t groups) {
    void *t = malloc((size_t)(groups ? groups : 1) * 8);
    if (t) F->tmp[F->ntmp++] = t;
    return t;
}
static int sel_key_columns_plan(const sel_keys_t *K, const rfx_groups_t *R, obj_p *okcols, sel_fetch_t *F, int64_t **k8) {
    const int64_t groups = R->groups;
    int ok = 1;
    *k8 = NULL;
    if (K->nkeys == 1 && K->key_out_type == RFX_TYPE_DATE) { /* the virtual Date column: 4-byte days */
        okcols[0] = H.vector(RFX_TYPE_DATE, groups);
        *k8 = (int64_t *)sel_fetch_tmp(F, groups);
        ok = *k8 && sel_fetch_add(F, R->d_keys, *k8);
    } else if (K->nkeys == 1) {
        okcols[0] = H.vector(K->key_out_type, groups);
        ok = sel_fetch_add(F, R->d_keys, RFX_AS_RAW(okcols[0]));
    } else {
        for (int i = 0; i < K->nkeys && ok; i++) {
            okcols[i] = H.vector(K->kcs[i]->type, groups);
            ok = sel_fetch_add(F, R->d_keycols[i], RFX_AS_RAW(okcols[i]));
        }
    }
    return ok;
}
static int sel_key_columns_finish(const sel_keys_t *K, const rfx_groups_t *R, obj_p *okcols, const int64_t *k8) {
    const int64_t groups = R->groups;
    if (K->nkeys == 1 && K->key_out_type == RFX_TYPE_DATE) {
        for (int64_t g = 0; g < groups; g++) ((int32_t *)RFX_AS_RAW(okcols[0]))[g] = (int32_t)k8[g];
    } else if (K->nkeys == 1 && K->kenum) { /* indices -> symbols of the enum's domain (the global its key names) */
        obj_p dom = enum_domain(K->kenum);
        int good = dom != NULL;
        int64_t *kk = RFX_AS_I64(okcols[0]);
        for (int64_t g = 0; g < groups && good; g++) {
            if (kk[g] < 0 || kk[g] >= dom->len) good = 0;
            else kk[g] = RFX_AS_I64(dom)[kk[g]];
        }
        if (dom) H.drop(dom);
        if (!good) {
            H.drop(okcols[0]);
            okcols[0] = NULL;
            return SEL_OUT;
        }
    }
    return SEL_GO;
}

/* by: a column symbol, or a dict {name: column | (xbar column positive-width) ...} (get_gkeys / get_gvals, core/query.c:165-240): the key
 * columns as the table holds them, the names they take in the result, and the bucket width of the bucketed ones */
static int sel_by_shape(obj_p tab, obj_p by, obj_p *kcs, int64_t *knames, int64_t *kxbar, int *nkeys, const char **why) {
    *nkeys = 0;
    if (by->type == -RFX_TYPE_SYMBOL) {
        knames[0] = by->i64;
        kxbar[0] = 0;
        kcs[(*nkeys)++] = table_col(tab, by->i64);
        return SEL_GO;
    }
    if (!(by->type == RFX_TYPE_DICT && RFX_AS_LIST(by)[0]->type == RFX_TYPE_SYMBOL)) { *why = "by: is neither a column nor a dict of columns"; return SEL_OUT; }
    obj_p bk = RFX_AS_LIST(by)[0], bv = RFX_AS_LIST(by)[1];
    if (bk->len < 1 || bk->len > RFX_MAX_KEYS || bv->len != bk->len) { *why = "by: dict shape"; return SEL_OUT; }
    for (int64_t i = 0; i < bk->len; i++) {
        int64_t sym;
        obj_p bx = (bv->type == RFX_TYPE_LIST) ? RFX_AS_LIST(bv)[i] : NULL;
        kxbar[*nkeys] = 0;
        if (bv->type == RFX_TYPE_SYMBOL) sym = RFX_AS_I64(bv)[i];
        else if (bx && bx->type == -RFX_TYPE_SYMBOL) sym = bx->i64;
        else if (bx && bx->type == RFX_TYPE_LIST && bx->len == 3 && fn_id(RFX_AS_LIST(bx)[0]) == F_XBAR && RFX_AS_LIST(bx)[1]->type == -RFX_TYPE_SYMBOL &&
                 RFX_AS_LIST(bx)[2]->type == -RFX_TYPE_I64 && RFX_AS_LIST(bx)[2]->i64 > 0) {
            sym = RFX_AS_LIST(bx)[1]->i64; /* (xbar column width): bucketed key, evaluated on the device by the caller */
            kxbar[*nkeys] = RFX_AS_LIST(bx)[2]->i64;
        } else { *why = "by: key is an expression other than (xbar column positive-width)"; return SEL_OUT; }
        knames[*nkeys] = RFX_AS_I64(bk)[i];
        kcs[(*nkeys)++] = table_col(tab, sym);
    }
    return SEL_GO;
}

/* select without aggregates: filter_collect of every column (core/filter.c:51-165) -- where -> ids (every shard its own, rfx_exec_where) ->
 * every column gathered at them where its rows live, straight into the result vectors */
static int sel_projection(obj_p tab, const rfx_query_t *Q, int parted, obj_p *res, const char **why) {
    obj_p tcols = RFX_AS_LIST(tab)[1];
    if (parted) { *why = "parted table: projection"; return SEL_OUT; } /* the reference keeps such a result lazy (filter maps over the partitions) */
    if (!Q->npred && !Q->d_mask) { *res = H.clone(tab); g_last_gpu = 1; return SEL_DONE; }
    for (int64_t i = 0; i < tcols->len; i++)
        if (!col_ctype(RFX_AS_LIST(tcols)[i])) { *why = "projection of a non-8-byte column"; return SEL_OUT; }
    rfx_ids_t ids;
    if (rfx_exec_where(g_x, Q, &ids) != RFX_OK) { *res = fail(rfx_exec_last_error(g_x)); return SEL_DONE; }
    const int64_t nsel = ids.total;
    obj_p rv = H.vector(RFX_TYPE_LIST, tcols->len);
    int ok = 1;
    for (int64_t i = 0; i < tcols->len && ok; i++) {
        obj_p c = RFX_AS_LIST(tcols)[i];
        obj_p o = H.vector(c->type, nsel);
        RFX_AS_LIST(rv)[i] = o;
        const void *dc;
        if (nsel == 0) continue;
        ok = resident(c, 0, &dc) == RFX_OK;
        int64_t at = 0;
        for (int sh = 0; sh < ids.nshards && ok; sh++) {
            if (!ids.count[sh]) continue;
            int64_t r0;
            rfx_exec_split(Q->nrows, ids.nshards, sh, &r0, NULL);
            const void *dcs = dc; /* this shard's slice, addressed by the GLOBAL ids it emitted */
            for (int k = 0; k < g_nqcols && sh > 0; k++)
                if (g_qcols[k].d[0] == dc) dcs = g_qcols[k].d[sh];
            void *dg = NULL;
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[sh]);
            ok = rfx_hip_malloc(g_ctxs[sh], &dg, (size_t)ids.count[sh] * 8) == RFX_OK &&
                 rfx_hip_gather(g_ctxs[sh], (const char *)dcs - (size_t)r0 * 8, ids.d_ids[sh], ids.count[sh], dg) == RFX_OK &&
                 rfx_hip_d2h(g_ctxs[sh], (char *)RFX_AS_RAW(o) + (size_t)at * 8, dg, (size_t)ids.count[sh] * 8) == RFX_OK;
            if (dg) rfx_hip_free(g_ctxs[sh], dg);
            at += ids.count[sh];
        }
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    }
    rfx_exec_ids_free(g_x, &ids);
    if (!ok) { H.drop(rv); *res = fail_hip("projection"); return SEL_DONE; }
    *res = H.table(H.clone(RFX_AS_LIST(tab)[0]), rv);
    g_last_gpu = 1;
    return SEL_DONE;
}

/* rfx_select = PLAN (the dict's clauses as descriptors over resident columns: sel_mappings, plan_where, sel_by_shape -- and what the
 * reference answers differently is handed back before anything runs) -> RUN (the planner: rfx_exec_group_by / rfx_exec_filter_aggr /
 * rfx_exec_where over the operator layer's shards) -> BUILD (the result table from the planner's device columns). */
/* ---- reproducible grouped f64 sums (round 6; opt-in: RFX_DETERMINISTIC=1 / rfx_ops_set_deterministic(1)) ----
 * The reference's grouped sums are bit-stable for a fixed pool size (its workers' partials merge in task order, core/pool.c:415-424); here rows reach a
 * group's f64 accumulator through atomics in the order the waves happen to run.  In this mode every `(sum x)` / `(avg x)` over f64 under by: runs as an
 * INTEGER sum -- associative, so any order gives the same bits -- over x scaled by a power of two and rounded to i64 once per cell:
 *   M = max |x| over the column (rfx_hip_absmax_f64; a NaN / infinity anywhere leaves the aggregate as it was), 2^e > M, 2^b >= rows,
 *   k = 62 - e - b  (no sum of <= rows cells can leave 63 bits),   fixed = llrint(x * 2^k)  (rfx_hip_fix_f64, every shard its rows),
 *   result = (double)sum(fixed) * 2^-k   (/ the group's row count for avg: one hidden COUNT aggregate).
 * What it costs: two more passes over the argument (8 B read; 8 B read + 8 B written per row) and an i64 sum (no f64 fast paths).  What it gives up: a
 * cell is rounded to a multiple of 2^-k = 2^(e + b - 62): for 1e9 rows of values up to 1.0 that is 2^-32 -- 1.2e-10 absolute per cell; data whose typical
 * magnitude is far below its maximum loses relative precision accordingly (a group's sum is off by at most rows_in_group * 2^-(k + 1)). */
static int g_det = -1;
static int det_mode(void) {
    if (g_det < 0) {
        const char *e = getenv("RFX_DETERMINISTIC");
        g_det = e ? atoi(e) : 0;
        if (g_det < 0 || g_det > 2) g_det = 1;
    }
    return g_det;
}
int rfx_ops_set_deterministic(int mode) {
    g_det = mode <= 0 ? 0 : (mode >= 2 ? 2 : 1);
    return RFX_OK;
}
/* One process per device, every rank a row range of the table (rfx_ops_dist_init / a transport): by default every rank's rfx_select returns the WHOLE answer
 * (a replicated evaluator); with rank slices on (RFX_RANK_SLICES=1 / rfx_ops_set_rank_slices) a grouped select returns only the rank's range of the groups --
 * rfx_exec_split(groups, ranks, rank), in the answer's own order, so the ranks' tables laid end to end in rank order ARE the answer -- and reads only that
 * range back over its PCIe link (the whole 16 MB result of the metric's query on every rank is a constant 0.31 ms that does not shrink with the ranks). */
static int g_rank_slices = -1;
static int rank_slices_mode(void) {
    if (g_rank_slices < 0) {
        const char *e = getenv("RFX_RANK_SLICES");
        g_rank_slices = e && atoi(e) != 0;
    }
    return g_rank_slices;
}
int rfx_ops_set_rank_slices(int on) {
    g_rank_slices = on ? 1 : 0;
    return RFX_OK;
}
static const void *shard_piece(const void *p, int s);
/* 0: done (aggregates rewritten where possible), -2: device failure */
static int det_rewrite(sel_maps_t *M, int64_t nrows) {
    if (nrows <= 0 && det_ranks() == 1) return 0; /* (a rank without rows still takes part in the ranks' agreement on the scale, and folds the same aggregates) */
    const int limbs = det_mode() >= 2 ? 2 : 1;
    int need_count = 0, count_type = RFX_I64;
    const void *count_col = NULL;
    for (int a = 0; a < M->nagg; a++) {
        rfx_agg_t *ag = &M->aggs[a];
        if (!(ag->kind == RFX_AGG_SUM || ag->kind == RFX_AGG_AVG) || rfx_agg_input_type(ag) != RFX_F64) continue;
        if (g_nqtmp + limbs > (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) continue;
        const int is_expr = ag->nxnodes > 0 || ag->xop != RFX_X_NONE, is_avg = ag->kind == RFX_AGG_AVG;
        int k = 0, m = 0, done = 0;
        const void *img[2] = {NULL, NULL};
        if (!is_expr) { /* a plain column the cache holds by ownership: its fixed-point image is made ONCE and kept with it (resident_fixed) */
            const int frc = resident_fixed(ag->d_col, nrows, limbs, &k, &m, img);
            if (frc == 2) continue; /* (a NaN / an infinity in the column: the default path and its poisoning rules) */
            if (frc != RFX_OK && frc != 1) return -2;
            if (frc == RFX_OK) {
                if (!count_col) count_col = ag->d_col, count_type = ag->col_type;
                done = 1;
            }
        }
        if (!done) {
            /* otherwise (an expression; a device vector; checksum mode) the argument as ONE f64 scratch column of this query, every shard its rows: an
             * expression evaluated, a plain column read where it lies -- and the image(s) made from it */
            void *devs[2][RFX_MAX_SHARDS];
            for (int l = 0; l < limbs; l++) {
                if (shards_alloc(devs[l], nrows, 8, 0) != RFX_OK) return -2;
                memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
                for (int s = 0; s < g_nshards; s++) g_qtmp[g_nqtmp].d[s] = devs[l][s];
                g_nqtmp++; /* (released with the query's scratch) */
            }
            int rc = RFX_OK, bad = 0, f64_out = 1;
            double mx = 0.0;
            for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
                int64_t n;
                rfx_exec_split(nrows, g_nshards, s, NULL, &n);
                if (g_nshards == 1) n = nrows;
                if (n <= 0) continue;
                if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
                const void *src = shard_piece(ag->d_col, s);
                if (is_expr) {
                    rfx_agg_t as = *ag;
                    rfx_xnode_t xn[RFX_MAX_XNODES];
                    as.d_col = shard_piece(ag->d_col, s);
                    as.d_xrhs_col = shard_piece(ag->d_xrhs_col, s);
                    for (int j = 0; j < ag->nxnodes && j < RFX_MAX_XNODES; j++) {
                        xn[j] = ag->xnodes[j];
                        if (xn[j].l.kind == RFX_XK_COL) xn[j].l.d_col = shard_piece(ag->xnodes[j].l.d_col, s);
                        if (xn[j].r.kind == RFX_XK_COL) xn[j].r.d_col = shard_piece(ag->xnodes[j].r.d_col, s);
                    }
                    if (ag->nxnodes > 0) as.xnodes = xn;
                    int32_t ot = RFX_I64;
                    rc = rfx_hip_eval_expr(g_ctxs[s], &as, n, devs[0][s], &ot);
                    if (ot != RFX_F64) f64_out = 0;
                    src = devs[0][s];
                }
                double m1 = 0.0;
                int b1 = 0;
                if (rc == RFX_OK && f64_out) rc = rfx_hip_absmax_f64(g_ctxs[s], (const double *)src, n, &m1, &b1);
                mx = m1 > mx ? m1 : mx;
                bad |= b1;
            }
            int64_t world_rows = nrows;
            bad |= !f64_out;
            if (rc == RFX_OK) rc = det_world_agree(&mx, &world_rows, &bad);
            if (rc == RFX_OK && !bad) {
                k = det_scale(mx, world_rows, &m);
                if (k > -1000 && k < 1000) {
                    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
                        int64_t n;
                        rfx_exec_split(nrows, g_nshards, s, NULL, &n);
                        if (g_nshards == 1) n = nrows;
                        if (n <= 0) continue;
                        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
                        const double *src = (const double *)(is_expr ? devs[0][s] : shard_piece(ag->d_col, s));
                        if (limbs == 2) rc = rfx_hip_fix_f64_low(g_ctxs[s], src, n, k, m, (int64_t *)devs[1][s]); /* (first: the other limb goes in place) */
                        if (rc == RFX_OK) rc = rfx_hip_fix_f64(g_ctxs[s], src, n, k, (int64_t *)devs[0][s]);
                    }
                    for (int l = 0; l < limbs && rc == RFX_OK; l++) {
                        rc = qcol_add(devs[l]);
                        img[l] = devs[l][0];
                    }
                    done = rc == RFX_OK;
                }
            }
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
            if (rc != RFX_OK) return -2;
        }
        if (!done) continue;
        M->det_on[a] = 1;
        M->det_avg[a] = (unsigned char)is_avg;
        M->det_k[a] = k;
        M->det_m[a] = m;
        need_count |= is_avg;
        memset(ag, 0, sizeof(*ag));
        ag->d_col = img[0];
        ag->col_type = RFX_I64;
        ag->kind = RFX_AGG_SUM;
        if (limbs == 2) { /* the second limb: a hidden i64 SUM behind the query's own aggregates */
            const int idx = M->nagg + M->nhidden++;
            rfx_agg_t *lo = &M->aggs[idx];
            memset(lo, 0, sizeof(*lo));
            lo->d_col = img[1];
            lo->col_type = RFX_I64;
            lo->kind = RFX_AGG_SUM;
            M->det_lo[a] = (signed char)idx;
        }
    }
    if (need_count) { /* the groups' row counts, once, behind everything else */
        const int idx = M->nagg + M->nhidden++;
        rfx_agg_t *c = &M->aggs[idx];
        memset(c, 0, sizeof(*c));
        for (int a = 0; a < M->nagg && !count_col; a++)
            if (M->det_on[a]) count_col = M->aggs[a].d_col; /* (a scratch image: i64) */
        c->d_col = count_col;
        c->col_type = count_type;
        c->kind = RFX_AGG_COUNT;
        M->det_cnt = idx;
    }
    return 0;
}

static obj_p sel_build_groups(const rfx_groups_t *R, const sel_maps_t *M, const sel_keys_t *K, const int64_t *knames, const char **why) {
    const int nagg = M->nagg, nkeys = K->nkeys;
    obj_p ocols[RFX_MAX_AGGS] = {0}, okcols[RFX_MAX_KEYS] = {0};
    int ok = 1, enum_out = 0;
    int64_t *hid[SEL_AGGS] = {0}; /* the hidden aggregates' result columns (second limbs, the row counts) */
    if (R->groups > 0) {
        /* every result vector first, then ONE read-back of all of them (every slice of a sliced result by the shard that holds it, over that
         * device's own link: the table construction of core/query.c:559-605 with N writers), then the 4-byte narrowing / enum decoding */
        sel_fetch_t F;
        int64_t *k8 = NULL, *c8[RFX_MAX_AGGS] = {0};
        memset(&F, 0, sizeof(F));
        ok = sel_key_columns_plan(K, R, okcols, &F, &k8);
        for (int a = 0; a < nagg && ok; a++) {
            ocols[a] = H.vector((int8_t)M->outtype[a], R->groups);
            if (IS_I32_FAMILY(M->outtype[a])) {
                c8[a] = (int64_t *)sel_fetch_tmp(&F, R->groups);
                ok = c8[a] && sel_fetch_add(&F, R->d_results[a], c8[a]);
            } else ok = sel_fetch_add(&F, R->d_results[a], RFX_AS_RAW(ocols[a]));
        }
        /* reproducible sums: the integer sums back as f64 (an i64 sum's null cannot occur: no null went in, no sum leaves 63 bits) -- ON THE DEVICE, in place,
         * every slice where it lies (the hidden columns then never cross the link); a small dense result is mirrored on the host already: there below */
        const int on_device = !R->h_block;
        for (int a = 0; a < nagg && ok && on_device; a++) {
            if (!M->det_on[a]) continue;
            const double sc = ldexp(1.0, -M->det_k[a]), sc2 = ldexp(1.0, -(M->det_k[a] + M->det_m[a]));
            const int lo = M->det_lo[a], cn = M->det_avg[a] ? M->det_cnt : -1;
            if (R->nslices <= 1) ok = rfx_hip_unfix_f64(g_ctxs[R->nslices == 1 ? R->slice[0].shard : 0], (int64_t *)R->d_results[a], lo >= 0 ? (const int64_t *)R->d_results[lo] : NULL,
                                                        cn >= 0 ? (const int64_t *)R->d_results[cn] : NULL, R->groups, sc, sc2) == RFX_OK;
            else {
                for (int i = 0; i < R->nslices && ok; i++) {
                    const struct rfx_gslice *sl = &R->slice[i];
                    rfx_hip_ctx_bind_thread(g_ctxs[sl->shard]);
                    ok = rfx_hip_unfix_f64(g_ctxs[sl->shard], (int64_t *)sl->d_results[a], lo >= 0 ? (const int64_t *)sl->d_results[lo] : NULL,
                                           cn >= 0 ? (const int64_t *)sl->d_results[cn] : NULL, sl->n, sc, sc2) == RFX_OK;
                }
                rfx_hip_ctx_bind_thread(g_ctx);
            }
        }
        for (int j = nagg; j < nagg + M->nhidden && ok && !on_device; j++) {
            hid[j] = (int64_t *)sel_fetch_tmp(&F, R->groups);
            ok = hid[j] && sel_fetch_add(&F, R->d_results[j], hid[j]);
        }
        if (ok) ok = rfx_exec_groups_fetch_all(g_x, R, F.n, F.src, F.dst) == RFX_OK;
        for (int a = 0; a < nagg && ok && !on_device; a++) {
            if (!M->det_on[a]) continue;
            int64_t *raw = (int64_t *)RFX_AS_RAW(ocols[a]);
            double *out = (double *)RFX_AS_RAW(ocols[a]);
            /* (powers of two: the products below are exact -- ldexp per cell cost 2.5 ms per 1e6 groups; the second limb's scale may underflow to 0 for
             * columns of tiny values: what it carries is below the subnormals then) */
            const double sc = ldexp(1.0, -M->det_k[a]), sc2 = ldexp(1.0, -(M->det_k[a] + M->det_m[a]));
            const int64_t *lo = M->det_lo[a] >= 0 ? hid[(int)M->det_lo[a]] : NULL, *cnt = M->det_cnt >= 0 ? hid[M->det_cnt] : NULL;
            const int64_t ng = R->groups;
            if (!lo && !M->det_avg[a]) for (int64_t g = 0; g < ng; g++) out[g] = (double)raw[g] * sc;
            else if (!lo) for (int64_t g = 0; g < ng; g++) out[g] = cnt[g] ? ((double)raw[g] * sc) / (double)cnt[g] : NAN;
            else if (!M->det_avg[a]) for (int64_t g = 0; g < ng; g++) out[g] = (double)raw[g] * sc + (double)lo[g] * sc2;
            else for (int64_t g = 0; g < ng; g++) out[g] = cnt[g] ? ((double)raw[g] * sc + (double)lo[g] * sc2) / (double)cnt[g] : NAN;
        }
        if (ok) {
            enum_out = sel_key_columns_finish(K, R, okcols, k8) == SEL_OUT;
            for (int a = 0; a < nagg && !enum_out; a++)
                if (c8[a]) sel_narrow_i32(ocols[a], c8[a], R->groups, M->aggs[a].kind);
        }
        for (int i = 0; i < F.ntmp; i++) free(F.tmp[i]);
    }
    if (!ok || enum_out) {
        for (int i = 0; i < nkeys; i++) if (okcols[i]) H.drop(okcols[i]);
        for (int a = 0; a < nagg; a++) if (ocols[a]) H.drop(ocols[a]);
        if (enum_out) {
            *why = "by: enum column whose domain cannot be resolved";
            return NULL;
        }
        return fail_hip("group-by result");
    }
    obj_p rk = H.vector(RFX_TYPE_SYMBOL, nagg + nkeys), rv = H.vector(RFX_TYPE_LIST, nagg + nkeys);
    for (int i = 0; i < nkeys; i++) {
        RFX_AS_I64(rk)[i] = knames[i];
        RFX_AS_LIST(rv)[i] = okcols[i] ? okcols[i] : H.vector(nkeys == 1 ? K->key_out_type : K->kcs[i]->type, 0);
    }
    for (int a = 0; a < nagg; a++) {
        RFX_AS_I64(rk)[a + nkeys] = M->names[a];
        RFX_AS_LIST(rv)[a + nkeys] = ocols[a] ? ocols[a] : H.vector((int8_t)M->outtype[a], 0);
    }
    return H.table(rk, rv);
}
static obj_p sel_build_scalar(const rfx_value_t *vals, const sel_maps_t *M) {
    const int nagg = M->nagg;
    obj_p rk = H.vector(RFX_TYPE_SYMBOL, nagg), rv = H.vector(RFX_TYPE_LIST, nagg);
    for (int a = 0; a < nagg; a++) {
        RFX_AS_I64(rk)[a] = M->names[a];
        if (IS_I32_FAMILY(M->outtype[a])) {
            RFX_AS_LIST(rv)[a] = H.vector((int8_t)M->outtype[a], 1);
            sel_narrow_i32(RFX_AS_LIST(rv)[a], &vals[a].i, 1, M->aggs[a].kind);
        } else {
            RFX_AS_LIST(rv)[a] = one_row(&vals[a]);
            if (M->outtype[a] == RFX_TYPE_TIMESTAMP && vals[a].type != RFX_F64) RFX_AS_LIST(rv)[a]->type = RFX_TYPE_TIMESTAMP; /* min / max / first of a TIMESTAMP column */
        }
    }
    return H.table(rk, rv);
}

static obj_p select_impl(obj_p dict) {
    rfx_host_bind();
    g_ntm = 0;
    tm_mark();
    if (!dict || dict->type != RFX_TYPE_DICT || RFX_AS_LIST(dict)[0]->type != RFX_TYPE_SYMBOL) return fail("select: expected a dict");
    obj_p from = dict_get(dict, "from");
    if (!from) return fail("'select' expects 'from' param"); /* core/query.c:281 */
    /* take: is applied to the finished result table (ray_take(res, take), core/query.c:294-303,596-599): by the host's own ray_take */
    obj_p take = dict_get(dict, "take");
    if (take && !(H.bound == 1 && H.f[F_TAKE])) return delegate_select(dict, "take: without the host's ray_take");
    obj_p host_tab = HOST_CALL(H.eval(from)); /* (the host may fan this out to pool workers that call rfx_* built-ins: not under our lock; no device state is held yet) */
    if (!host_tab || host_tab->type == RFX_TYPE_ERR) return host_tab;
    obj_p tab = host_tab; /* the table the plan reads: host_tab itself, or the view of a parted table */
    int parted = 0;
    obj_p res = NULL;
    const char *why = NULL;
    void *tmp[2 * RFX_MAX_KEYS + 6]; /* device scratch of this query on shard 0 (a mask): freed at `done` */
    int ntmp = 0;
    obj_p where = dict_get(dict, "where"), by = dict_get(dict, "by");
    obj_p dkeys = RFX_AS_LIST(dict)[0], dvals = RFX_AS_LIST(dict)[1];
    if (tab->type != RFX_TYPE_TABLE) { why = "from: is not a table"; goto out; }
    if (ensure_ctx() != RFX_OK) { res = fail_hip("no usable MI355X"); goto done; }
    if (is_parted_table(host_tab)) {
        if (g_nshards > 1) { why = "parted table: the sharded operator layer takes in-memory tables"; goto out; }
        tab = parted_view(host_tab);
        if (!tab) { parted_view_release(); tab = host_tab; why = "parted table: view"; goto out; }
        parted = 1;
    }
    {
        /* ---------------------------------------------------------------- PLAN */
        obj_p tcols = RFX_AS_LIST(tab)[1];
        const int64_t nrows = tcols->len ? RFX_AS_LIST(tcols)[0]->len : 0;
        wplan_t wp;
        int flat = 1;
        g_where_virtual = g_where_data = 0;
        int rc = plan_where(tab, where, &wp);
        if (rc == -2) { res = fail_hip("column upload"); goto done; }
        if (rc) { /* more comparisons / levels than the fused form carries: its selection comes as a mask */
            flat = 0;
            wp.npred = 0;
            wp.logic = RFX_AND;
        }
        if (parted) {
            /* What the reference answers correctly over a parted table, and so what is answered here: aggregates, over everything or
             * grouped by the virtual column, filtered by the virtual column (partition pruning) or -- ungrouped -- by data columns.
             * A filter mixing both kinds, and a data-column filter under by:, come out wrong there (DESIGN.md "reference defects"):
             * left to the host so that this entry point never answers differently. */
            if (!flat) { why = "parted table: where: is not a flat and / or of comparisons"; goto out; }
            for (int i = 0; i < wp.npred; i++)
                if (wp.preds[i].more) { why = "parted table: where: is not a flat and / or of comparisons"; goto out; }
            if (g_where_virtual && g_where_data) { why = "parted table: where: mixes the virtual column with data columns"; goto out; }
            if (by && g_where_data) { why = "parted table: by: under a data-column filter"; goto out; }
        }
        sel_maps_t M;
        {
            const int mrc = sel_mappings(tab, dkeys, dvals, by != NULL, &M, &why);
            if (mrc == SEL_DONE) { res = fail_hip("column upload"); goto done; }
            if (mrc == SEL_OUT) goto out;
        }
        /* by: a column symbol, or a dict {name: column ...} (get_gkeys / get_gvals, core/query.c:165-240) */
        obj_p kcs[RFX_MAX_KEYS] = {0};
        const void *dks[RFX_MAX_KEYS] = {0};
        int64_t knames[RFX_MAX_KEYS], kxbar[RFX_MAX_KEYS] = {0};
        int nkeys = 0;
        int8_t key_out_type = RFX_TYPE_I64; /* one key: type of the result's key column */
        obj_p kenum = NULL;                 /* one key, an ENUM column */
        if (by) {
            if (sel_by_shape(tab, by, kcs, knames, kxbar, &nkeys, &why) == SEL_OUT) goto out;
            for (int i = 0; i < nkeys; i++) {
                if (parted) { /* only the virtual column groups a parted table in the reference (INDEX_TYPE_PARTEDCOMMON, core/index.c:2199-2222) */
                    const proxy_t *px = kcs[i] ? proxy_of(kcs[i]) : NULL;
                    if (nkeys != 1 || !px || px->kind != 2 || kcs[i]->type != RFX_TYPE_I64) { why = "parted table: by: is not the virtual column"; goto out; }
                    key_out_type = px->vtype;
                } else if (kcs[i] && kcs[i]->type == RFX_TYPE_ENUM && nkeys == 1 && !kxbar[i]) {
                    /* an enumerated symbol column groups on its indices (index_group_i64(ENUM_VAL(val)), core/index.c:2190-2191); the
                     * result's key column is decoded through the enum's domain (aggr_first, core/aggr.c:515-546) */
                    kenum = kcs[i];
                    if (resident(enum_indices(kcs[i]), 0, &dks[i]) != RFX_OK) { res = fail_hip("column upload"); goto done; }
                    key_out_type = RFX_TYPE_SYMBOL;
                    continue;
                }
                /* index_group's 8-byte integer arms: I64 / SYMBOL / TIMESTAMP group on the raw i64 (core/index.c:2183-2186); an F64 key
                 * column groups on its BIT PATTERN through the open-addressing path (index_group_f64 = index_group_i64_unscoped,
                 * core/index.c:2108,1959-1977): the same device column read as i64 -- a range of bit patterns is never dense, so the
                 * hashed tables take it here too; -0.0 has the bits of NULL_I64, the reference's empty-slot marker: handed back like
                 * any null key.  One key column only (several keys with an f64 among them are the host's). */
                const int f64key = kcs[i] && kcs[i]->type == RFX_TYPE_F64 && nkeys == 1 && !kxbar[i] && !parted;
                if (!kcs[i] || !(kcs[i]->type == RFX_TYPE_I64 || kcs[i]->type == RFX_TYPE_SYMBOL || kcs[i]->type == RFX_TYPE_TIMESTAMP || f64key)) {
                    why = "by: key is not an 8-byte integer column";
                    goto out;
                }
                if (kxbar[i] > 0 && kcs[i]->type == RFX_TYPE_SYMBOL) { why = "xbar over a symbol column"; goto out; }
                if (nkeys == 1 && !parted) key_out_type = kcs[i]->type;
                if (resident(kcs[i], 0, &dks[i]) != RFX_OK) { res = fail_hip("column upload"); goto done; }
            }
            /* where: + several keys: the reference's own result is defective (its composite index drops the filter, so key
             * columns and aggregates are taken from the wrong rows -- DESIGN.md "reference defects"); leave that to the host
             * so that this entry point never answers differently from ray_select. */
            if (nkeys > 1 && where) { why = "where: with several by: columns"; goto out; }
        }
        rfx_query_t Q;
        memset(&Q, 0, sizeof(Q));
        Q.preds = wp.preds;
        Q.npred = wp.npred;
        Q.logic = wp.logic;
        Q.aggs = M.aggs;
        Q.nagg = M.nagg;
        Q.nkeys = nkeys;
        Q.d_keys = dks;
        Q.kxbar = kxbar;
        Q.nrows = nrows;
        if (!flat && g_nshards > 1) { /* over the shards: the tree as a 0 / 1 column per shard, read by ONE comparison of the fused pass */
            const void *mcol = NULL;
            const int mrc = mask_column_sharded(tab, where, nrows, &mcol);
            if (mrc == -1) { why = "where: shape"; goto out; }
            if (mrc) { res = fail_hip("where"); goto done; }
            memset(&wp.preds[0], 0, sizeof(wp.preds[0]));
            wp.preds[0].d_col = mcol;
            wp.preds[0].col_type = RFX_I64;
            wp.preds[0].op = RFX_NE;
            wp.preds[0].rhs_type = RFX_I64;
            wp.preds[0].rhs_i = 0;
            wp.npred = 1;
            wp.logic = RFX_AND;
            Q.preds = wp.preds;
            Q.npred = 1;
            Q.logic = RFX_AND;
        } else if (!flat) { /* the tree as ONE B8 mask on the device (core/cmp.c -> K2, core/logic.c in place), handed to the planner beside the query */
            int8_t *m = NULL;
            const int mrc = mask_of_expr(tab, where, nrows, &m);
            if (mrc == -1) { why = "where: shape"; goto out; }
            if (mrc) { res = fail_hip("where"); goto done; }
            tmp[ntmp++] = m;
            Q.d_mask = m;
        }
        if (by && !parted && det_mode()) { /* reproducible grouped f64 sums: the aggregates' arguments as scaled i64 columns (opt-in) */
            if (det_rewrite(&M, nrows) != 0) { res = fail_hip("deterministic sums"); goto done; }
            Q.nagg = M.nagg + M.nhidden;
        }
        Q.cols = g_nshards > 1 ? g_qcols : NULL;
        Q.ncols = g_nqcols;
        tm_mark();
        /* ---------------------------------------------------------------- RUN + BUILD */
        if (!by && M.nagg == 0) { /* projection */
            if (sel_projection(tab, &Q, parted, &res, &why) == SEL_OUT) goto out;
            goto done;
        }
        if (by) {
            /* small inputs over a plain resident key column: its whole-column scope, remembered with the device copy (or taken now, without
             * the filter: the same pass) -- a superset of any selection's, which is all the tables' sizing needs; the planner takes it when it
             * is LDS-sized and saves the scope round trip */
            int64_t kscope[2];
            Q.flags = RFX_Q_REFUSE_NULL_KEY | /* the reference opens one group per null-key row (core/index.c:1808-1816): its own select answers those */
                      RFX_Q_SLICED;           /* the result is read through rfx_exec_groups_fetch_all only: every device may keep and read back its own slice */
            resident_t *ke = (g_nshards == 1 && nkeys == 1 && !parted && flat && nrows > 0 && nrows < ((int64_t)1 << 24) && !kxbar[0]) ? resident_entry(dks[0]) : NULL;
            if (ke) {
                if (!ke->scope_ok) {
                    int64_t c0 = 0;
                    if (rfx_hip_scope_i64(g_ctx, (const int64_t *)dks[0], NULL, 0, RFX_AND, nrows, &ke->smin, &ke->smax, &c0) != RFX_OK) { res = fail_hip("scope"); goto done; }
                    ke->scope_ok = 1;
                }
                kscope[0] = ke->smin;
                kscope[1] = ke->smax;
                Q.key_scope = kscope;
            }
            rfx_groups_t R;
            const int grc = rfx_exec_group_by(g_x, &Q, &R);
            tm_mark();
            if (grc == RFX_EXEC_NULL_KEY) { why = "null group key"; goto out; }
            if (grc == RFX_ESTATE && strstr(rfx_exec_last_error(g_x), "collision")) { why = "row-hash collision between two key tuples"; goto out; }
            if (grc == RFX_ELIMIT && g_nshards > 1) { why = "sharded table: shape the planner runs on one shard"; goto out; }
            if (grc != RFX_OK) { res = fail(rfx_exec_last_error(g_x)); goto done; }
            const sel_keys_t K = {nkeys, key_out_type, kenum, kcs};
            rfx_groups_t Rw;
            const rfx_groups_t *Ru = &R;
            if (rank_slices_mode()) { /* this rank's range of the groups only */
                int rank = 0;
                const int ranks = rfx_exec_ranks(g_x, &rank);
                int64_t g0 = 0, gn = R.groups;
                if (ranks > 1) rfx_exec_split(R.groups, ranks, rank, &g0, &gn);
                if (ranks > 1 && rfx_exec_groups_window(&R, g0, gn, &Rw) == RFX_OK) Ru = &Rw;
            }
            res = sel_build_groups(Ru, &M, &K, knames, &why);
            rfx_exec_groups_free(g_x, &R);
            tm_mark();
            if (!res) goto out;
            g_last_gpu = res->type == RFX_TYPE_TABLE;
            goto done;
        }
        rfx_value_t vals[RFX_MAX_AGGS];
        int64_t selected = 0;
        if (rfx_exec_filter_aggr(g_x, &Q, vals, &selected) != RFX_OK) { res = fail(rfx_exec_last_error(g_x)); goto done; }
        res = sel_build_scalar(vals, &M);
        g_last_gpu = 1;
        goto done;
    }
out: /* hand the query to the host -- with this call's device scratch released first (the host may fan out to its pool: HOST_CALL) */
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    ntmp = 0;
    qtmp_release();
    if (parted) parted_view_release();
    parted = 0;
    res = delegate_select(dict, why ? why : "unsupported");
done:
    for (int i = 0; i < ntmp; i++) rfx_hip_free(g_ctx, tmp[i]);
    qtmp_release();
    if (parted) parted_view_release();
    H.drop(host_tab);
    tm_mark();
    tm_print();
    if (take && g_last_gpu && res && res->type == RFX_TYPE_TABLE) { /* (a delegated query had its take: applied by ray_select) */
        obj_p tv = HOST_CALL(H.eval(take));
        if (tv && tv->type != RFX_TYPE_ERR) {
            obj_p cut = HOST_CALL(((rfx_binary_f)H.f[F_TAKE])(res, tv));
            H.drop(res);
            res = cut;
        } else {
            H.drop(res);
            res = tv;
        }
        if (tv && res != tv) H.drop(tv);
    }
    return res;
}
rfx_obj_p rfx_select(rfx_obj_p dict) {
    op_begin();
    g_last_gpu = 0;
    obj_p r = select_impl(dict);
    g_stat[g_last_gpu ? ST_SELECT_GPU : ST_SELECT_DELEGATED]++;
    op_end();
    return r;
}
