/* rfx_ops_plan.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * table access; where: / by: / output mappings turned into rfx_pred_t / rfx_agg_t descriptors over resident columns; masks of trees the fused form cannot carry. */
/* ------------------------------------------------------------------------------------------------ table access */
static obj_p table_col(obj_p tab, int64_t sym) {
    obj_p names = RFX_AS_LIST(tab)[0], cols = RFX_AS_LIST(tab)[1];
    for (int64_t i = 0; i < names->len; i++)
        if (RFX_AS_I64(names)[i] == sym) return RFX_AS_LIST(cols)[i];
    return NULL;
}
static obj_p dict_get(obj_p d, const char *key) {
    int64_t id = H.intern(key, (int64_t)strlen(key));
    obj_p keys = RFX_AS_LIST(d)[0], vals = RFX_AS_LIST(d)[1];
    for (int64_t i = 0; i < keys->len; i++)
        if (RFX_AS_I64(keys)[i] == id) return RFX_AS_LIST(vals)[i];
    return NULL;
}

/* ------------------------------------------------------------------------------------------------ planning */
typedef struct {
    rfx_pred_t preds[RFX_MAX_PREDS];
    int npred, logic;
} wplan_t;

/* one comparison `(op colsym atom|colsym)` -> descriptor; 0 ok, -1 unsupported shape */
/* device scratch a query's PREDICATES allocate (operands that are expressions): released at the end of rfx_select */
static struct { void *d[RFX_MAX_SHARDS]; } g_qtmp[2 * RFX_MAX_PREDS * 4]; /* (per shard: every shard evaluates its own rows) */
static int g_nqtmp;
static void qtmp_release(void) {
    for (int i = 0; i < g_nqtmp; i++)
        for (int s = 0; s < g_nshards; s++) {
            if (!g_qtmp[i].d[s]) continue;
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
            rfx_hip_free(g_ctxs[s], g_qtmp[i].d[s]);
        }
    if (g_nqtmp && g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    g_nqtmp = 0;
}
static int build_xnodes(obj_p tab, obj_p e, rfx_xnode_t *nodes, int *nn, int *ncols, const char **why);
/* a comparison operand that is an element-wise expression (op x y): the reference evaluates it first (eval -> binop_map), so do
 * we -- one pass into a scratch column (rfx_hip_eval_expr), then the comparison reads it like any column */
static int expr_operand(obj_p tab, obj_p e, const void **d, int *ctype) {
    rfx_xnode_t nodes[RFX_MAX_XNODES];
    int nn = 0, ncols = 0;
    const char *why = NULL;
    int top = build_xnodes(tab, e, nodes, &nn, &ncols, &why);
    if (top == -2) return -2;
    if (top < 0 || ncols == 0 || g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) return -1;
    obj_p tcols = RFX_AS_LIST(tab)[1];
    const int64_t nrows = tcols->len ? RFX_AS_LIST(tcols)[0]->len : 0;
    rfx_agg_t a;
    memset(&a, 0, sizeof(a));
    a.kind = RFX_AGG_SUM;
    a.col_type = RFX_I64;
    a.nxnodes = nn;
    a.xnodes = nodes;
    int32_t ot = RFX_I64;
    /* every shard evaluates ITS rows of the operand columns on its own context (a shard holds its row range only: one evaluation over
     * the whole length would read past shard 0's piece); the scratch column then is a column of the query like any other (qcol_add) */
    memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
    void **devs = g_qtmp[g_nqtmp++].d;
    int rc = RFX_OK;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        rfx_xnode_t mine[RFX_MAX_XNODES];
        int64_t n = nrows;
        if (g_nshards > 1) {
            rfx_exec_split(nrows, g_nshards, s, NULL, &n);
            for (int j = 0; j < nn; j++) {
                mine[j] = nodes[j];
                rfx_xoperand_t *o[2] = {&mine[j].l, &mine[j].r};
                for (int k = 0; k < 2; k++) {
                    if (o[k]->kind != RFX_XK_COL) continue;
                    const void *there = NULL;
                    for (int i = 0; i < g_nqcols && !there; i++)
                        if (g_qcols[i].d[0] == o[k]->d_col) there = g_qcols[i].d[s];
                    if (!there) rc = RFX_EINVAL; /* (cannot happen: build_xnodes made every column resident, shard by shard) */
                    o[k]->d_col = there;
                }
            }
            a.xnodes = mine;
            rfx_hip_ctx_bind_thread(g_ctxs[s]);
        }
        if (rc == RFX_OK) rc = rfx_hip_malloc(g_ctxs[s], &devs[s], (size_t)(n ? n : 1) * 8);
        if (rc == RFX_OK) rc = rfx_hip_eval_expr(g_ctxs[s], &a, n, devs[s], &ot);
    }
    if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
    if (rc != RFX_OK || qcol_add(devs) != RFX_OK) return -2;
    *d = devs[0];
    *ctype = ot;
    return 0;
}
/* the SYMBOL vector an ENUM column indexes: the global its key names (in-memory pair: the key symbol; mmapped: the key's characters sit
 * one page before the indices, core/util.h:103-105, core/binary.c:135-137).  NULL when it does not resolve; the caller drops it. */
static obj_p enum_domain(obj_p e) {
    int64_t key_id;
    if (e->mmod == RFX_MMOD_INTERNAL) key_id = RFX_AS_LIST(e)[0]->i64;
    else {
        const char *ks = (const char *)e - 4096 + sizeof(rfx_obj_t);
        key_id = H.intern(ks, (int64_t)strnlen(ks, 4096 - sizeof(rfx_obj_t)));
    }
    obj_p ka = H.i64(key_id);
    ka->type = -RFX_TYPE_SYMBOL;
    obj_p dom = H.eval(ka);
    H.drop(ka);
    if (dom && dom->type != RFX_TYPE_SYMBOL) {
        H.drop(dom);
        dom = NULL;
    }
    return dom;
}
#define RFX_ATTR_QUOTED 8 /* ATTR_QUOTED, core/ops.h:55: a symbol atom that stands for itself ('x), not for a column */
static int g_where_virtual, g_where_data; /* comparisons of the where: in flight that read the virtual column / data columns of a parted table */
static int plan_cmp(obj_p tab, obj_p e, rfx_pred_t *p) {
    if (e->type == RFX_TYPE_LIST && e->len == 2 && fn_id(RFX_AS_LIST(e)[0]) == F_NOT) {
        /* (not (cmp x y)) = the complementary comparison: the reference's order is total (nulls and NaN sort lowest, core/ops.h:97), so
         * exactly one of < == > holds for every pair of cells and the complement of a set of them is the rest */
        static const int COMPLEMENT[6] = {RFX_NE, RFX_EQ, RFX_GE, RFX_LE, RFX_GT, RFX_LT}; /* of EQ NE LT GT LE GE */
        const int rc = plan_cmp(tab, RFX_AS_LIST(e)[1], p);
        if (rc == 0) p->op = COMPLEMENT[p->op];
        return rc;
    }
    if (e->type != RFX_TYPE_LIST || e->len != 3) return -1;
    int f = fn_id(RFX_AS_LIST(e)[0]);
    if (f < F_EQ || f > F_GE) return -1;
    obj_p l = RFX_AS_LIST(e)[1], r = RFX_AS_LIST(e)[2];
    memset(p, 0, sizeof(*p));
    p->op = f - F_EQ; /* F_EQ..F_GE are in RFX_EQ..RFX_GE order */
    const void *d;
    int64_t llen = -1;
    int lvirt = 0, ldate = 0;
    if (l->type == RFX_TYPE_LIST) {
        int ct = RFX_I64, rc0 = expr_operand(tab, l, &d, &ct);
        if (rc0) return rc0;
        p->col_type = ct;
    } else {
        if (l->type != -RFX_TYPE_SYMBOL) return -1;
        obj_p lc = table_col(tab, l->i64);
        if (lc && lc->type == RFX_TYPE_ENUM) {
            /* (== enum-column 'sym): the reference compares the domain's symbol at every index with the atom (MTYPE2(TYPE_ENUM, -TYPE_SYMBOL),
             * core/cmp.c:260-281); the symbol's place in the domain is found once on the host and the INDEX column is compared on the
             * device -- a symbol the domain does not hold selects nothing (index -1).  Only == : the other operators are the host's. */
            if (f != F_EQ || r->type != -RFX_TYPE_SYMBOL || !(r->attrs & RFX_ATTR_QUOTED)) return -1;
            if (lc->mmod != RFX_MMOD_INTERNAL) return -1; /* an mmapped enum (splayed table): the reference's own `where:` answers `type` there -- the host's to say */
            obj_p dom = enum_domain(lc);
            if (!dom) return -1;
            int64_t at = -1;
            for (int64_t i = 0; i < dom->len && at < 0; i++)
                if (RFX_AS_I64(dom)[i] == r->i64) at = i;
            H.drop(dom);
            if (resident(enum_indices(lc), 0, &d) != RFX_OK) return -2;
            g_where_data++;
            p->d_col = d;
            p->col_type = RFX_I64;
            p->rhs_type = RFX_I64;
            p->rhs_i = at;
            return 0;
        }
        if (lc && IS_I32_FAMILY(lc->type) && !(g_npx && proxy_of(lc))) { /* (a parted table's 4-byte columns are the host's: proxies upload 8-byte partitions only) */
            /* a 4-byte integer column (I32 / DATE / TIME) in a comparison: its widened device copy against an atom or a column of the
             * types the reference's i32 arms take (core/cmp.c:148-166: the same 4-byte type; for I32 also I64 / F64, promoted as
             * i32_to_i64 / i32_to_f64 do -- which is what the widened column compares as) */
            if (resident(lc, 0, &d) != RFX_OK) return -2;
            g_where_data++;
            p->d_col = d;
            p->col_type = RFX_I64;
            const int8_t lt = lc->type;
            if (r->type == -lt) { p->rhs_type = RFX_I64; p->rhs_i = r->i32 == INT32_MIN ? RFX_NULL_I64 : (int64_t)r->i32; return 0; }
            if (lt == RFX_TYPE_I32 && r->type == -RFX_TYPE_I64) { p->rhs_type = RFX_I64; p->rhs_i = r->i64; return 0; }
            if (lt == RFX_TYPE_I32 && r->type == -RFX_TYPE_F64) { p->rhs_type = RFX_F64; p->rhs_f = r->f64; return 0; }
            if (r->type == -RFX_TYPE_SYMBOL && !(r->attrs & RFX_ATTR_QUOTED)) {
                obj_p rc = table_col(tab, r->i64);
                if (!rc || rc->len != lc->len) return -1;
                if (!(rc->type == lt || (lt == RFX_TYPE_I32 && (rc->type == RFX_TYPE_I64 || rc->type == RFX_TYPE_F64)))) return -1;
                if (resident(rc, 0, &d) != RFX_OK) return -2;
                p->d_rhs_col = d;
                p->rhs_type = rc->type == RFX_TYPE_F64 ? RFX_F64 : RFX_I64;
                return 0;
            }
            return -1;
        }
        if (!lc || !col_ctype(lc)) return -1;
        p->col_type = col_ctype(lc);
        if (resident(lc, 0, &d) != RFX_OK) return -2;
        llen = lc->len;
        const proxy_t *px = g_npx ? proxy_of(lc) : NULL;
        lvirt = px && px->kind == 2;
        ldate = lvirt && px->vtype == RFX_TYPE_DATE;
    }
    if (lvirt) g_where_virtual++;
    else g_where_data++;
    p->d_col = d;
    if (r->type == -RFX_TYPE_I64) { p->rhs_type = RFX_I64; p->rhs_i = r->i64; }
    else if (r->type == -RFX_TYPE_TIMESTAMP && l->type == -RFX_TYPE_SYMBOL && table_col(tab, l->i64) && table_col(tab, l->i64)->type == RFX_TYPE_TIMESTAMP) {
        p->rhs_type = RFX_I64; /* a TIMESTAMP column against a timestamp atom: nanoseconds as i64 on both sides (core/cmp.c) */
        p->rhs_i = r->i64;
    }
    else if (r->type == -RFX_TYPE_DATE && ldate) { p->rhs_type = RFX_I64; p->rhs_i = (int64_t)r->i32; } /* (== Date 2024.01.03): partition pruning, core/cmp.c:341-358 */
    else if (r->type == -RFX_TYPE_F64) { p->rhs_type = RFX_F64; p->rhs_f = r->f64; }
    else if (r->type == -RFX_TYPE_SYMBOL && (r->attrs & RFX_ATTR_QUOTED)) {
        /* a quoted symbol is a value, never a column name -- even when the table has a column of that name (eval_sym, core/eval.c:829):
         * a SYMBOL column compares its interned ids with it (== and != ; the ordering of symbols is the host's business) */
        obj_p lc = (l->type == -RFX_TYPE_SYMBOL) ? table_col(tab, l->i64) : NULL;
        if (!lc || lc->type != RFX_TYPE_SYMBOL || (f != F_EQ && f != F_NE)) return -1;
        p->rhs_type = RFX_I64;
        p->rhs_i = r->i64;
    } else if (r->type == -RFX_TYPE_SYMBOL) {
        obj_p rc = table_col(tab, r->i64);
        if (!rc || !col_ctype(rc) || (llen >= 0 && rc->len != llen)) return -1;
        g_where_data++;
        if (resident(rc, 0, &d) != RFX_OK) return -2;
        p->d_rhs_col = d;
        p->rhs_type = col_ctype(rc);
    } else if (r->type == RFX_TYPE_LIST) {
        int ct = RFX_I64, rc0 = expr_operand(tab, r, &d, &ct);
        if (rc0) return rc0;
        p->d_rhs_col = d;
        p->rhs_type = ct;
    } else return -1;
    return 0;
}
/* (within col [lo hi]) = lo <= col <= hi (ray_within, core/items.c:848-872: an I64 column against a two-element I64 vector, raw integer
 * order) and (in col [v1 .. vn]) = col == v1 or ... (ray_in, core/items.c:736+ -> index_in_i64_i64: raw equality; I64 / TIMESTAMP / SYMBOL
 * columns against a vector of their own type) as comparisons of the fused pass: appends them to out[0 .. room) and says through *glogic
 * how they combine among themselves.  Returns how many (>= 1), -1 when `e` is not such a form (or too long), -2 on an upload error. */
static int plan_set_cmp(obj_p tab, obj_p e, rfx_pred_t *out, int room, int *glogic) {
    if (!e || e->type != RFX_TYPE_LIST || e->len != 3) return -1;
    const int f = fn_id(RFX_AS_LIST(e)[0]);
    if (f != F_IN && f != F_WITHIN) return -1;
    obj_p l = RFX_AS_LIST(e)[1], r = RFX_AS_LIST(e)[2];
    if (l->type != -RFX_TYPE_SYMBOL || (l->attrs & RFX_ATTR_QUOTED) || r->type <= 0) return -1;
    obj_p lc = table_col(tab, l->i64);
    if (!lc || g_npx) return -1; /* (parted tables: the reference prunes partitions through these forms -- not taken apart here) */
    int n;
    if (f == F_WITHIN) {
        if (lc->type != RFX_TYPE_I64 || r->type != RFX_TYPE_I64 || r->len != 2) return -1;
        n = 2;
        *glogic = RFX_AND;
    } else {
        if (!(lc->type == RFX_TYPE_I64 || lc->type == RFX_TYPE_TIMESTAMP || lc->type == RFX_TYPE_SYMBOL) || r->type != lc->type || r->len < 1 || r->len > RFX_MAX_PREDS) return -1;
        n = (int)r->len;
        *glogic = RFX_OR;
    }
    if (n > room) return -1;
    const void *d;
    if (resident(lc, 0, &d) != RFX_OK) return -2;
    g_where_data++;
    for (int i = 0; i < n; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        out[i].d_col = d;
        out[i].col_type = RFX_I64;
        out[i].rhs_type = RFX_I64;
        out[i].rhs_i = RFX_AS_I64(r)[i];
        out[i].op = f == F_WITHIN ? (i == 0 ? RFX_GE : RFX_LE) : RFX_EQ;
    }
    return n;
}
/* where: a comparison, or ANY tree of and / or over comparisons (logic_map nests freely, core/logic.c:89-260) -- its leaves in order, each
 * with the depth of parentheses it sits in and the parentheses that close after it.  Level 0 combines with the root's operator, every
 * deeper level with the opposite of the level above: the same operator nested in itself is associative and stays on its level.  Up to
 * RFX_MAX_PREDS comparisons and four levels run in ONE fused pass (rfx_pred_t: the two-level `more` form where it suffices -- the
 * kernels' short path -- else the RFX_PRED_TREE form); anything beyond: -1 (the mask path answers it). */
typedef struct {
    int dep[RFX_MAX_PREDS], clo[RFX_MAX_PREDS];
} wtree_t;
static int plan_node(obj_p tab, obj_p e, int level_op, int depth, wplan_t *wp, wtree_t *wt) {
    if (!e || e->type != RFX_TYPE_LIST || e->len < 1) return -1;
    const int f = fn_id(RFX_AS_LIST(e)[0]);
    if (f == F_IN || f == F_WITHIN) { /* a group of comparisons: on this level when it combines like it, else a parenthesis of its own */
        int gl = RFX_AND;
        const int n = plan_set_cmp(tab, e, &wp->preds[wp->npred], RFX_MAX_PREDS - wp->npred, &gl);
        if (n < 0) return n;
        const int own = n > 1 && gl != (level_op == F_AND ? RFX_AND : RFX_OR);
        for (int i = 0; i < n; i++) {
            wt->dep[wp->npred + i] = depth + own;
            wt->clo[wp->npred + i] = 0;
        }
        if (own) wt->clo[wp->npred + n - 1] = 1;
        wp->npred += n;
        return 0;
    }
    if (f != F_AND && f != F_OR) {
        if (wp->npred >= RFX_MAX_PREDS) return -1;
        const int rc = plan_cmp(tab, e, &wp->preds[wp->npred]);
        if (rc) return rc;
        wt->dep[wp->npred] = depth;
        wt->clo[wp->npred] = 0;
        wp->npred++;
        return 0;
    }
    if (e->len < 2) return -1;
    const int own = f != level_op; /* the opposite operator: a parenthesis one level down, closed after its last leaf */
    const int first = wp->npred;
    for (int64_t i = 1; i < e->len; i++) {
        const int rc = plan_node(tab, RFX_AS_LIST(e)[i], f, depth + own, wp, wt);
        if (rc) return rc;
    }
    if (own && wp->npred > first) wt->clo[wp->npred - 1]++;
    return 0;
}
static int plan_where(obj_p tab, obj_p w, wplan_t *wp) {
    wp->npred = 0;
    wp->logic = RFX_AND;
    if (!w) return 0;
    if (w->type != RFX_TYPE_LIST || w->len < 1) return -1;
    const int f = fn_id(RFX_AS_LIST(w)[0]);
    wtree_t wt;
    if (f == F_AND || f == F_OR) {
        if (w->len < 2) return -1;
        wp->logic = (f == F_AND) ? RFX_AND : RFX_OR;
    } else if (f == F_IN || f == F_WITHIN) {
        const int n = plan_set_cmp(tab, w, wp->preds, RFX_MAX_PREDS, &wp->logic);
        if (n < 0) return n;
        wp->npred = n;
        return 0;
    }
    const int rc = plan_node(tab, w, (f == F_AND || f == F_OR) ? f : F_AND, 0, wp, &wt);
    if (rc) return rc;
    int maxd = 0;
    for (int i = 0; i < wp->npred; i++) {
        if (wt.dep[i] > maxd) maxd = wt.dep[i];
        if (wt.clo[i] > wt.dep[i]) return -1; /* (cannot happen: a parenthesis closes on the level it opened) */
    }
    if (maxd > 3 || (maxd > 1 && wp->npred < 3)) return -1; /* deeper than four levels (or a degenerate nest of one-armed parentheses): through masks */
    if (maxd <= 1) { /* flat, or parentheses of the opposite operator over comparisons: the two-level form */
        for (int i = 0; i < wp->npred; i++) wp->preds[i].more = (wt.dep[i] == 1 && wt.clo[i] == 0) ? 1 : 0;
        return 0;
    }
    for (int i = 0; i < wp->npred; i++) wp->preds[i].more = RFX_PRED_LEAF(wt.dep[i], wt.clo[i]);
    return 0;
}


/* ---- nested boolean trees: evaluated the way the reference does (mask per comparison, and/or in place, where), but on
 * the GPU: core/cmp.c -> K2 rfx_hip_cmp_mask, core/logic.c -> rfx_hip_mask_logic, core/ops.c:254 -> K3 ---- */
static int mask_of_expr(obj_p tab, obj_p e, int64_t nrows, int8_t **out) {
    *out = NULL;
    if (!e || e->type != RFX_TYPE_LIST || e->len < 2) return -1;
    int f = fn_id(RFX_AS_LIST(e)[0]);
    void *m = NULL;
    if (f >= F_EQ && f <= F_GE) {
        rfx_pred_t p;
        int rc = plan_cmp(tab, e, &p);
        if (rc) return rc;
        if (rfx_hip_malloc(g_ctx, &m, (size_t)nrows + 16) != RFX_OK) return -2;
        if (rfx_hip_cmp_mask(g_ctx, &p, nrows, (int8_t *)m) != RFX_OK) { rfx_hip_free(g_ctx, m); return -2; }
        *out = (int8_t *)m;
        return 0;
    }
    if (f != F_AND && f != F_OR) return -1;
    int8_t *acc = NULL;
    for (int64_t i = 1; i < e->len; i++) {
        int8_t *sub = NULL;
        int rc = mask_of_expr(tab, RFX_AS_LIST(e)[i], nrows, &sub);
        if (rc) { if (acc) rfx_hip_free(g_ctx, acc); return rc; }
        if (!acc) acc = sub;
        else {
            rc = rfx_hip_mask_logic(g_ctx, f == F_AND ? RFX_AND : RFX_OR, acc, sub, 0, nrows);
            rfx_hip_free(g_ctx, sub);
            if (rc != RFX_OK) { rfx_hip_free(g_ctx, acc); return -2; }
        }
    }
    *out = acc;
    return 0;
}

/* ... over the shards (round 5): every comparison of the tree as a mask piece per shard (its operands are resident shard by shard), and / or in place per
 * shard, and the finished mask WIDENED to an i64 column of 0 / 1 that the query reads as ONE comparison `(!= m 0)` of its fused pass on every shard -- no
 * ids, no gathered columns, first rows stay table rows.  The column is a per-call scratch column of the query (g_qtmp, qcol_add).  0 / -1 shape / -2 device */
static int mask_tree_sharded(obj_p tab, obj_p e, int64_t nrows, int8_t **m) {
    for (int s = 0; s < RFX_MAX_SHARDS; s++) m[s] = NULL;
    if (!e || e->type != RFX_TYPE_LIST || e->len < 2) return -1;
    const int f = fn_id(RFX_AS_LIST(e)[0]);
    int rc = 0;
    if (f >= F_EQ && f <= F_GE) {
        rfx_pred_t p;
        if ((rc = plan_cmp(tab, e, &p)) != 0) return rc;
        for (int s = 0; s < g_nshards && rc == 0; s++) {
            int64_t n;
            rfx_exec_split(nrows, g_nshards, s, NULL, &n);
            rfx_pred_t ps = p;
            for (int k = 0; k < g_nqcols && s > 0; k++) {
                if (g_qcols[k].d[0] == p.d_col) ps.d_col = g_qcols[k].d[s];
                if (p.d_rhs_col && g_qcols[k].d[0] == p.d_rhs_col) ps.d_rhs_col = g_qcols[k].d[s];
            }
            rfx_hip_ctx_bind_thread(g_ctxs[s]);
            void *b = NULL;
            if (rfx_hip_malloc(g_ctxs[s], &b, (size_t)n + 16) != RFX_OK) rc = -2;
            m[s] = (int8_t *)b;
            if (rc == 0 && n > 0 && rfx_hip_cmp_mask(g_ctxs[s], &ps, n, m[s]) != RFX_OK) rc = -2;
        }
    } else if (f == F_AND || f == F_OR) {
        for (int64_t i = 1; i < e->len && rc == 0; i++) {
            int8_t *sub[RFX_MAX_SHARDS];
            rc = mask_tree_sharded(tab, RFX_AS_LIST(e)[i], nrows, sub);
            for (int s = 0; s < g_nshards; s++) {
                int64_t n;
                rfx_exec_split(nrows, g_nshards, s, NULL, &n);
                rfx_hip_ctx_bind_thread(g_ctxs[s]);
                if (rc == 0 && !m[s]) { m[s] = sub[s]; continue; }
                if (rc == 0 && n > 0 && rfx_hip_mask_logic(g_ctxs[s], f == F_AND ? RFX_AND : RFX_OR, m[s], sub[s], 0, n) != RFX_OK) rc = -2;
                if (sub[s]) { rfx_hip_ctx_sync(g_ctxs[s]); rfx_hip_free(g_ctxs[s], sub[s]); }
            }
        }
    } else rc = -1;
    if (rc != 0)
        for (int s = 0; s < g_nshards; s++)
            if (m[s]) { rfx_hip_ctx_bind_thread(g_ctxs[s]); rfx_hip_ctx_sync(g_ctxs[s]); rfx_hip_free(g_ctxs[s], m[s]); m[s] = NULL; }
    rfx_hip_ctx_bind_thread(g_ctx);
    return rc;
}
static int mask_column_sharded(obj_p tab, obj_p where, int64_t nrows, const void **d_col) {
    int8_t *m[RFX_MAX_SHARDS];
    int rc = mask_tree_sharded(tab, where, nrows, m);
    if (rc != 0) return rc;
    if (g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) rc = -2;
    void *devs[RFX_MAX_SHARDS];
    if (rc == 0 && shards_alloc(devs, nrows, 8, 0) != RFX_OK) rc = -2;
    if (rc == 0) {
        memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
        for (int s = 0; s < g_nshards; s++) g_qtmp[g_nqtmp].d[s] = devs[s];
        g_nqtmp++;
    }
    for (int s = 0; s < g_nshards; s++) {
        int64_t n;
        rfx_exec_split(nrows, g_nshards, s, NULL, &n);
        rfx_hip_ctx_bind_thread(g_ctxs[s]);
        if (rc == 0 && n > 0 && rfx_hip_widen_b8(g_ctxs[s], m[s], n, (int64_t *)devs[s]) != RFX_OK) rc = -2;
        if (m[s]) { rfx_hip_ctx_sync(g_ctxs[s]); rfx_hip_free(g_ctxs[s], m[s]); }
    }
    rfx_hip_ctx_bind_thread(g_ctx);
    if (rc == 0 && qcol_add(devs) != RFX_OK) rc = -2;
    if (rc == 0) *d_col = devs[0];
    return rc;
}

/* selection of `where` as ascending device row ids (flat predicates fused, nested trees through masks) */
static int where_ids(obj_p tab, obj_p where, const wplan_t *wp, int flat, int64_t nrows, int64_t **d_ids, int64_t *count) {
    *d_ids = NULL;
    *count = 0;
    int8_t *mask = NULL;
    int rc;
    if (flat) {
        /* one pass over the predicate columns (rfx_where_once.hip): buffer by sampled estimate, exact count back, a second run if the
         * sample underestimated a clustered selection */
        int64_t cap = 0;
        void *d = NULL;
        if (rfx_hip_where_estimate(g_ctx, wp->preds, wp->npred, wp->logic, nrows, &cap) != RFX_OK) return -2;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (cap > 0 && rfx_hip_malloc(g_ctx, &d, (size_t)cap * 8) != RFX_OK) return -2;
            const int wrc = rfx_hip_where_once(g_ctx, wp->preds, wp->npred, wp->logic, nrows, 0, (int64_t *)d, cap, count);
            if (wrc == RFX_OK) {
                if (*count > 0) *d_ids = (int64_t *)d;
                else if (d) rfx_hip_free(g_ctx, d);
                return 0;
            }
            if (d) rfx_hip_free(g_ctx, d);
            d = NULL;
            if (wrc != RFX_ELIMIT || *count <= cap) break;
            cap = *count;
        }
        *count = 0;
        return -2;
    } else {
        rc = mask_of_expr(tab, where, nrows, &mask);
        if (rc == 0) rc = rfx_hip_where_begin(g_ctx, NULL, 0, RFX_AND, mask, nrows, count) == RFX_OK ? 0 : -2;
    }
    if (rc == 0 && *count > 0) {
        void *d = NULL;
        if (rfx_hip_malloc(g_ctx, &d, (size_t)*count * 8) != RFX_OK || rfx_hip_where_emit(g_ctx, 0, (int64_t *)d) != RFX_OK) {
            if (d) rfx_hip_free(g_ctx, d);
            rc = -2;
        } else *d_ids = (int64_t *)d;
    }
    if (mask) rfx_hip_free(g_ctx, mask);
    return rc;
}

static obj_p value_atom(const rfx_value_t *v) { return v->type == RFX_F64 ? H.f64(v->f) : H.i64(v->i); }
static obj_p one_row(const rfx_value_t *v) {
    obj_p c = H.vector(v->type == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64, 1);
    RFX_AS_I64(c)[0] = v->i;
    return c;
}

static obj_p refused1(int f, obj_p x) {
    if (g_refused_sharded && H.bound == 1 && f >= 0 && f < F_N && H.f[f]) return HOST_CALL(((rfx_unary_f)H.f[f])(x));
    return fail_ctx();
}
__attribute__((unused)) static obj_p refused2(int f, obj_p x, obj_p y) {
    if (g_refused_sharded && H.bound == 1 && f >= 0 && f < F_N && H.f[f]) return HOST_CALL(((rfx_binary_f)H.f[f])(x, y));
    return fail_ctx();
}
static obj_p refusedn(int f, obj_p *x, int64_t n) {
    if (g_refused_sharded && H.bound == 1 && f >= 0 && f < F_N && H.f[f]) return HOST_CALL(((rfx_vary_f)H.f[f])(x, n));
    return fail_ctx();
}
static obj_p delegate_select(obj_p dict, const char *why) {
    g_last_gpu = 0;
    snprintf(g_err, sizeof(g_err), "rfx_select: handed to the host (%s)", why); /* rfx_ops_last_error(): why the last query was delegated */
    if (getenv("RFX_TRACE")) fprintf(stderr, "[rfx] select delegated: %s\n", why);
    if (H.bound == 1 && H.f[F_SELECT]) return HOST_CALL(((rfx_unary_f)H.f[F_SELECT])(dict));
    char b[300];
    snprintf(b, sizeof(b), "rfx_select: query shape not covered by the MI355X path (%s) and no host ray_select to delegate to", why);
    return fail(b);
}

/* (op x y) with x / y a column symbol, an i64 / f64 atom or another such list -> nodes in evaluation order (rfx_xnode_t).
 * Returns the index of the node holding the value, -1 with *why set when the shape is not covered, -2 on an upload error. */
static int build_xnodes(obj_p tab, obj_p e, rfx_xnode_t *nodes, int *nn, int *ncols, const char **why) {
    if (e->type != RFX_TYPE_LIST || e->len != 3) { *why = "expression is not (op x y)"; return -1; }
    int xf = fn_id(RFX_AS_LIST(e)[0]);
    if (xf < F_ADD || xf > F_MOD) { *why = "expression operator is not + - * div / %"; return -1; }
    rfx_xnode_t node;
    memset(&node, 0, sizeof(node));
    node.op = RFX_X_ADD + (xf - F_ADD);
    rfx_xoperand_t *ops[2] = {&node.l, &node.r};
    for (int j = 0; j < 2; j++) {
        obj_p x = RFX_AS_LIST(e)[1 + j];
        if (x->type == -RFX_TYPE_SYMBOL) {
            obj_p c = table_col(tab, x->i64);
            if (!c || !(c->type == RFX_TYPE_I64 || c->type == RFX_TYPE_F64)) { *why = "expression operand column type"; return -1; }
            const void *d;
            if (resident(c, 0, &d) != RFX_OK) return -2;
            ops[j]->kind = RFX_XK_COL;
            ops[j]->type = col_ctype(c);
            ops[j]->d_col = d;
            (*ncols)++;
        } else if (x->type == -RFX_TYPE_I64) {
            ops[j]->kind = RFX_XK_ATOM;
            ops[j]->type = RFX_I64;
            ops[j]->i = x->i64;
        } else if (x->type == -RFX_TYPE_F64) {
            ops[j]->kind = RFX_XK_ATOM;
            ops[j]->type = RFX_F64;
            ops[j]->f = x->f64;
        } else if (x->type == RFX_TYPE_LIST) {
            int sub = build_xnodes(tab, x, nodes, nn, ncols, why);
            if (sub < 0) return sub;
            ops[j]->kind = RFX_XK_NODE;
            ops[j]->node = sub;
        } else { *why = "expression operand is neither a column, an i64/f64 atom nor an expression"; return -1; }
    }
    if (*nn >= RFX_MAX_XNODES) { *why = "expression deeper than RFX_MAX_XNODES operations"; return -1; }
    nodes[*nn] = node;
    return (*nn)++;
}
