/* rfx_ops_operators.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * the single operators: comparisons, arithmetic, and / or (+ special forms), where, the sharded folds' helpers. */
/* ------------------------------------------------------------------------------------------------ single operators */
/* ---- element-wise operators over the shards (round 5): every shard works on its rows of the operands and its piece of the result goes to the
 * host vector at the piece's offset (the reference maps them over its pool in row chunks: cmp_map core/cmp.c:35-68, binop_map core/math.c:2280-2345) ---- */
static int transient_sharded(obj_p v, const void **dev); /* (below, with the sharded folds' helpers) */
static const void *shard_piece(const void *p, int s) {
    if (!p || s == 0) return p;
    for (int i = 0; i < g_nqcols; i++)
        if (g_qcols[i].d[0] == p) return g_qcols[i].d[s];
    return NULL;
}
typedef int (*piece_fn)(void *arg, int s, int64_t r0, int64_t n, void *d_out); /* the shard's kernel(s), enqueued on g_ctxs[s] */
static int map_shards(obj_p out, size_t esz, piece_fn fn, void *arg) {
    void *dout[RFX_MAX_SHARDS] = {0};
    const int64_t len = out->len;
    int rc = RFX_OK;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) { /* everything enqueued first ... */
        int64_t r0, n;
        rfx_exec_split(len, g_nshards, s, &r0, &n);
        if (n <= 0) continue;
        rfx_hip_ctx_bind_thread(g_ctxs[s]);
        rc = rfx_hip_malloc(g_ctxs[s], &dout[s], (size_t)n * esz + 16);
        if (rc == RFX_OK) rc = fn(arg, s, r0, n, dout[s]);
        if (rc == RFX_OK) rc = rfx_hip_d2h_async(g_ctxs[s], (char *)RFX_AS_RAW(out) + (size_t)r0 * esz, dout[s], (size_t)n * esz);
    }
    for (int s = 0; s < g_nshards; s++) { /* ... then one wait per shard */
        if (!dout[s]) continue;
        rfx_hip_ctx_bind_thread(g_ctxs[s]);
        const int src = rfx_hip_ctx_sync(g_ctxs[s]);
        if (rc == RFX_OK) rc = src;
        rfx_hip_free(g_ctxs[s], dout[s]);
    }
    rfx_hip_ctx_bind_thread(g_ctx);
    return rc;
}
typedef struct { rfx_pred_t p; } cmp_arg_t;
static int cmp_piece(void *arg, int s, int64_t r0, int64_t n, void *d_out) {
    (void)r0;
    rfx_pred_t p = ((cmp_arg_t *)arg)->p;
    p.d_col = shard_piece(p.d_col, s);
    p.d_rhs_col = shard_piece(p.d_rhs_col, s);
    return rfx_hip_cmp_mask(g_ctxs[s], &p, n, (int8_t *)d_out);
}
typedef struct { rfx_agg_t a; int32_t ot; } arith_arg_t;
static int arith_piece(void *arg, int s, int64_t r0, int64_t n, void *d_out) {
    (void)r0;
    rfx_agg_t a = ((arith_arg_t *)arg)->a;
    a.d_col = shard_piece(a.d_col, s);
    a.d_xrhs_col = shard_piece(a.d_xrhs_col, s);
    int32_t ot = RFX_I64;
    return rfx_hip_eval_expr(g_ctxs[s], &a, n, d_out, &ot);
}
typedef struct { int logic; int64_t nin; const void *d_in[16]; } logic_arg_t;
static int logic_piece(void *arg, int s, int64_t r0, int64_t n, void *d_out) {
    (void)r0;
    logic_arg_t *L = (logic_arg_t *)arg;
    int rc = rfx_hip_d2d(g_ctxs[s], d_out, shard_piece(L->d_in[0], s), (size_t)n);
    for (int64_t i = 1; i < L->nin && rc == RFX_OK; i++) rc = rfx_hip_mask_logic(g_ctxs[s], L->logic, (int8_t *)d_out, (const int8_t *)shard_piece(L->d_in[i], s), 0, n);
    return rc;
}

static obj_p cmp_impl(int op, obj_p x, obj_p y) {
    rfx_host_bind();
    if (!x || !y) return fail("cmp: null argument");
    if (!(x->type > 0 && col_ctype(x) && (y->type == -RFX_TYPE_I64 || y->type == -RFX_TYPE_F64 || (y->type > 0 && col_ctype(y))))) {
        if (H.bound == 1 && H.f[F_EQ + op]) return HOST_CALL(((rfx_binary_f)H.f[F_EQ + op])(x, y));
        return fail("cmp: only i64/f64 column (x) atom|column runs on the MI355X path");
    }
    if (y->type > 0 && y->len != x->len) return fail("length"); /* err_length, core/cmp.c:633-640 */
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    rfx_pred_t p;
    memset(&p, 0, sizeof(p));
    const void *d;
    if (resident(x, 0, &d) != RFX_OK) return fail_hip("column upload");
    p.d_col = d;
    p.col_type = col_ctype(x);
    p.op = op;
    if (y->type == -RFX_TYPE_I64) { p.rhs_type = RFX_I64; p.rhs_i = y->i64; }
    else if (y->type == -RFX_TYPE_F64) { p.rhs_type = RFX_F64; p.rhs_f = y->f64; }
    else {
        if (resident(y, 0, &d) != RFX_OK) return fail_hip("column upload");
        p.d_rhs_col = d;
        p.rhs_type = col_ctype(y);
    }
    if (g_nshards > 1) {
        cmp_arg_t A = {p};
        obj_p outs = H.vector(RFX_TYPE_B8, x->len);
        if (map_shards(outs, 1, cmp_piece, &A) != RFX_OK) { H.drop(outs); return fail_hip("cmp_mask"); }
        return outs;
    }
    void *dm = NULL;
    if (rfx_hip_malloc(g_ctx, &dm, (size_t)x->len + 8) != RFX_OK) return fail_hip("mask");
    obj_p out = H.vector(RFX_TYPE_B8, x->len);
    int ok = rfx_hip_cmp_mask(g_ctx, &p, x->len, (int8_t *)dm) == RFX_OK && rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dm, (size_t)x->len) == RFX_OK;
    rfx_hip_free(g_ctx, dm);
    if (!ok) { H.drop(out); return fail_hip("cmp_mask"); }
    return out;
}
static obj_p cmp_op(int op, obj_p x, obj_p y) {
    op_begin();
    obj_p r = cmp_impl(op, x, y);
    op_end();
    return r;
}
/* ray_add / ray_sub / ray_mul / ray_fdiv / ray_div / ray_mod over an i64 / f64 vector and a vector or atom (binop_map, core/math.c:2280-2345) */
static obj_p arith_impl(int xop, int fidx, obj_p x, obj_p y) {
    rfx_host_bind();
    if (!x || !y) return fail("arith: null argument");
    const int xv = x->type > 0 && col_ctype(x) && x->type != RFX_TYPE_SYMBOL, yv = y->type > 0 && col_ctype(y) && y->type != RFX_TYPE_SYMBOL;
    const int xa = x->type == -RFX_TYPE_I64 || x->type == -RFX_TYPE_F64, ya = y->type == -RFX_TYPE_I64 || y->type == -RFX_TYPE_F64;
    if (!((xv && (yv || ya)) || (xa && yv))) {
        if (H.bound == 1 && H.f[fidx]) return HOST_CALL(((rfx_binary_f)H.f[fidx])(x, y));
        return fail("arith: only i64/f64 vector (x) vector|atom runs on the MI355X path");
    }
    if (xv && yv && x->len != y->len) return fail("length");
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    rfx_agg_t a;
    memset(&a, 0, sizeof(a));
    a.kind = RFX_AGG_SUM;
    a.xop = xop;
    obj_p col = xv ? x : y, other = xv ? y : x;
    if (!xv) a.xflags = RFX_XF_SWAP; /* atom (op) vector */
    const void *d;
    if (resident(col, 0, &d) != RFX_OK) return fail_hip("column upload");
    a.d_col = d;
    a.col_type = col_ctype(col);
    if (other->type > 0) {
        if (resident(other, 0, &d) != RFX_OK) return fail_hip("column upload");
        a.d_xrhs_col = d;
        a.xrhs_type = col_ctype(other);
    } else if (other->type == -RFX_TYPE_I64) { a.xrhs_type = RFX_I64; a.xrhs_i = other->i64; }
    else { a.xrhs_type = RFX_F64; a.xrhs_f = other->f64; }
    const int64_t n = col->len;
    if (g_nshards > 1) {
        arith_arg_t A;
        A.a = a;
        A.ot = RFX_I64;
        if (rfx_hip_eval_expr(g_ctx, &a, 0, NULL, &A.ot) != RFX_OK) return fail_hip("eval_expr"); /* (no rows: the result type only) */
        obj_p outs = H.vector(A.ot == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64, n);
        if (map_shards(outs, 8, arith_piece, &A) != RFX_OK) { H.drop(outs); return fail_hip("eval_expr"); }
        return outs;
    }
    void *dout = NULL;
    if (rfx_hip_malloc(g_ctx, &dout, (size_t)(n ? n : 1) * 8) != RFX_OK) return fail_hip("arith");
    int32_t ot = RFX_I64;
    int ok = rfx_hip_eval_expr(g_ctx, &a, n, dout, &ot) == RFX_OK;
    obj_p out = NULL;
    if (ok) {
        out = H.vector(ot == RFX_F64 ? RFX_TYPE_F64 : RFX_TYPE_I64, n);
        ok = n == 0 || rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), dout, (size_t)n * 8) == RFX_OK;
    }
    rfx_hip_free(g_ctx, dout);
    if (!ok) { if (out) H.drop(out); return fail_hip("eval_expr"); }
    return out;
}
static obj_p arith_op(int xop, int fidx, obj_p x, obj_p y) {
    op_begin();
    obj_p r = arith_impl(xop, fidx, x, y);
    op_end();
    return r;
}
rfx_obj_p rfx_add(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_ADD, F_ADD, x, y); }
rfx_obj_p rfx_sub(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_SUB, F_SUB, x, y); }
rfx_obj_p rfx_mul(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_MUL, F_MUL, x, y); }
rfx_obj_p rfx_div(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_FDIV, F_FDIV, x, y); }
rfx_obj_p rfx_floordiv(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_DIV, F_DIV, x, y); } /* the reference's `/` (ray_div) */
rfx_obj_p rfx_mod(rfx_obj_p x, rfx_obj_p y) { return arith_op(RFX_X_MOD, F_MOD, x, y); }      /* `%` (ray_mod) */

rfx_obj_p rfx_eq(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_EQ, x, y); }
rfx_obj_p rfx_ne(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_NE, x, y); }
rfx_obj_p rfx_lt(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_LT, x, y); }
rfx_obj_p rfx_gt(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_GT, x, y); }
rfx_obj_p rfx_le(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_LE, x, y); }
rfx_obj_p rfx_ge(rfx_obj_p x, rfx_obj_p y) { return cmp_op(RFX_GE, x, y); }

static obj_p logic_op(int logic, obj_p *x, int64_t n) {
    rfx_host_bind();
    if (n == 0) return rfx_host_b8(0); /* logic_map: (and) -> false, core/logic.c:96-97 */
    for (int64_t i = 0; i < n; i++)
        if (!x[i] || x[i]->type != RFX_TYPE_B8 || x[i]->len != x[0]->len) return fail("and/or: expected B8 masks of one length");
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    int64_t len = x[0]->len;
    if (g_nshards > 1) {
        logic_arg_t A;
        A.logic = logic;
        A.nin = n;
        if (n > 16) { g_refused_sharded = 1; return refusedn(logic == RFX_AND ? F_AND : F_OR, x, n); }
        int rcs = RFX_OK;
        for (int64_t i = 0; i < n && rcs == RFX_OK; i++) rcs = len ? transient_sharded(x[i], &A.d_in[i]) : RFX_OK;
        obj_p outs = H.vector(RFX_TYPE_B8, len);
        if (rcs == RFX_OK && len) rcs = map_shards(outs, 1, logic_piece, &A);
        qtmp_release();
        if (rcs != RFX_OK) { H.drop(outs); return fail_hip("mask_logic"); }
        return outs;
    }
    void *acc = NULL, *nxt = NULL;
    if (rfx_hip_malloc(g_ctx, &acc, (size_t)len + 8) != RFX_OK || rfx_hip_malloc(g_ctx, &nxt, (size_t)len + 8) != RFX_OK) return fail_hip("mask");
    int ok = rfx_hip_h2d(g_ctx, acc, RFX_AS_RAW(x[0]), (size_t)len) == RFX_OK;
    for (int64_t i = 1; i < n && ok; i++)
        ok = rfx_hip_h2d(g_ctx, nxt, RFX_AS_RAW(x[i]), (size_t)len) == RFX_OK && rfx_hip_mask_logic(g_ctx, logic, (int8_t *)acc, (const int8_t *)nxt, 0, len) == RFX_OK;
    obj_p out = H.vector(RFX_TYPE_B8, len);
    ok = ok && rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), acc, (size_t)len) == RFX_OK;
    rfx_hip_free(g_ctx, acc);
    rfx_hip_free(g_ctx, nxt);
    if (!ok) { H.drop(out); return fail_hip("mask_logic"); }
    return out;
}
rfx_obj_p rfx_and(rfx_obj_p *x, int64_t n) { return logic_op(RFX_AND, x, n); }
rfx_obj_p rfx_or(rfx_obj_p *x, int64_t n) { return logic_op(RFX_OR, x, n); }

/* ---- `and` / `or` as the SPECIAL FORMS the reference registers (FN_SPECIAL_FORM, core/env.c:224-225): the arms arrive UNEVALUATED and
 * logic_map evaluates them itself (core/logic.c:89-260).  rfx_and_sf / rfx_or_sf take the same (obj_p *arms, n): when every arm is a
 * comparison -- or a nested and / or of comparisons -- over i64 / f64 vectors (a symbol the host's eval resolves, or the vector object
 * itself) and atoms, the whole tree becomes one B8 mask on the device (K2 masks + rfx_hip_mask_logic, no host round trip between the arms);
 * arms that are already B8 masks take rfx_and / rfx_or; anything else is the host's own ray_and / ray_or. ---- */
#define SF_MAX_COLS 16
typedef struct {
    int n;
    obj_p src[SF_MAX_COLS];  /* the operand as written: a symbol atom or a vector object */
    obj_p val[SF_MAX_COLS];  /* what it evaluates to (owned) */
    int64_t name[SF_MAX_COLS];
} sf_cols_t;
/* a copy of `e` whose vector / symbol operands are replaced by synthetic column symbols (collected in c); NULL: shape not covered */
static obj_p sf_rewrite(obj_p e, sf_cols_t *c, int top) {
    if (!e) return NULL;
    if (e->type == RFX_TYPE_LIST) {
        if (e->len != 3 && !(e->len >= 2 && (fn_id(RFX_AS_LIST(e)[0]) == F_AND || fn_id(RFX_AS_LIST(e)[0]) == F_OR))) return NULL;
        const int f = fn_id(RFX_AS_LIST(e)[0]);
        if (f < 0 || (top && !((f >= F_EQ && f <= F_GE) || f == F_AND || f == F_OR))) return NULL;
        obj_p out = H.vector(RFX_TYPE_LIST, e->len);
        RFX_AS_LIST(out)[0] = H.clone(RFX_AS_LIST(e)[0]);
        for (int64_t i = 1; i < e->len; i++) {
            const int sub_top = (f == F_AND || f == F_OR); /* arms of and / or must be boolean trees again; operands of a comparison may be arithmetic */
            obj_p r = sf_rewrite(RFX_AS_LIST(e)[i], c, sub_top);
            if (!r) {
                for (int64_t j = i; j < e->len; j++) RFX_AS_LIST(out)[j] = H.null_obj ? H.null_obj : rfx_host_null();
                H.drop(out);
                return NULL;
            }
            RFX_AS_LIST(out)[i] = r;
        }
        return out;
    }
    if (top) return NULL; /* an arm that is not a call */
    if (e->type == -RFX_TYPE_I64 || e->type == -RFX_TYPE_F64) return H.clone(e);
    if (e->type == -RFX_TYPE_SYMBOL || (e->type > 0 && col_ctype(e) && e->type != RFX_TYPE_SYMBOL)) {
        int k = 0;
        for (; k < c->n; k++)
            if (c->src[k] == e || (e->type == -RFX_TYPE_SYMBOL && c->src[k]->type == -RFX_TYPE_SYMBOL && c->src[k]->i64 == e->i64)) break;
        if (k == c->n) {
            if (c->n >= SF_MAX_COLS) return NULL;
            obj_p v = H.eval(e); /* a symbol: the host's binding; a vector: itself */
            if (!v || v->type <= 0 || !col_ctype(v) || v->type == RFX_TYPE_SYMBOL || (c->n > 0 && v->len != c->val[0]->len)) {
                if (v) H.drop(v);
                return NULL;
            }
            char nm[16];
            snprintf(nm, sizeof(nm), "rfxsf%d", c->n);
            c->src[c->n] = e;
            c->val[c->n] = v;
            c->name[c->n] = H.intern(nm, (int64_t)strlen(nm));
            c->n++;
        }
        obj_p sym = H.i64(c->name[k]);
        sym->type = -RFX_TYPE_SYMBOL;
        return sym;
    }
    return NULL;
}
static obj_p sf_logic_impl(int f, obj_p *x, int64_t n) {
    rfx_host_bind();
    if (n == 0) return rfx_host_b8(0); /* logic_map: (and) -> false, core/logic.c:96-97 */
    int all_masks = 1;
    for (int64_t i = 0; i < n; i++) all_masks = all_masks && x[i] && x[i]->type == RFX_TYPE_B8;
    if (all_masks) return logic_op(f == F_AND ? RFX_AND : RFX_OR, x, n); /* bound through a loader that evaluates the arguments first */
    const char *why = "an arm is not a comparison tree over i64 / f64 vectors";
    sf_cols_t c;
    memset(&c, 0, sizeof(c));
    obj_p tree = H.vector(RFX_TYPE_LIST, n + 1), tab = NULL, res = NULL;
    obj_p fo = H.i64((int64_t)(intptr_t)OUR_FN[f]);
    fo->type = RFX_TYPE_VARY;
    RFX_AS_LIST(tree)[0] = fo;
    int ok = 1;
    for (int64_t i = 0; i < n; i++) {
        obj_p r = ok ? sf_rewrite(x[i], &c, 1) : NULL;
        if (!r) ok = 0;
        RFX_AS_LIST(tree)[1 + i] = r ? r : (H.null_obj ? H.null_obj : rfx_host_null());
    }
    if (ok && c.n == 0) { ok = 0; why = "no vector operand"; }
    if (ok && ensure_ctx() != RFX_OK) {
        res = fail_hip("no usable MI355X");
        ok = 0;
    }
    if (ok) {
        obj_p names = H.vector(RFX_TYPE_SYMBOL, c.n), cols = H.vector(RFX_TYPE_LIST, c.n);
        for (int k = 0; k < c.n; k++) {
            RFX_AS_I64(names)[k] = c.name[k];
            RFX_AS_LIST(cols)[k] = c.val[k];
            c.val[k] = NULL; /* the table owns it now */
        }
        tab = H.table(names, cols);
        const int64_t nrows = RFX_AS_LIST(RFX_AS_LIST(tab)[1])[0]->len;
        int8_t *mask = NULL;
        if (g_nshards > 1) { /* every shard evaluates the tree over its rows (mask_tree_sharded), the pieces of the mask at their offsets */
            int8_t *ms[RFX_MAX_SHARDS];
            const int rcs = mask_tree_sharded(tab, tree, nrows, ms);
            if (rcs == 0) {
                res = H.vector(RFX_TYPE_B8, nrows);
                int good = 1;
                for (int sh = 0; sh < g_nshards; sh++) {
                    int64_t r0, len;
                    rfx_exec_split(nrows, g_nshards, sh, &r0, &len);
                    rfx_hip_ctx_bind_thread(g_ctxs[sh]);
                    if (good && len > 0 && rfx_hip_d2h(g_ctxs[sh], (char *)RFX_AS_RAW(res) + r0, ms[sh], (size_t)len) != RFX_OK) good = 0;
                    if (ms[sh]) rfx_hip_free(g_ctxs[sh], ms[sh]);
                }
                rfx_hip_ctx_bind_thread(g_ctx);
                if (!good) { H.drop(res); res = fail_hip("mask read-back"); }
            } else if (rcs == -2) res = fail_hip("and/or: device");
            else ok = 0;
            qtmp_release();
        } else {
        const int rc = mask_of_expr(tab, tree, nrows, &mask);
        if (rc == 0) {
            res = H.vector(RFX_TYPE_B8, nrows);
            if (nrows && rfx_hip_d2h(g_ctx, RFX_AS_RAW(res), mask, (size_t)nrows) != RFX_OK) {
                H.drop(res);
                res = fail_hip("mask read-back");
            }
            rfx_hip_free(g_ctx, mask);
        } else if (rc == -2) res = fail_hip("and/or: device");
        else ok = 0;
        qtmp_release();
        }
    }
    for (int k = 0; k < c.n; k++)
        if (c.val[k]) H.drop(c.val[k]);
    H.drop(tree);
    if (tab) H.drop(tab);
    if (res) return res;
    /* not covered: the host's own special form evaluates the arms */
    if (H.bound == 1 && H.f[f]) return HOST_CALL(((rfx_vary_f)H.f[f])(x, n));
    char b[256];
    snprintf(b, sizeof(b), "and/or (special form): not covered by the MI355X path (%s) and no host function to delegate to", why);
    return fail(b);
}
static obj_p sf_logic(int f, obj_p *x, int64_t n) {
    op_begin();
    obj_p r = sf_logic_impl(f, x, n);
    op_end();
    return r;
}
rfx_obj_p rfx_and_sf(rfx_obj_p *x, int64_t n) { return sf_logic(F_AND, x, n); }
rfx_obj_p rfx_or_sf(rfx_obj_p *x, int64_t n) { return sf_logic(F_OR, x, n); }

/* ---- the operators beside rfx_select over the shards (round 5): what the reference parallelises over its pool for every FN_AGGR
 * built-in (aggr_map core/aggr.c:375, unop_fold core/math.c:2176-2231), planned through rfx_exec on every shard ---- */
/* a per-call device copy of a host vector, every shard its row range (rfx_exec_split) -- a column of the query like any other */
static int transient_sharded(obj_p v, const void **dev) {
    if (g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) return RFX_ELIMIT;
    const size_t esz = v->type == RFX_TYPE_B8 ? 1 : 8;
    void *devs[RFX_MAX_SHARDS];
    int rc = shards_alloc(devs, v->len, esz, 0);
    if (rc != RFX_OK) return rc;
    memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
    for (int s = 0; s < g_nshards; s++) g_qtmp[g_nqtmp].d[s] = devs[s];
    g_nqtmp++; /* (released by qtmp_release, shard by shard) */
    rc = payload_upload(v->type == RFX_TYPE_B8 ? RFX_TYPE_B8 : RFX_TYPE_I64, devs, RFX_AS_RAW(v), v->len, 0);
    g_stat[ST_UPLOADS]++;
    if (rc != RFX_OK) return rc;
    *dev = devs[0];
    return qcol_add(devs);
}
/* the ids of a lazy MAPFILTER pair cut at the shards' row boundaries: piece s = the ids inside shard s's rows of an nrows-row column.  Filter ids
 * ascend (ops_where, core/ops.c:254-273), so the pieces are sub-ranges of the vector, found by binary search; every piece is then PROVEN to lie
 * inside its shard's rows on the device (min / max of the piece) -- ids in any other order answer 1 and the caller hands the pair to the host */
static int sel_ids_sharded(obj_p ids, int64_t nrows, const int64_t **d_ids, int64_t *cnt) {
    const int64_t *p = RFX_AS_I64(ids), n = ids->len;
    int64_t cut[RFX_MAX_SHARDS + 1];
    cut[0] = 0;
    for (int s = 1; s < g_nshards; s++) {
        int64_t r0, lo = cut[s - 1], hi = n;
        rfx_exec_split(nrows, g_nshards, s, &r0, NULL);
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (p[mid] < r0) lo = mid + 1;
            else hi = mid;
        }
        cut[s] = lo;
    }
    cut[g_nshards] = n;
    if (g_nqtmp >= (int)(sizeof(g_qtmp) / sizeof(g_qtmp[0]))) return RFX_ELIMIT;
    memset(&g_qtmp[g_nqtmp], 0, sizeof(g_qtmp[0]));
    void **devs = g_qtmp[g_nqtmp++].d;
    int rc = RFX_OK, outside = 0;
    for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
        int64_t r0, len;
        rfx_exec_split(nrows, g_nshards, s, &r0, &len);
        cnt[s] = cut[s + 1] - cut[s];
        rfx_hip_ctx_bind_thread(g_ctxs[s]);
        rc = rfx_hip_malloc(g_ctxs[s], &devs[s], (size_t)(cnt[s] ? cnt[s] : 1) * 8);
        if (rc == RFX_OK && cnt[s]) rc = rfx_hip_h2d_pipelined(g_ctxs[s], devs[s], p + cut[s], (size_t)cnt[s] * 8);
        if (rc == RFX_OK && cnt[s]) {
            int64_t mn = 0, mx = -1, seen = 0;
            rc = rfx_hip_scope_i64(g_ctxs[s], (const int64_t *)devs[s], NULL, 0, RFX_AND, cnt[s], &mn, &mx, &seen);
            if (rc == RFX_OK && (mn < r0 || mx >= r0 + len)) outside = 1;
        }
        d_ids[s] = (const int64_t *)devs[s];
    }
    rfx_hip_ctx_bind_thread(g_ctx);
    g_stat[ST_UPLOADS]++;
    return rc != RFX_OK ? rc : (outside ? 1 : RFX_OK);
}
/* one aggregate of a column over every shard (the whole column, or its rows at per-shard ids) */
static int fold_sharded(const rfx_agg_t *a, int64_t nrows, const int64_t *const *d_ids, const int64_t *cnt, rfx_value_t *v) {
    rfx_query_t Q;
    memset(&Q, 0, sizeof(Q));
    Q.aggs = a;
    Q.nagg = 1;
    Q.logic = RFX_AND;
    Q.nrows = nrows;
    Q.cols = g_qcols;
    Q.ncols = g_nqcols;
    Q.d_sel_ids = d_ids;
    Q.sel_count = cnt;
    return rfx_exec_filter_aggr(g_x, &Q, v, NULL);
}

static obj_p where_impl(obj_p mask) {
    rfx_host_bind();
    if (!mask || mask->type != RFX_TYPE_B8) return fail("where: expected a B8 mask"); /* err_type, core/items.c:1395 */
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    if (g_nshards > 1) { /* every shard turns ITS rows of the mask into ids (global: its row offset added), the runs concatenated in shard order */
        const void *dms = NULL;
        rfx_query_t Q;
        rfx_ids_t ids;
        memset(&Q, 0, sizeof(Q));
        int rc = mask->len ? transient_sharded(mask, &dms) : RFX_OK;
        Q.d_mask = (const int8_t *)dms;
        Q.logic = RFX_AND;
        Q.nrows = mask->len;
        Q.cols = g_qcols;
        Q.ncols = g_nqcols;
        if (rc == RFX_OK && mask->len == 0) { qtmp_release(); return H.vector(RFX_TYPE_I64, 0); }
        if (rc == RFX_OK) rc = rfx_exec_where(g_x, &Q, &ids);
        if (rc != RFX_OK) { qtmp_release(); return fail(rc == RFX_OK ? "where" : (rfx_exec_last_error(g_x)[0] ? rfx_exec_last_error(g_x) : rfx_hip_last_error())); }
        obj_p outv = H.vector(RFX_TYPE_I64, ids.total);
        int64_t at = 0;
        int ok2 = 1;
        for (int sh = 0; sh < ids.nshards && ok2; sh++) {
            if (!ids.count[sh]) continue;
            rfx_hip_ctx_bind_thread(g_ctxs[sh]);
            ok2 = rfx_hip_d2h(g_ctxs[sh], (char *)RFX_AS_RAW(outv) + (size_t)at * 8, ids.d_ids[sh], (size_t)ids.count[sh] * 8) == RFX_OK;
            at += ids.count[sh];
        }
        rfx_hip_ctx_bind_thread(g_ctx);
        rfx_exec_ids_free(g_x, &ids);
        qtmp_release();
        if (!ok2) { H.drop(outv); return fail_hip("where"); }
        return outv;
    }
    const void *dm;
    if (transient(mask, &dm) != RFX_OK) return fail_hip("mask upload"); /* a mask is a temporary: per-call scratch, never cached */
    int64_t count = 0;
    if (rfx_hip_where_begin(g_ctx, NULL, 0, RFX_AND, (const int8_t *)dm, mask->len, &count) != RFX_OK) return fail_hip("where");
    obj_p out = H.vector(RFX_TYPE_I64, count);
    void *di = NULL;
    int ok = 1;
    if (count > 0) {
        ok = rfx_hip_malloc(g_ctx, &di, (size_t)count * 8) == RFX_OK && rfx_hip_where_emit(g_ctx, 0, (int64_t *)di) == RFX_OK &&
             rfx_hip_d2h(g_ctx, RFX_AS_RAW(out), di, (size_t)count * 8) == RFX_OK;
        if (di) rfx_hip_free(g_ctx, di);
    }
    if (!ok) { H.drop(out); return fail_hip("where"); }
    return out;
}

rfx_obj_p rfx_where(rfx_obj_p mask) {
    op_begin();
    obj_p r = where_impl(mask);
    op_end();
    return r;
}
