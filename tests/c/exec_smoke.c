/* Pure C host of the PLANNER (include/rfx_exec.h): three row-range shards on one device, no Python, no torch.  Built and run by
 * tests/test_c_host_gpu.py.  select sum(a), count(a), first(v) from t where a < 300000 or (v > 0.5 and a > 900000) by k -- and the
 * scalar form -- answered from all shards and checked on the host. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rfx_exec.h"

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        int _rc = (x);                                                                                   \
        if (_rc != RFX_OK) {                                                                             \
            fprintf(stderr, "%s -> %d: %s / %s\n", #x, _rc, rfx_hip_last_error(), rfx_exec_last_error(X)); \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)
#define S 3

int main(void) {
    const int64_t n = 2000011, keys = 5000;
    rfx_ctx_t *ctx[S];
    rfx_exec_t *X = NULL;
    for (int s = 0; s < S; s++) CHECK(rfx_hip_ctx_create(0, NULL, &ctx[s]));
    CHECK(rfx_exec_create(ctx, S, &X));
    /* every shard generates ITS rows of the three columns (the generator is counter-based: row0 = the shard's first row) */
    rfx_qcol_t cols[3];
    memset(cols, 0, sizeof(cols));
    for (int s = 0; s < S; s++) {
        int64_t r0, len;
        rfx_exec_split(n, S, s, &r0, &len);
        void *dk, *da, *dv;
        CHECK(rfx_hip_malloc(ctx[s], &dk, (size_t)(len ? len : 1) * 8));
        CHECK(rfx_hip_malloc(ctx[s], &da, (size_t)(len ? len : 1) * 8));
        CHECK(rfx_hip_malloc(ctx[s], &dv, (size_t)(len ? len : 1) * 8));
        CHECK(rfx_hip_gen_i64(ctx[s], (int64_t *)dk, len, 4, r0, (uint64_t)keys));
        CHECK(rfx_hip_gen_i64(ctx[s], (int64_t *)da, len, 2, r0, 1000000));
        CHECK(rfx_hip_gen_f64(ctx[s], (double *)dv, len, 5, r0));
        CHECK(rfx_hip_ctx_sync(ctx[s]));
        cols[0].d[s] = dk; cols[1].d[s] = da; cols[2].d[s] = dv;
    }
    /* the same columns on the host (whole), for the check */
    int64_t *k = malloc(n * 8), *a = malloc(n * 8);
    double *v = malloc(n * 8);
    for (int s = 0; s < S; s++) {
        int64_t r0, len;
        rfx_exec_split(n, S, s, &r0, &len);
        CHECK(rfx_hip_d2h(ctx[s], k + r0, cols[0].d[s], (size_t)len * 8));
        CHECK(rfx_hip_d2h(ctx[s], a + r0, cols[1].d[s], (size_t)len * 8));
        CHECK(rfx_hip_d2h(ctx[s], v + r0, cols[2].d[s], (size_t)len * 8));
    }
    /* where: (or (< a 300000) (and (> v 0.5) (> a 900000))) -- a two-level tree: the parenthesis' comparisons carry `more` */
    rfx_pred_t p[3];
    memset(p, 0, sizeof(p));
    p[0].d_col = cols[1].d[0]; p[0].col_type = RFX_I64; p[0].rhs_type = RFX_I64; p[0].op = RFX_LT; p[0].rhs_i = 300000;
    p[1].d_col = cols[2].d[0]; p[1].col_type = RFX_F64; p[1].rhs_type = RFX_F64; p[1].op = RFX_GT; p[1].rhs_f = 0.5; p[1].more = 1;
    p[2].d_col = cols[1].d[0]; p[2].col_type = RFX_I64; p[2].rhs_type = RFX_I64; p[2].op = RFX_GT; p[2].rhs_i = 900000;
    rfx_agg_t g[3];
    memset(g, 0, sizeof(g));
    g[0].d_col = cols[1].d[0]; g[0].col_type = RFX_I64; g[0].kind = RFX_AGG_SUM;
    g[1].d_col = cols[1].d[0]; g[1].col_type = RFX_I64; g[1].kind = RFX_AGG_COUNT;
    g[2].d_col = cols[2].d[0]; g[2].col_type = RFX_F64; g[2].kind = RFX_AGG_FIRST;
    const void *dkeys[1] = {cols[0].d[0]};
    rfx_query_t q;
    memset(&q, 0, sizeof(q));
    q.preds = p; q.npred = 3; q.logic = RFX_OR;
    q.aggs = g; q.nagg = 3;
    q.nrows = n;
    q.cols = cols; q.ncols = 3;
#define SEL(i) (a[i] < 300000 || (v[i] > 0.5 && a[i] > 900000))
    /* scalar */
    rfx_value_t val[3];
    int64_t selected = 0;
    CHECK(rfx_exec_filter_aggr(X, &q, val, &selected));
    int64_t s0 = 0, c0 = 0, f0 = -1;
    for (int64_t i = 0; i < n; i++)
        if (SEL(i)) { s0 += a[i]; c0++; if (f0 < 0) f0 = i; }
    if (val[0].i != s0 || val[1].i != c0 || selected != c0 || val[2].f != v[f0]) { fprintf(stderr, "scalar mismatch\n"); return 2; }
    /* grouped */
    q.nkeys = 1; q.d_keys = dkeys; q.flags = RFX_Q_WANT_FIRST;
    rfx_groups_t R;
    CHECK(rfx_exec_group_by(X, &q, &R));
    int64_t *hs = calloc(keys, 8), *hc = calloc(keys, 8), *hf = malloc(keys * 8), *order = malloc(keys * 8), ng = 0;
    for (int64_t j = 0; j < keys; j++) hf[j] = -1;
    for (int64_t i = 0; i < n; i++)
        if (SEL(i)) {
            if (hf[k[i]] < 0) { hf[k[i]] = i; order[ng++] = k[i]; }
            hs[k[i]] += a[i];
            hc[k[i]]++;
        }
    if (R.groups != ng) { fprintf(stderr, "group count %lld vs %lld\n", (long long)R.groups, (long long)ng); return 3; }
    int64_t *gk = malloc(ng * 8), *gf = malloc(ng * 8), *gs = malloc(ng * 8), *gc = malloc(ng * 8);
    double *gv = malloc(ng * 8);
    CHECK(rfx_exec_groups_fetch(X, &R, gk, R.d_keys, (size_t)ng * 8));
    CHECK(rfx_exec_groups_fetch(X, &R, gf, R.d_first, (size_t)ng * 8));
    CHECK(rfx_exec_groups_fetch(X, &R, gs, R.d_results[0], (size_t)ng * 8));
    CHECK(rfx_exec_groups_fetch(X, &R, gc, R.d_results[1], (size_t)ng * 8));
    CHECK(rfx_exec_groups_fetch(X, &R, gv, R.d_results[2], (size_t)ng * 8));
    for (int64_t j = 0; j < ng; j++)
        if (gk[j] != order[j] || gf[j] != hf[order[j]] || gs[j] != hs[order[j]] || gc[j] != hc[order[j]] || gv[j] != v[hf[order[j]]]) {
            fprintf(stderr, "group %lld mismatch\n", (long long)j);
            return 4;
        }
    rfx_exec_groups_free(X, &R);
    if (rfx_exec_stat(X, RFX_XSTAT_MERGES_KERNEL) < 2) { fprintf(stderr, "the shards' tables were not merged by the kernel\n"); return 5; }
    printf("c planner host ok: %d shards, %lld rows selected, %lld groups\n", S, (long long)c0, (long long)ng);
    rfx_exec_destroy(X);
    for (int s = 0; s < S; s++) rfx_hip_ctx_destroy(ctx[s]);
    return 0;
}
