/* Pure C host over the flat device ABI (include/rfx_hip.h): no Python, no torch.  Built and run by tests/test_c_host_gpu.py.
 * select sum(a), count(a), max(v * a) from t where a < 100000 and v > 0.25   -- and the same grouped by k. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rfx_hip.h"

#define CHECK(x)                                                                      \
    do {                                                                              \
        int _rc = (x);                                                                \
        if (_rc != RFX_OK) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #x, _rc, rfx_hip_last_error());         \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(void) {
    const int64_t n = 1000003;
    rfx_ctx_t *ctx;
    CHECK(rfx_hip_ctx_create(0, NULL, &ctx));
    int64_t *d_k, *d_a;
    double *d_v;
    CHECK(rfx_hip_malloc(ctx, (void **)&d_k, n * 8));
    CHECK(rfx_hip_malloc(ctx, (void **)&d_a, n * 8));
    CHECK(rfx_hip_malloc(ctx, (void **)&d_v, n * 8));
    CHECK(rfx_hip_gen_i64(ctx, d_k, n, 4, 0, 1000));
    CHECK(rfx_hip_gen_i64(ctx, d_a, n, 2, 0, 1000000));
    CHECK(rfx_hip_gen_f64(ctx, d_v, n, 5, 0));
    int64_t *k = malloc(n * 8), *a = malloc(n * 8);
    double *v = malloc(n * 8);
    CHECK(rfx_hip_d2h(ctx, k, d_k, n * 8));
    CHECK(rfx_hip_d2h(ctx, a, d_a, n * 8));
    CHECK(rfx_hip_d2h(ctx, v, d_v, n * 8));

    rfx_pred_t p[2];
    memset(p, 0, sizeof(p));
    p[0].d_col = d_a; p[0].col_type = RFX_I64; p[0].rhs_type = RFX_I64; p[0].op = RFX_LT; p[0].rhs_i = 100000;
    p[1].d_col = d_v; p[1].col_type = RFX_F64; p[1].rhs_type = RFX_F64; p[1].op = RFX_GT; p[1].rhs_f = 0.25;
    rfx_agg_t g[3];
    memset(g, 0, sizeof(g));
    g[0].d_col = d_a; g[0].col_type = RFX_I64; g[0].kind = RFX_AGG_SUM;
    g[1].d_col = d_a; g[1].col_type = RFX_I64; g[1].kind = RFX_AGG_COUNT;
    g[2].d_col = d_v; g[2].col_type = RFX_F64; g[2].kind = RFX_AGG_MAX; g[2].xop = RFX_X_MUL; g[2].d_xrhs_col = d_a; g[2].xrhs_type = RFX_I64;
    rfx_value_t out[3];
    int64_t selected = 0;
    CHECK(rfx_hip_filter_aggr_host(ctx, p, 2, RFX_AND, g, 3, n, out, &selected));
    int64_t s = 0, c = 0;
    double mx = -INFINITY;
    for (int64_t i = 0; i < n; i++)
        if (a[i] < 100000 && v[i] > 0.25) { s += a[i]; c++; if (v[i] * (double)a[i] > mx) mx = v[i] * (double)a[i]; }
    if (out[0].i != s || out[1].i != c || selected != c || out[2].f != mx) {
        fprintf(stderr, "scalar mismatch: %lld/%lld %lld/%lld %.17g/%.17g\n", (long long)out[0].i, (long long)s, (long long)out[1].i, (long long)c, out[2].f, mx);
        return 2;
    }

    /* grouped: dense tables over [kmin, kmax], ranked by first occurrence */
    int64_t kmin, kmax, seen;
    CHECK(rfx_hip_scope_i64(ctx, d_k, p, 2, RFX_AND, n, &kmin, &kmax, &seen));
    const int64_t range = kmax - kmin + 1;
    int narr = 0;
    CHECK(rfx_hip_group_table_arrays(g, 2, &narr));
    int64_t *store;
    CHECK(rfx_hip_malloc(ctx, (void **)&store, (size_t)narr * range * 8));
    rfx_group_tables_t t;
    memset(&t, 0, sizeof(t));
    t.kmin = kmin; t.range = range; t.nagg = 2;
    t.d_first = store; t.d_acc[0] = store + range; t.d_cnt[0] = store + 2 * range; t.d_acc[1] = store + 3 * range;
    CHECK(rfx_hip_group_tables_init(ctx, g, &t));
    CHECK(rfx_hip_group_dense_accumulate(ctx, d_k, p, 2, RFX_AND, g, n, 0, &t));
    int64_t groups = 0;
    CHECK(rfx_hip_group_rank(ctx, &t, n, &groups));
    int64_t *d_out;
    CHECK(rfx_hip_malloc(ctx, (void **)&d_out, (size_t)3 * groups * 8));
    void *res[2] = {d_out + groups, d_out + 2 * groups};
    CHECK(rfx_hip_group_emit(ctx, g, &t, d_out, NULL, res));
    int64_t *h = malloc((size_t)3 * groups * 8);
    CHECK(rfx_hip_d2h(ctx, h, d_out, (size_t)3 * groups * 8));
    /* host check: first-occurrence order and per-group sum / count */
    int64_t *hs = calloc(1000, 8), *hc = calloc(1000, 8), *order = malloc(1000 * 8), ng = 0;
    char *seen_k = calloc(1000, 1);
    for (int64_t i = 0; i < n; i++)
        if (a[i] < 100000 && v[i] > 0.25) {
            if (!seen_k[k[i]]) { seen_k[k[i]] = 1; order[ng++] = k[i]; }
            hs[k[i]] += a[i];
            hc[k[i]]++;
        }
    if (ng != groups) { fprintf(stderr, "group count %lld vs %lld\n", (long long)groups, (long long)ng); return 3; }
    for (int64_t j = 0; j < groups; j++)
        if (h[j] != order[j] || h[groups + j] != hs[order[j]] || h[2 * groups + j] != hc[order[j]]) {
            fprintf(stderr, "group %lld mismatch\n", (long long)j);
            return 4;
        }
    printf("c host ok: %lld rows selected, %lld groups\n", (long long)c, (long long)groups);
    rfx_hip_ctx_destroy(ctx);
    return 0;
}
