/* rfx_ops_repro.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * Reproducible grouped f64 sums, the residency side: the scale of a column's fixed-point image, the ranks' agreement on it, and the images of resident columns
 * (made once, cached under the column's owner).  The query side -- which aggregates are rewritten, the hidden aggregates, the way back to f64 -- is
 * rfx_ops_select.c: det_rewrite / sel_build_groups. */
/* Reproducible grouped f64 sums (rfx_ops_select.c: det_rewrite): the scale of a column's fixed-point image.  2^e > max |x|, 2^b >= rows: no sum of `rows`
 * cells llrint(x * 2^k), k = 62 - e - b, leaves 63 bits -- and k depends on the column and the table's length alone, not on how the rows are sharded. */
static int det_scale(double amax, int64_t nrows, int *m) {
    int e = 0, b = 0;
    if (amax > 0.0) (void)frexp(amax, &e); /* amax = f * 2^e, 0.5 <= f < 1 */
    while (b < 62 && ((int64_t)1 << b) < nrows) b++;
    if (m) *m = 62 - b; /* the second limb: |cell| <= 2^(m - 1), `rows` of them stay inside 63 bits too */
    return 62 - e - b;
}
/* One process per device (rfx_ops_dist_init: every rank holds a row range of the table and the planner merges the ranks' tables): the scale must be the SAME
 * on every rank -- the ranks' integer sums are added to each other -- so max |x|, the row count and the "holds a NaN / an infinity" flag are taken over all
 * ranks' ranges (through the planner's own inter-process side: the host's transport or the RCCL communicator).  Exactly ONE small all-gather per rewritten aggregate on every rank, whichever path (cached image / scratch) a rank takes; nothing in one
 * process, however many devices it drives (its shards are looked at together). */
static int det_ranks(void) { return g_x ? rfx_exec_ranks(g_x, NULL) : 1; } /* ranks of a one-process-per-device world this process is one of; 1 otherwise */
static int det_world_agree(double *amax, int64_t *nrows, int *bad) {
    const int world = det_ranks();
    if (world <= 1) return RFX_OK;
    int rc;
    typedef struct { double a; int64_t n, b; } agree_t;
    agree_t mine = {*amax, *nrows, *bad}, *all = (agree_t *)malloc(sizeof(agree_t) * (size_t)world);
    if (!all) return RFX_ENOMEM;
    rc = rfx_exec_allgather_host(g_x, &mine, sizeof(mine), all);
    if (rc == RFX_OK) {
        double a = 0.0;
        int64_t n = 0, b = 0;
        for (int r = 0; r < world; r++) { a = all[r].a > a ? all[r].a : a; n += all[r].n; b |= all[r].b; }
        *amax = a; *nrows = n; *bad = b != 0;
    }
    free(all);
    return rc;
}
/* ... and the image itself for a column the cache holds BY OWNERSHIP: made once (max |x| remembered with the copy, the image a cache entry of its own --
 * type code + 256, the same owner: immutable for as long as it lives, released with the owner, evicted like any unpinned copy), so that a repeated query in
 * the reproducible mode runs at the default path's speed; the price is a second 8 bytes per row of HBM for the f64 columns such queries sum.  1: not a
 * column this applies to (a device vector, checksum mode) -- det_rewrite's per-query scratch path decides; 2: a NaN / an infinity inside (on some rank): this
 * aggregate keeps the default path. */
static int resident_fixed(const void *base_dev, int64_t nrows, int limbs, int *k_out, int *m_out, const void **dev_out /* [limbs] */) {
    resident_t *re = resident_entry(base_dev);
    if (!re || !re->owner || re->type != RFX_TYPE_F64 || re->len != nrows || nrows <= 0) return 1;
    int rc = RFX_OK;
    if (!re->amax_ok) {
        double mx = 0.0;
        int bad = 0;
        for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
            int64_t n;
            rfx_exec_split(nrows, g_nshards, s, NULL, &n);
            if (g_nshards == 1) n = nrows;
            if (n <= 0) continue;
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
            double m1 = 0.0;
            int b1 = 0;
            rc = rfx_hip_absmax_f64(g_ctxs[s], (const double *)re->devs[s], n, &m1, &b1);
            mx = m1 > mx ? m1 : mx;
            bad |= b1;
        }
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
        if (rc != RFX_OK) return rc;
        re->amax = mx;
        re->amax_ok = bad ? 2 : 1;
    }
    double amax = re->amax;
    int64_t world_rows = nrows;
    int bad = re->amax_ok != 1;
    rc = det_world_agree(&amax, &world_rows, &bad);
    if (rc != RFX_OK) return rc;
    int m = 0;
    const int k = det_scale(amax, world_rows, &m);
    if (bad || k <= -1000 || k >= 1000) return 2;
    *k_out = k;
    *m_out = m;
    for (int limb = 0; limb < limbs; limb++) {
        const int itype = RFX_TYPE_F64 + 256 * (limb + 1);
        int found = 0;
        re = resident_entry(base_dev);
        if (!re) return 1;
        for (int i = 0; i < g_nres && !found; i++) {
            if (g_res[i].owner != re->owner || g_res[i].type != itype) continue;
            if (g_res[i].len != nrows || g_res[i].fix_k != k || g_res[i].fix_m != m) { res_free(i); break; } /* (one process: cannot happen, the owner's cells do not change; ranks: another rank's did) */
            g_res[i].tick = ++g_tick;
            g_res[i].epoch = g_epoch;
            g_fix_hits++;
            dev_out[limb] = g_res[i].dev;
            rc = qcol_add(g_res[i].devs);
            if (rc != RFX_OK) return rc;
            found = 1;
        }
        if (found) continue;
        re = resident_entry(base_dev);
        if (!re) return 1;
        const resident_t base = *re; /* (the table of entries may move below) */
        res_make_room((size_t)nrows * 8);
        void *devs[RFX_MAX_SHARDS];
        rc = shards_alloc(devs, nrows, 8, 0);
        if (rc != RFX_OK) return rc;
        for (int s = 0; s < g_nshards && rc == RFX_OK; s++) {
            int64_t n;
            rfx_exec_split(nrows, g_nshards, s, NULL, &n);
            if (g_nshards == 1) n = nrows;
            if (n <= 0) continue;
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]);
            rc = limb ? rfx_hip_fix_f64_low(g_ctxs[s], (const double *)base.devs[s], n, k, m, (int64_t *)devs[s])
                      : rfx_hip_fix_f64(g_ctxs[s], (const double *)base.devs[s], n, k, (int64_t *)devs[s]);
        }
        if (rc != RFX_OK) {
            for (int s = 0; s < g_nshards; s++) { if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctxs[s]); rfx_hip_free(g_ctxs[s], devs[s]); }
            if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
            return rc;
        }
        if (g_nshards > 1) rfx_hip_ctx_bind_thread(g_ctx);
        resident_t e;
        memset(&e, 0, sizeof(e));
        e.host = base.host; e.len = nrows; e.type = itype; e.dev = devs[0]; e.bytes = base.bytes; e.tick = ++g_tick; e.epoch = g_epoch; e.dbytes = (size_t)nrows * 8;
        e.owner = H.clone(base.owner);
        e.fix_k = k;
        e.fix_m = m;
        for (int s = 0; s < g_nshards; s++) e.devs[s] = devs[s];
        res_append(&e);
        g_fix_built++;
        dev_out[limb] = devs[0];
        rc = qcol_add(devs);
        if (rc != RFX_OK) return rc;
    }
    return RFX_OK;
}
