/* rfx_exec_merge.c -- part of the planner's ONE translation unit (rfx_exec.c #includes it -- the Makefile does not compile it on its own; the pieces share struct rfx_exec
 * and file-static helpers).  group-by: the partial tables of the shards become one (kernel / fused RCCL exchange / transport), rank + emit per slice owner, FIRST values. */
/* every shard's partial tables -> the merged tables, on every shard that goes on to rank / emit.  *full: a hashed merge ran out of room */
static int merge_tables(gq_t *G, int *full) {
    rfx_exec_t *x = G->x;
    int rc = RFX_OK;
    *full = 0;
    if (G->S > x->ndev) { /* shards sharing a device */
        rc = run_shards(x, ph_merge_local, G);
        if (rc != RFX_OK) return rc;
        for (int s = 0; s < G->S; s++) *full |= (x->lead[s] == s && G->sh[s].flag);
    }
    /* a hashed re-insertion that ran out of room on ONE process must stop EVERY process before the exchange below: a process that skipped
     * it alone would meet the others' collective with the next pass' (a hang under RCCL, a size mismatch under gloo).  Asked by every
     * process alike, whatever its own shard layout; dense merges never fill up and need no such agreement. */
    if (G->exch && !G->dense) {
        int any = 0;
        if ((rc = xp_any(x, G->world, *full, &any)) != RFX_OK) return rc;
        *full = any;
    }
    if (x->ndev > 1 && !x->comm_all) {
        snprintf(x->err, sizeof(x->err), "rfx_exec: several devices without communicators (rfx_exec_comm_init_all)");
        return RFX_ESTATE;
    }
    if (x->comm_all && !*full) { /* the devices of this process: ONE fused exchange over xGMI */
        rfx_ctx_t *leads[RFX_MAX_SHARDS];
        for (int d = 0; d < x->ndev; d++) leads[d] = x->ctx[x->devlead[d]];
        if (G->dense) {
            const rfx_group_tables_t *ts[RFX_MAX_SHARDS];
            for (int d = 0; d < x->ndev; d++) ts[d] = &G->sh[x->devlead[d]].gt;
            rc = rfx_dist_group_tables_allreduce_all(leads, x->ndev, G->sh[0].aggs, ts);
        } else {
            const int64_t cells = G->cap + 1;
            const size_t one = (size_t)G->narr * (size_t)cells * 8;
            const void *ins[RFX_MAX_SHARDS];
            void *outs[RFX_MAX_SHARDS];
            for (int d = 0; d < x->ndev && rc == RFX_OK; d++) {
                shard_t *h = &G->sh[x->devlead[d]];
                rc = rfx_hip_ctx_bind_thread(leads[d]);
                if (rc == RFX_OK) rc = rfx_hip_malloc(leads[d], &h->dout, one * (size_t)x->ndev);
                ins[d] = h->store;
                outs[d] = h->dout;
            }
            rfx_hip_ctx_bind_thread(x->ctx[0]);
            if (rc == RFX_OK) rc = rfx_dist_allgather_all(leads, x->ndev, ins, one, outs);
            if (rc == RFX_OK) rc = run_shards(x, ph_merge_gathered, G);
            for (int d = 0; d < x->ndev; d++) {
                shard_t *h = &G->sh[x->devlead[d]];
                if (h->dout) {
                    rfx_hip_ctx_bind_thread(leads[d]);
                    rfx_hip_free(leads[d], h->dout);
                    h->dout = NULL;
                }
                *full |= h->flag;
            }
            rfx_hip_ctx_bind_thread(x->ctx[0]);
        }
        if (rc != RFX_OK) {
            if (!x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
            return rc;
        }
        x->stat[RFX_XSTAT_MERGES_RCCL]++;
        rc = run_shards(x, ph_sync, G);
        if (rc != RFX_OK) return rc;
    }
    if (G->exch && !*full) { /* the other processes: the lead context's exchange */
        shard_t *h = &G->sh[0];
        if (G->dense) {
            if (x->has_tr) {
                /* a host transport reduces array by array; neighbours of one class as one call */
                struct { void *p; int64_t n; int type, op; } calls[1 + 2 * RFX_MAX_AGGS];
                int n = 0;
                const int64_t cells = (int64_t)G->range;
                calls[n].p = h->gt.d_first; calls[n].n = cells; calls[n].type = 0; calls[n].op = 1; n++;
                for (int a = 0; a < G->na; a++) {
                    const int f64 = rfx_agg_input_type(&h->aggs[a]) == RFX_F64, kind = h->aggs[a].kind;
                    int type = 0, op = 0;
                    if (kind == RFX_AGG_MIN) op = 1;
                    else if (kind == RFX_AGG_MAX) op = 2;
                    else if (kind == RFX_AGG_AVG || (kind == RFX_AGG_SUM && f64)) type = 1;
                    if (n && calls[n - 1].type == type && calls[n - 1].op == op && (char *)calls[n - 1].p + calls[n - 1].n * 8 == (char *)h->gt.d_acc[a]) calls[n - 1].n += cells;
                    else { calls[n].p = h->gt.d_acc[a]; calls[n].n = cells; calls[n].type = type; calls[n].op = op; n++; }
                    if (h->gt.d_cnt[a]) {
                        if (calls[n - 1].type == 0 && calls[n - 1].op == 0 && (char *)calls[n - 1].p + calls[n - 1].n * 8 == (char *)h->gt.d_cnt[a]) calls[n - 1].n += cells;
                        else { calls[n].p = h->gt.d_cnt[a]; calls[n].n = cells; calls[n].type = 0; calls[n].op = 0; n++; }
                    }
                }
                for (int i = 0; i < n && rc == RFX_OK; i++) rc = xp_allreduce(x, calls[i].p, calls[i].n, calls[i].type, calls[i].op);
            } else {
                x->stat[RFX_XSTAT_MERGES_TRANSPORT]++;
                rc = rfx_dist_group_tables_allreduce(x->ctx[0], h->aggs, &h->gt);
            }
            if (rc == RFX_OK) rc = rfx_hip_ctx_sync(x->ctx[0]);
        } else {
            const int64_t cells = G->cap + 1;
            const size_t one = (size_t)G->narr * (size_t)cells * 8;
            void *all = NULL;
            rc = rfx_hip_malloc(x->ctx[0], &all, one * (size_t)G->world);
            if (rc == RFX_OK) rc = xp_allgather_dev(x, h->store, one, all);
            for (int r = 0; r < G->world && rc == RFX_OK; r++) {
                if (r == G->rank) continue;
                rfx_hash_tables_t o = h->ht;
                int64_t *base = (int64_t *)((char *)all + (size_t)r * one);
                int k = 0;
                o.d_keys = base + (k++) * cells;
                o.d_first = base + (k++) * cells;
                for (int a = 0; a < G->na; a++) {
                    o.d_acc[a] = base + (k++) * cells;
                    o.d_cnt[a] = has_cnt(&h->aggs[a]) ? base + (k++) * cells : NULL;
                }
                rc = rfx_hip_hash_tables_merge(x->ctx[0], h->aggs, &h->ht, &o);
                if (rc == RFX_ELIMIT) {
                    *full = 1;
                    rc = RFX_OK;
                    break;
                }
            }
            if (rc == RFX_OK) rc = rfx_hip_ctx_sync(x->ctx[0]);
            if (all) rfx_hip_free(x->ctx[0], all);
            int any = 0;
            if (rc == RFX_OK) rc = xp_any(x, G->exch ? G->world : 0, *full, &any);
            *full = any;
        }
        if (rc != RFX_OK) {
            if (!x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
            return rc;
        }
    }
    /* FIRST values are read where the rows are: the merged tables go down from every device's lead to the shards beside it */
    if ((G->all_rank || G->slice_all) && !*full && G->S > x->ndev) rc = run_shards(x, ph_copy_back, G);
    return rc;
}

/* rank by first row (first-occurrence order, core/index.c:2037-2055) and emit: on the lead; on every shard when FIRST values are asked
 * for (a group's first value is read by the shard that owns its first row, the others write 0); on every SLICE OWNER of a sliced result
 * (every owner ranks the same merged tables -- redundant, and parallel -- and emits only its range of the groups) */
static int ph_rank_emit(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    const int si = G->slidx[s];
    if (si < 0 && !G->all_rank) return RFX_OK;
    shard_t *h = &G->sh[s];
    rfx_ctx_t *c = x->ctx[s];
    /* FIRST: a first row is owned by the shard whose rows [row0, row0 + nloc) hold it (nloc 0 = the one shard owns every row); a shard
     * without rows owns none (a row offset no first row reaches) */
    const int multi = G->S > 1 || G->exch;
    const int64_t nloc = multi ? (h->nrows > 0 ? h->nrows : 1) : 0, r0 = (multi && h->nrows == 0) ? INF_I64 : h->row0;
    const int nsl = G->nsl > 1 ? G->nsl : 1, sl = G->nsl > 1 ? si : 0;
    void *ptrs[RFX_MAX_AGGS];
    int rc;
    h->g0 = h->gn = 0;
    const int64_t slots = G->dense ? (int64_t)G->range : G->cap + 1;
    /* (the one-launch form sizes its outputs by the BOUND min(slots, selected rows), and the result keeps that block until it is freed: beyond 64 MB --
     *  8 aggregates over 4M slots would pin 300 MB for what may be a handful of groups -- the group count comes back first and the block is exact) */
    const int64_t bound0 = (slots < G->seen ? slots : G->seen) < 1 ? 1 : (slots < G->seen ? slots : G->seen);
    const int one_launch_fits = (size_t)(G->na + 1) * (size_t)(bound0 / (G->nsl > 1 ? G->nsl : 1) + 1) * 8 <= ((size_t)64 << 20);
    /* (RFX_EMIT_BY_ROWS=2: wherever the probe arrays exist, whatever the sizes -- how the small tests reach this path; a PACKED table has no other tail) */
    if (h->packed && !(h->probe_ids && h->probe_slots && nsl == 1 && !multi)) { snprintf(x->err, sizeof(x->err), "rfx_exec: a packed table without its rows' slots"); return RFX_ESTATE; }
    if (by_rows_env() && !G->dense && !multi && nsl == 1 && h->probe_ids && h->probe_slots && ((slots > RFX_RANK_EMIT_MAX && h->nrows <= 4 * slots) || by_rows_env() == 2 || h->packed)) {
        /* MANY groups in a table that is large against the rows (the row-hash route's 1e8 groups in 2.7e8 slots): the groups are the rows that head
         * their own group, in ascending order -- a compaction over the probe's first rows and one gather per table array at those rows' slots, instead
         * of five passes over the slots, a slot -> id array, an inverse permutation and a gather through it (rfx_hip_hash_rows_*) */
        rc = rfx_hip_hash_rows_begin(c, h->probe_ids, h->nrows, &h->groups);
        if (x->timing) h->t_rank = now_ns();
        if (rc != RFX_OK || h->groups == 0) return rc;
        const int64_t gn = h->groups;
        h->g0 = 0;
        h->gn = gn;
        h->gstride = gn;
        if ((rc = rfx_hip_malloc(c, &h->dout, (size_t)(G->na + 1) * (size_t)gn * 8)) != RFX_OK) return rc;
        if ((rc = rfx_hip_malloc(c, &h->dfirst, (size_t)gn * 8)) != RFX_OK) return rc;
        for (int a = 0; a < G->na; a++) ptrs[a] = (int64_t *)h->dout + (size_t)(a + 1) * (size_t)gn;
        if ((rc = rfx_hip_hash_rows_emit(c, h->aggs, &h->ht, h->probe_slots, (const int64_t *)h->key, r0, nloc, gn, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs)) != RFX_OK) return rc;
    } else if (!x->two_step_rank && slots <= RFX_RANK_EMIT_MAX && one_launch_fits) {
        /* rank -> emit with no host round trip between them: the outputs are sized before the group count is known -- groups <= min(slots, selected
         * rows), a slice its share + 1 -- and the count comes back once everything is enqueued */
        int64_t bound = slots < G->seen ? slots : G->seen;
        if (bound < 1) bound = 1;
        const int64_t cap = bound / nsl + 1;
        if ((rc = rfx_hip_malloc(c, &h->dout, (size_t)(G->na + 1) * (size_t)cap * 8)) != RFX_OK) return rc;
        if ((G->want_first || !G->dense) && (rc = rfx_hip_malloc(c, &h->dfirst, (size_t)cap * 8)) != RFX_OK) return rc;
        for (int a = 0; a < G->na; a++) ptrs[a] = (int64_t *)h->dout + (size_t)(a + 1) * (size_t)cap;
        h->gstride = cap;
        rc = G->dense ? rfx_hip_group_rank_emit(c, h->aggs, &h->gt, G->total_rows, r0, nloc, nsl, sl, cap, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs, &h->groups)
                      : rfx_hip_hash_rank_emit(c, h->aggs, &h->ht, G->total_rows, r0, nloc, nsl, sl, cap, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs, &h->groups);
        if (x->timing) h->t_rank = now_ns();
        if (rc != RFX_OK) return rc;
        const int64_t g = h->groups;
        h->g0 = nsl > 1 ? RFX_SLICE_G0(g, sl, nsl) : 0;
        h->gn = nsl > 1 ? RFX_SLICE_GN(g, sl, nsl) : g;
        if (g == 0 || h->gn == 0) return RFX_OK;
    } else {
        rc = G->dense ? rfx_hip_group_rank(c, &h->gt, G->total_rows, &h->groups) : rfx_hip_hash_rank(c, &h->ht, G->total_rows, &h->groups);
        if (x->timing) h->t_rank = now_ns();
        if (rc != RFX_OK || h->groups == 0) return rc;
        const int64_t g = h->groups;
        int64_t g0 = 0, gn = g;
        if (nsl > 1) { /* this owner's range of the groups */
            g0 = RFX_SLICE_G0(g, sl, nsl);
            gn = RFX_SLICE_GN(g, sl, nsl);
        }
        h->g0 = g0;
        h->gn = gn;
        h->gstride = gn;
        if (gn == 0) return RFX_OK; /* (fewer groups than slices) */
        if ((rc = rfx_hip_malloc(c, &h->dout, (size_t)(G->na + 1) * (size_t)gn * 8)) != RFX_OK) return rc;
        if ((G->want_first || !G->dense) && (rc = rfx_hip_malloc(c, &h->dfirst, (size_t)gn * 8)) != RFX_OK) return rc;
        for (int a = 0; a < G->na; a++) ptrs[a] = (int64_t *)h->dout + (size_t)(a + 1) * (size_t)gn;
        if (nsl > 1 && (rc = rfx_hip_ctx_emit_window(c, g0, gn)) != RFX_OK) return rc;
        rc = G->dense ? rfx_hip_group_emit_sharded(c, h->aggs, &h->gt, r0, nloc, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs)
                      : rfx_hip_hash_emit_sharded(c, h->aggs, &h->ht, r0, nloc, (int64_t *)h->dout, (int64_t *)h->dfirst, ptrs);
        if (nsl > 1) rfx_hip_ctx_emit_window(c, 0, 0);
        if (rc != RFX_OK) return rc;
    }
    if (G->nsl > 1 && G->first_pass && G->nkeys > 1 && !G->rowhash) /* this slice's key columns, decoded from its composite keys (core/query.c:110-135) */
        for (int k = 0; k < G->nkeys; k++) {
            if ((rc = rfx_hip_malloc(c, &h->kc[k], (size_t)h->gn * 8)) != RFX_OK) return rc;
            if ((rc = rfx_hip_composite_decode(c, (const int64_t *)h->dout, h->gn, G->kmins[k], G->kmults[k], G->kmaxs[k] - G->kmins[k] + 1, (int64_t *)h->kc[k])) != RFX_OK) return rc;
        }
    /* FIRST values merge across the shards next: their streams must be idle.  A slice is read back on its own stream (fetch_all) and one
     * shard goes on in stream order: no wait (the tables go back to the pool of the stream that read them) */
    return (G->all_rank || x->timing) ? rfx_hip_ctx_sync(c) : RFX_OK;
}
/* FIRST columns of the shards beside a lead, added into the lead's (exactly one shard wrote each value) */
static int ph_first_local(void *arg, int s) {
    gq_t *G = (gq_t *)arg;
    rfx_exec_t *x = G->x;
    if (x->lead[s] != s) return RFX_OK;
    shard_t *h = &G->sh[s];
    const int64_t g = h->groups;
    for (int t = s + 1; t < G->S; t++) {
        if (x->lead[t] != s) continue;
        for (int a = 0; a < G->na; a++) {
            if (h->aggs[a].kind != RFX_AGG_FIRST) continue;
            const int rc = rfx_hip_add_i64(x->ctx[s], (int64_t *)h->dout + (size_t)(a + 1) * (size_t)h->gstride, (const int64_t *)G->sh[t].dout + (size_t)(a + 1) * (size_t)G->sh[t].gstride, g);
            if (rc != RFX_OK) return rc;
        }
    }
    return rfx_hip_ctx_sync(x->ctx[s]);
}
static int merge_first_values(gq_t *G) {
    rfx_exec_t *x = G->x;
    const int64_t g = G->sh[0].groups;
    int rc = RFX_OK;
    if (g == 0) return RFX_OK;
    if (G->S > x->ndev) rc = run_shards(x, ph_first_local, G);
    for (int a = 0; a < G->na && rc == RFX_OK; a++) {
        if (G->sh[0].aggs[a].kind != RFX_AGG_FIRST) continue;
        if (x->comm_all) {
            rfx_ctx_t *leads[RFX_MAX_SHARDS];
            int64_t *bufs[RFX_MAX_SHARDS];
            for (int d = 0; d < x->ndev; d++) {
                leads[d] = x->ctx[x->devlead[d]];
                bufs[d] = (int64_t *)G->sh[x->devlead[d]].dout + (size_t)(a + 1) * (size_t)G->sh[x->devlead[d]].gstride;
            }
            rc = rfx_dist_allreduce_i64_all(leads, x->ndev, bufs, g, 0);
            x->stat[RFX_XSTAT_MERGES_RCCL]++;
        }
        if (rc == RFX_OK && G->exch) rc = xp_allreduce(x, (int64_t *)G->sh[0].dout + (size_t)(a + 1) * (size_t)G->sh[0].gstride, g, 0, 0);
    }
    if (rc == RFX_OK && (x->comm_all || G->exch)) rc = run_shards(x, ph_sync, G);
    if (rc != RFX_OK && !x->err[0]) snprintf(x->err, sizeof(x->err), "%s", rfx_hip_last_error());
    return rc;
}
