/* rfx_ops_verbs.c -- part of the operator layer's ONE translation unit (rfx_ops.c #includes it -- the Makefile does not compile it on its own; the pieces share file-static state and helpers).
 * residency verbs (pin / unpin / invalidate), rfx_stats. */
/* ------------------------------------------------------------------------------------------------ residency verbs */
static obj_p pin_impl(obj_p x, int pin);
static obj_p pin_op(obj_p x, int pin) {
    op_begin();
    obj_p r = pin_impl(x, pin);
    op_end();
    return r;
}
static obj_p pin_impl(obj_p x, int pin) {
    rfx_host_bind();
    if (!x) return fail("pin: null argument");
    if (ensure_ctx() != RFX_OK) return fail_hip("no usable MI355X");
    obj_p cols = (x->type == RFX_TYPE_TABLE) ? RFX_AS_LIST(x)[1] : NULL;
    int64_t n = cols ? cols->len : 1;
    if (cols && is_parted_table(x)) { /* a get-parted table: its columns are known to the cache by their LIST objects (parted_view) */
        if (g_nshards > 1) return H.clone(x); /* (selects over parted tables are the host's under RFX_SHARDS / RFX_DEVICES: nothing to keep resident) */
        obj_p view = pin ? parted_view(x) : NULL;
        int bad = pin && !view;
        for (int64_t i = 0; i < n && !bad; i++) {
            obj_p c = RFX_AS_LIST(cols)[i];
            if (pin) {
                obj_p pc = RFX_AS_LIST(RFX_AS_LIST(view)[1])[i];
                const void *d;
                /* only 8-byte proxies: proxy_upload / proxy_sum address partitions as 8-byte cells (a B8 proxy would overrun its 1-byte-per-row
                 * device block and read past the mmapped partition files) */
                if (col_ctype(pc) && pc->type > 0 && resident(pc, 1, &d) != RFX_OK) bad = 1;
            } else {
                for (int j = 0; j < g_nres;) /* (every copy of it: row ranges and, after a join over shards, the whole copies) */
                    if (g_res[j].host == (const void *)c || g_res[j].host == RFX_AS_RAW(c)) res_free(j);
                    else j++;
            }
        }
        parted_view_release();
        return bad ? fail_hip("pin") : H.clone(x);
    }
    for (int64_t i = 0; i < n; i++) {
        obj_p c = cols ? RFX_AS_LIST(cols)[i] : x;
        if (!(c->type > 0 && (col_ctype(c) || c->type == RFX_TYPE_B8))) continue;
        if (pin) {
            const void *d;
            if (resident(c, 1, &d) != RFX_OK) return fail_hip("pin");
        } else {
            for (int j = 0; j < g_nres;)
                if (g_res[j].host == RFX_AS_RAW(c)) res_free(j);
                else j++;
        }
    }
    return H.clone(x);
}
rfx_obj_p rfx_pin(rfx_obj_p x) { return pin_op(x, 1); }
rfx_obj_p rfx_unpin(rfx_obj_p x) { return pin_op(x, 0); }
/* (rfx_invalidate x): the host is about to write (or has just written) into vector x / the columns of table x in place: every
 * cached device copy that overlaps their payload is dropped, pinned or not.  The hook a host patch calls from `set` on a column and
 * from the rc == 1 in-place arithmetic (core/math.c:2248, :2310), see INTEGRATION.md. */
rfx_obj_p rfx_invalidate(rfx_obj_p x) {
    rfx_host_bind();
    if (!x) return fail("invalidate: null argument");
    op_begin();
    if (x->type == RFX_TYPE_TABLE) {
        obj_p cols = RFX_AS_LIST(x)[1];
        for (int64_t i = 0; i < cols->len; i++) invalidate_payload(RFX_AS_LIST(cols)[i]);
    } else invalidate_payload(x);
    op_end();
    return H.clone(x);
}
/* (rfx_stats 0): counters since load as an I64 vector -- [selects answered on the GPU, selects handed back to the host's
 * ray_select, joins on the GPU, joins delegated, host-to-device uploads, cache hits, stale cache entries refreshed, operator
 * calls].  What a drop-in test asserts to know that an answer really came from the device. */
rfx_obj_p rfx_stats(rfx_obj_p x) {
    (void)x;
    rfx_host_bind();
    op_begin(); /* (the counters of other threads' operators stand still; entries only the cache refers to are released like at any call) */
    obj_p out = H.vector(RFX_TYPE_I64, 17);
    for (int i = 0; i < 10; i++) RFX_AS_I64(out)[i] = g_stat[i];
    if (g_x) { /* scopes sampled / sampled scopes retried exactly: the planner's counters */
        RFX_AS_I64(out)[ST_SCOPE_SAMPLED] = rfx_exec_stat(g_x, RFX_XSTAT_SCOPE_SAMPLED);
        RFX_AS_I64(out)[ST_SCOPE_RETRIED] = rfx_exec_stat(g_x, RFX_XSTAT_SCOPE_RETRIED);
    }
    RFX_AS_I64(out)[10] = g_ctx ? rfx_hip_ctx_stat(g_ctx, RFX_STAT_MASK_PASSES) : 0;
    RFX_AS_I64(out)[11] = g_sd_hits; /* unpinned columns proven current by soft-dirty page bits (0: the kernel has no such tracking) */
    RFX_AS_I64(out)[12] = g_sum_validations; /* uses of a cached column that cost a checksum over its whole payload (checksum mode only) */
    RFX_AS_I64(out)[13] = g_own_hits;        /* uses proven current by ownership: the object the cache holds a reference to */
    RFX_AS_I64(out)[14] = g_own_released;    /* entries released because the cache's reference was the last one */
    RFX_AS_I64(out)[15] = g_fix_built;       /* reproducible sums: fixed-point images of resident f64 columns made ... */
    RFX_AS_I64(out)[16] = g_fix_hits;        /* ... and found again by a later query */
    op_end();
    return out;
}
