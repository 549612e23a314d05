/*
 * rfx_host.c -- minimal standalone host: just enough of RayforceDB's object model to build a select dictionary and
 * to receive result tables when no reference process is around (tests, bench, the GPU box).  Layout-compatible with
 * the reference (include/rfx_abi.h), NOT its allocator: objects are malloc'ed blocks whose payload sits 32-byte
 * aligned like the reference's buddy blocks (core/heap.c:45-55).  Reference counting follows core/rayforce.c:3003-3034
 * (clone = rc++, drop = rc-- and free at zero, lists recurse).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rfx_abi.h"
#include "rfx_ops.h"

static rfx_obj_t g_null = {.mmod = RFX_MMOD_INTERNAL, .type = RFX_TYPE_NULL, .rc = 1};
rfx_obj_p rfx_host_null(void) { return &g_null; }

static int elem_size(int8_t t) {
    switch (t < 0 ? -t : t) {
        case RFX_TYPE_B8: case RFX_TYPE_U8: return 1;
        case RFX_TYPE_I16: return 2;
        case RFX_TYPE_I32: case RFX_TYPE_DATE: case RFX_TYPE_TIME: return 4;
        default: return 8; /* I64 SYMBOL TIMESTAMP F64 and LIST (pointers) */
    }
}

/* Big blocks (>= 1 MB: result columns) are recycled like the reference's buddy heap recycles its blocks (core/heap.c): a fresh malloc of
 * that size is an mmap whose pages fault in one by one under the device-to-host copy (measured: 16 MB of result columns 0.34 ms
 * recycled, 1 - 1.8 ms fresh).  The block's capacity sits in the 16 pad bytes in front of the header. */
#include <pthread.h>
#define BIG_BLOCK ((size_t)1 << 20)
#define BIG_KEEP 24
static struct { void *blk; size_t cap; } g_big[BIG_KEEP];
static pthread_mutex_t g_big_lock = PTHREAD_MUTEX_INITIALIZER;
static rfx_obj_p alloc_obj(size_t payload) {
    /* block = [16 pad][16 header][payload] with the block 32-aligned => payload 32-aligned */
    void *blk = NULL;
    size_t cap = 32 + payload + 32;
    if (cap >= BIG_BLOCK) {
        pthread_mutex_lock(&g_big_lock);
        int best = -1;
        for (int i = 0; i < BIG_KEEP; i++)
            if (g_big[i].blk && g_big[i].cap >= cap && g_big[i].cap <= 2 * cap && (best < 0 || g_big[i].cap < g_big[best].cap)) best = i;
        if (best >= 0) {
            blk = g_big[best].blk;
            cap = g_big[best].cap;
            g_big[best].blk = NULL;
        }
        pthread_mutex_unlock(&g_big_lock);
    }
    if (!blk && posix_memalign(&blk, 32, cap)) return NULL;
    *(size_t *)blk = cap;
    rfx_obj_p o = (rfx_obj_p)((char *)blk + 16);
    memset(o, 0, sizeof(*o));
    o->mmod = RFX_MMOD_INTERNAL;
    o->rc = 1;
    return o;
}
static void free_obj(rfx_obj_p o) {
    void *blk = (char *)o - 16;
    const size_t cap = *(size_t *)blk;
    /* (the reference's heap_free keeps EVERY block in its buddy free lists, whatever its order -- core/heap.c:340-410, returned to the system by
     * heap_gc only; here blocks up to 1 GB -- the 800 MB columns of a 1e8-group result: freeing and faulting them in again cost 0.5 s + 0.5 s
     * per query -- are kept while the kept total stays below 16 GB; whole 8 GB table columns go straight back) */
    if (cap >= BIG_BLOCK && cap <= ((size_t)1 << 30)) {
        pthread_mutex_lock(&g_big_lock);
        size_t kept = 0;
        for (int i = 0; i < BIG_KEEP; i++)
            if (g_big[i].blk) kept += g_big[i].cap;
        for (int i = 0; i < BIG_KEEP && kept + cap <= ((size_t)16 << 30); i++)
            if (!g_big[i].blk) {
                g_big[i].blk = blk;
                g_big[i].cap = cap;
                blk = NULL;
                break;
            }
        pthread_mutex_unlock(&g_big_lock);
    }
    free(blk);
}

/* the blocks kept for reuse go back to the system (rfx_cache_clear: a long-lived standalone process gives up to 16 GB back after one large result) */
void rfx_host_trim(void) {
    void *blk[BIG_KEEP];
    int n = 0;
    pthread_mutex_lock(&g_big_lock);
    for (int i = 0; i < BIG_KEEP; i++)
        if (g_big[i].blk) {
            blk[n++] = g_big[i].blk;
            g_big[i].blk = NULL;
            g_big[i].cap = 0;
        }
    pthread_mutex_unlock(&g_big_lock);
    for (int i = 0; i < n; i++) free(blk[i]);
}

rfx_obj_p rfx_host_vector(int8_t type, int64_t len) {
    if (len < 0) return NULL;
    int8_t t = type < 0 ? -type : type;
    rfx_obj_p o = alloc_obj((size_t)len * elem_size(t));
    if (!o) return NULL;
    o->type = t;
    o->len = len;
    if (t == RFX_TYPE_LIST)
        for (int64_t i = 0; i < len; i++) RFX_AS_LIST(o)[i] = &g_null;
    return o;
}
rfx_obj_p rfx_host_list(int64_t len) { return rfx_host_vector(RFX_TYPE_LIST, len); }

static rfx_obj_p atom(int8_t t) {
    rfx_obj_p o = alloc_obj(0);
    if (o) o->type = -t;
    return o;
}
rfx_obj_p rfx_host_i64(int64_t v) { rfx_obj_p o = atom(RFX_TYPE_I64); if (o) o->i64 = v; return o; }
rfx_obj_p rfx_host_f64(double v) { rfx_obj_p o = atom(RFX_TYPE_F64); if (o) o->f64 = v; return o; }
rfx_obj_p rfx_host_b8(int8_t v) { rfx_obj_p o = atom(RFX_TYPE_B8); if (o) o->b8 = v; return o; }

/* ---- symbols: id = index into a growing table of strings ---- */
static char **g_syms;
static int64_t g_nsyms, g_capsyms;
int64_t rfx_host_intern(const char *s, int64_t len) {
    for (int64_t i = 0; i < g_nsyms; i++)
        if ((int64_t)strlen(g_syms[i]) == len && memcmp(g_syms[i], s, (size_t)len) == 0) return i;
    if (g_nsyms == g_capsyms) {
        g_capsyms = g_capsyms ? g_capsyms * 2 : 64;
        g_syms = (char **)realloc(g_syms, sizeof(char *) * (size_t)g_capsyms);
    }
    g_syms[g_nsyms] = strndup(s, (size_t)len);
    return g_nsyms++;
}
const char *rfx_host_symbol_name(int64_t id) { return (id >= 0 && id < g_nsyms) ? g_syms[id] : ""; }
rfx_obj_p rfx_host_symbol(const char *name) {
    rfx_obj_p o = atom(RFX_TYPE_SYMBOL);
    if (o) o->i64 = rfx_host_intern(name, (int64_t)strlen(name));
    return o;
}

static rfx_obj_p pair(int8_t type, rfx_obj_p keys, rfx_obj_p vals) {
    rfx_obj_p o = rfx_host_vector(RFX_TYPE_LIST, 2);
    if (!o) return NULL;
    o->type = type;
    RFX_AS_LIST(o)[0] = keys;
    RFX_AS_LIST(o)[1] = vals;
    return o;
}
rfx_obj_p rfx_host_table(rfx_obj_p keys, rfx_obj_p vals) { return pair(RFX_TYPE_TABLE, keys, vals); }
rfx_obj_p rfx_host_dict(rfx_obj_p keys, rfx_obj_p vals) { return pair(RFX_TYPE_DICT, keys, vals); }

rfx_obj_p rfx_host_clone(rfx_obj_p o) {
    if (o && o != &g_null) o->rc++;
    return o;
}
void rfx_host_drop(rfx_obj_p o) {
    if (!o || o == &g_null) return;
    if (o->type == RFX_TYPE_ERR) { free_obj(o); return; }
    if (--o->rc > 0) return;
    if (o->type == RFX_TYPE_LIST || o->type == RFX_TYPE_TABLE || o->type == RFX_TYPE_DICT || o->type == RFX_TYPE_MAPFILTER ||
        o->type == RFX_TYPE_MAPGROUP)
        for (int64_t i = 0; i < o->len; i++) rfx_host_drop(RFX_AS_LIST(o)[i]);
    free_obj(o);
}

/* error object: header + message text in the payload (the reference keeps the context in its VM instead) */
rfx_obj_p rfx_host_err(const char *msg) {
    size_t n = strlen(msg) + 1;
    rfx_obj_p o = alloc_obj(n);
    if (!o) return NULL;
    o->type = RFX_TYPE_ERR;
    o->len = (int64_t)n;
    memcpy(RFX_AS_RAW(o), msg, n);
    return o;
}
const char *rfx_host_error_text(rfx_obj_p err) { return (err && err->type == RFX_TYPE_ERR) ? (const char *)RFX_AS_RAW(err) : ""; }

/* the standalone "evaluator": tables evaluate to themselves, nothing else can be resolved without an environment */
rfx_obj_p rfx_host_eval(rfx_obj_p o) {
    if (o && (o->type == RFX_TYPE_TABLE || (o->type >= 0 && o->type <= RFX_TYPE_F64))) return rfx_host_clone(o);
    return rfx_host_err("standalone host cannot evaluate this expression (pass the table object itself as from:)");
}
