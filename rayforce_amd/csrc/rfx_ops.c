/*
 * rfx_ops.c -- the drop-in operator layer (host code stays in C, as in the reference): obj_p-shaped entry points that
 * plan a RayforceDB select / where / by query onto the flat HIP ABI (rfx_hip.h).
 *
 *   rfx_select walks the select dictionary the way ray_select does (core/query.c:243-654): `from:` is evaluated through
 *   the host's eval, `where:` is an expression LIST whose head is a function object (the parser already substituted the
 *   built-in for the symbol, core/parse.c:771-772), `by:` is a column symbol, every other key is an output mapping
 *   `(aggr col)`.  Supported shapes run as <= 4 kernel launches on HBM-resident columns; anything else goes back to
 *   the host's own ray_select (plugin mode) -- never to a CPU re-implementation of ours.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "rfx_abi.h"
#include "rfx_hip.h"
#include "rfx_exec.h"
#include "rfx_ops.h"

typedef rfx_obj_p obj_p;

/* rfx_host.c */
obj_p rfx_host_null(void);
obj_p rfx_host_b8(int8_t v);
obj_p rfx_host_err(const char *msg);
obj_p rfx_host_eval(obj_p o);
void rfx_host_trim(void);

/* ------------------------------------------------------------------------------------------------ host binding */
static struct {
    int bound; /* 0 = not yet, 1 = reference host, 2 = standalone */
    obj_p (*vector)(int8_t, int64_t);
    obj_p (*table)(obj_p, obj_p);
    obj_p (*i64)(int64_t);
    obj_p (*f64)(double);
    void (*drop)(obj_p);
    obj_p (*clone)(obj_p);
    obj_p (*eval)(obj_p);
    obj_p (*err)(const char *);
    int64_t (*intern)(const char *, int64_t);
    const char *(*symname)(int64_t);
    obj_p null_obj;
    /* the host's own built-ins, for recognising function objects inside parsed expressions and for delegation */
    void *f[32];
} H;
enum { F_SUM, F_AVG, F_MIN, F_MAX, F_COUNT, F_FIRST, F_EQ, F_NE, F_LT, F_GT, F_LE, F_GE, F_AND, F_OR, F_SELECT, F_ADD, F_SUB, F_MUL, F_FDIV, F_DIV, F_MOD, F_XBAR, F_LJ, F_IJ, F_UPDATE, F_TAKE, F_IN, F_WITHIN, F_NOT, F_N };
static const char *HOST_FN[F_N] = {"ray_sum", "ray_avg", "ray_min", "ray_max", "ray_count", "ray_first", "ray_eq",  "ray_ne",  "ray_lt",  "ray_gt",
                                   "ray_le",  "ray_ge",  "ray_and", "ray_or",  "ray_select", "ray_add",  "ray_sub", "ray_mul", "ray_fdiv", "ray_div", "ray_mod", "ray_xbar",
                                   "ray_left_join", "ray_inner_join", "ray_update", "ray_take", "ray_in", "ray_within", "ray_not"}; /* (in / within / not: recognised inside where: only) */
/* xbar is recognised inside `by:` only; the standalone object model still needs a distinct function object for it:
 * this stub is never called by this library. */
static obj_p x_stub_xbar(obj_p a, obj_p b) { (void)a; (void)b; return NULL; }
static void *OUR_FN[F_N];
static void *g_host_where, *g_host_at, *g_host_group; /* the host's built-ins behind rfx_where / rfx_at / rfx_group (NULL without a host) */
static char g_err[640];
static int g_last_gpu = 0;

const char *rfx_ops_last_error(void) { return g_err; }
int rfx_last_select_on_gpu(void) { return g_last_gpu; }

int rfx_host_bind(void) {
    if (H.bound) return H.bound == 1;
    OUR_FN[F_SUM] = (void *)rfx_sum; OUR_FN[F_AVG] = (void *)rfx_avg; OUR_FN[F_MIN] = (void *)rfx_min; OUR_FN[F_MAX] = (void *)rfx_max;
    OUR_FN[F_COUNT] = (void *)rfx_count; OUR_FN[F_FIRST] = (void *)rfx_first; OUR_FN[F_EQ] = (void *)rfx_eq; OUR_FN[F_NE] = (void *)rfx_ne;
    OUR_FN[F_LT] = (void *)rfx_lt; OUR_FN[F_GT] = (void *)rfx_gt; OUR_FN[F_LE] = (void *)rfx_le; OUR_FN[F_GE] = (void *)rfx_ge;
    OUR_FN[F_AND] = (void *)rfx_and; OUR_FN[F_OR] = (void *)rfx_or; OUR_FN[F_SELECT] = (void *)rfx_select;
    OUR_FN[F_ADD] = (void *)rfx_add; OUR_FN[F_SUB] = (void *)rfx_sub; OUR_FN[F_MUL] = (void *)rfx_mul; OUR_FN[F_FDIV] = (void *)rfx_div; OUR_FN[F_DIV] = (void *)rfx_floordiv; OUR_FN[F_MOD] = (void *)rfx_mod;
    OUR_FN[F_XBAR] = (void *)x_stub_xbar;
    OUR_FN[F_LJ] = (void *)rfx_left_join; OUR_FN[F_IJ] = (void *)rfx_inner_join; OUR_FN[F_UPDATE] = (void *)rfx_update;
    void *v = dlsym(RTLD_DEFAULT, "vector"), *t = dlsym(RTLD_DEFAULT, "table"), *e = dlsym(RTLD_DEFAULT, "eval");
    void *rs = dlsym(RTLD_DEFAULT, "ray_select"), *nu = dlsym(RTLD_DEFAULT, "__NULL_OBJ");
    if (v && t && e && rs && nu && !getenv("RFX_FORCE_STANDALONE")) {
        H.vector = (obj_p(*)(int8_t, int64_t))v;
        H.table = (obj_p(*)(obj_p, obj_p))t;
        H.eval = (obj_p(*)(obj_p))e;
        H.i64 = (obj_p(*)(int64_t))dlsym(RTLD_DEFAULT, "i64");
        H.f64 = (obj_p(*)(double))dlsym(RTLD_DEFAULT, "f64");
        H.drop = (void (*)(obj_p))dlsym(RTLD_DEFAULT, "drop_obj");
        H.clone = (obj_p(*)(obj_p))dlsym(RTLD_DEFAULT, "clone_obj");
        H.err = (obj_p(*)(const char *))dlsym(RTLD_DEFAULT, "ray_err");
        H.intern = (int64_t(*)(const char *, int64_t))dlsym(RTLD_DEFAULT, "symbols_intern");
        H.symname = (const char *(*)(int64_t))dlsym(RTLD_DEFAULT, "str_from_symbol");
        H.null_obj = (obj_p)nu;
        for (int i = 0; i < F_N; i++) H.f[i] = dlsym(RTLD_DEFAULT, HOST_FN[i]);
        g_host_where = dlsym(RTLD_DEFAULT, "ray_where"); /* (not in H.f: never recognised inside a query, only handed back to -- refused1x) */
        g_host_at = dlsym(RTLD_DEFAULT, "ray_at");
        g_host_group = dlsym(RTLD_DEFAULT, "ray_group");
        if (H.i64 && H.f64 && H.drop && H.clone && H.err && H.intern && H.symname) {
            H.bound = 1;
            return 1;
        }
    }
    H.vector = rfx_host_vector;
    H.table = rfx_host_table;
    H.i64 = rfx_host_i64;
    H.f64 = rfx_host_f64;
    H.drop = rfx_host_drop;
    H.clone = rfx_host_clone;
    H.eval = rfx_host_eval;
    H.err = rfx_host_err;
    H.intern = rfx_host_intern;
    H.symname = rfx_host_symbol_name;
    H.null_obj = rfx_host_null();
    memset(H.f, 0, sizeof(H.f));
    H.bound = 2;
    return 0;
}

obj_p rfx_host_fn(const char *name) {
    static const struct { const char *n; int f; int type; int attrs; } T[] = {
        {"sum", F_SUM, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"avg", F_AVG, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"min", F_MIN, RFX_TYPE_UNARY, RFX_FN_AGGR},
        {"max", F_MAX, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"count", F_COUNT, RFX_TYPE_UNARY, RFX_FN_AGGR}, {"first", F_FIRST, RFX_TYPE_UNARY, RFX_FN_AGGR},
        {"==", F_EQ, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"!=", F_NE, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"<", F_LT, RFX_TYPE_BINARY, RFX_FN_ATOMIC},
        {">", F_GT, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"<=", F_LE, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {">=", F_GE, RFX_TYPE_BINARY, RFX_FN_ATOMIC},
        {"and", F_AND, RFX_TYPE_VARY, RFX_FN_SPECIAL_FORM}, {"or", F_OR, RFX_TYPE_VARY, RFX_FN_SPECIAL_FORM}, {"select", F_SELECT, RFX_TYPE_UNARY, 0},
        {"+", F_ADD, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"-", F_SUB, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"*", F_MUL, RFX_TYPE_BINARY, RFX_FN_ATOMIC},
        {"div", F_FDIV, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"/", F_DIV, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"%", F_MOD, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"xbar", F_XBAR, RFX_TYPE_BINARY, RFX_FN_ATOMIC}, {"update", F_UPDATE, RFX_TYPE_UNARY, 0}};
    rfx_host_bind();
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); i++)
        if (strcmp(T[i].n, name) == 0) {
            obj_p o = rfx_host_i64((int64_t)(intptr_t)OUR_FN[T[i].f]);
            o->type = (int8_t)T[i].type; /* function objects carry the POSITIVE type code (core/env.c:66-74) */
            o->attrs = (uint8_t)T[i].attrs;
            return o;
        }
    return NULL;
}

static obj_p fail(const char *msg) {
    rfx_host_bind();
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return H.err(msg);
}
static obj_p fail_hip(const char *what) {
    char b[600];
    snprintf(b, sizeof(b), "%s: %s", what, rfx_hip_last_error());
    return fail(b);
}
static int g_refused_sharded;
static obj_p fail_ctx(void) {
    if (g_refused_sharded) return fail("this operator (or this shape of its arguments) needs its columns whole on one device: with RFX_SHARDS / RFX_DEVICES it is the host's own built-in -- joins, update, at over unordered ids, group over sparse keys, filtered MAPGROUP pairs");
    return fail_hip("no usable MI355X");
}
/* ... unless there is a host beside us: then the operator is simply the host's own again (the shards hold row ranges; RFX_SHARDS /
 * RFX_DEVICES is about rfx_select).  Defined below, once HOST_CALL is. */
static obj_p refused1(int f, obj_p x);
__attribute__((unused)) static obj_p refused2(int f, obj_p x, obj_p y);
static obj_p refusedn(int f, obj_p *x, int64_t n);

/* which built-in does this function object denote? -1 if none */
static int fn_id(obj_p o) {
    if (!o || (o->type != RFX_TYPE_UNARY && o->type != RFX_TYPE_BINARY && o->type != RFX_TYPE_VARY)) return -1;
    void *p = (void *)(intptr_t)o->i64;
    for (int i = 0; i < F_N; i++)
        if ((OUR_FN[i] && p == OUR_FN[i]) || (H.f[i] && p == H.f[i])) return i;
    return -1;
}

/* The operator layer by concern (round 5: one 3 100-line file before).  ONE translation unit -- the pieces share the lock, the residency cache and the
 * per-call scratch lists as file statics -- read in this order: */
#include "rfx_ops_residency.c"
#include "rfx_ops_repro.c"
#include "rfx_ops_plan.c"
#include "rfx_ops_select.c"
#include "rfx_ops_update.c"
#include "rfx_ops_operators.c"
#include "rfx_ops_join.c"
#include "rfx_ops_folds.c"
#include "rfx_ops_verbs.c"
