/*
 * rfx_exec.h -- the PLANNER of librfx.so (rayforce_amd/csrc/rfx_exec.c): one query over one or several row-range SHARDS.
 *
 * Three layers, top to bottom:
 *   rfx_ops.h   obj_p in, obj_p out: parses the reference's select dict, keeps host columns resident, builds result tables
 *   rfx_exec.h  device columns in, device results out: WHICH kernels run, in which order, under which scope, and how the
 *               shards' partial states merge -- this file.  Every host of the library (rfx_select, the Python test host, bench.py,
 *               a C host with its own object model) plans through these entry points; there is no second planner.
 *   rfx_hip.h   the kernels, one context = one device + one stream
 *
 * What the reference does in the same place: ray_select (core/query.c:607-654) fans a fold / a group index out over its pool's
 * workers in row chunks (pool_run, core/pool.c:369-424; aggr_map, core/aggr.c:375) and merges the per-worker partial states INSIDE
 * the one evaluator process (AGGR_COLLECT core/aggr.c:163-181, unop_fold's second level core/math.c:2206-2228, the sparse path's
 * re-insertion core/index.c:1866-1906).  A shard here is such a worker one level up: a device (or a slice of one) that owns the rows
 * [row0, row0 + n) of every column.
 *
 * Sharding models, all through the same calls:
 *   - ONE process, N devices (the evaluator process of INTEGRATION.md): rfx_exec_create over N contexts, rfx_exec_comm_init_all;
 *     dense group tables merge by ONE fused RCCL all-reduce over xGMI (rfx_dist_group_tables_allreduce_all), everything the host
 *     can fold itself (scopes, scalar partials, flags) is folded on the host.  Every shard is driven by its own host thread.
 *   - several shards on ONE device (RFX_SHARDS=k: how the merge logic is tested on a 1-GPU box): merged by a device kernel.
 *   - one process per device (torch.distributed launches, bench.py --gpus N): the lead context carries an inter-process communicator
 *     (rfx_dist_init, or a transport the host supplies) and the same merges run as collectives.
 *
 * Column addresses: a query's descriptors (rfx_pred_t, rfx_agg_t, key columns) are written with SHARD 0's device addresses; `cols`
 * lists, for every column the query names, its address on every shard.  With one shard `cols` may be NULL.
 * Shard s of an n-row table owns rows rfx_exec_split(n, S, s): equal spans of ceil(n / S) rounded up to 512 rows.
 */
#ifndef RFX_EXEC_H
#define RFX_EXEC_H

#include "rfx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#ifndef __HIPCC_RTC__
#pragma GCC visibility push(default) /* librfx.so is built with -fvisibility=hidden: what the headers under include/ declare is its WHOLE dynamic surface (plugins are
                                      * dlopen'ed RTLD_GLOBAL, core/dynlib.c:131 -- internals must not land in the host's namespace) */
#endif

#define RFX_MAX_SHARDS 16
#define RFX_GROUPS_OWN (RFX_MAX_SHARDS * (4 + 2 * RFX_MAX_KEYS + RFX_EXEC_MAX_AGGS))
#define RFX_EXEC_MAX_AGGS 32 /* more than RFX_MAX_AGGS outputs run as several passes over the same selection / the same groups */

typedef struct rfx_exec rfx_exec_t;

/* ---- shards ---- */
/* The contexts are BORROWED (the caller destroys them after rfx_exec_destroy); several may share a device. */
int rfx_exec_create(rfx_ctx_t *const *ctxs, int nshards, rfx_exec_t **out);
int rfx_exec_destroy(rfx_exec_t *x);
int rfx_exec_shards(const rfx_exec_t *x);
rfx_ctx_t *rfx_exec_ctx(const rfx_exec_t *x, int shard);
void rfx_exec_split(int64_t nrows, int nshards, int shard, int64_t *row0, int64_t *len);
/* one process, several devices: RCCL communicators among the first shard of every distinct device (no-op with one device) */
int rfx_exec_comm_init_all(rfx_exec_t *x);
/* Inter-process exchange (one process per device).  Default: the lead context's RCCL communicator (rfx_dist_init), identity without
 * one.  A host with its own channel (the tests: torch.distributed / gloo between two ranks that share a GPU) supplies these instead;
 * `user` is handed back.  Every function returns RFX_OK or an RFX_E* code.
 *   world_rank     how many processes, which one am I
 *   allgather_host bytes of host memory from every process, rank order
 *   allreduce      in place over n 8-byte cells of DEVICE memory on the lead context; type 0 i64 / 1 f64, op 0 SUM / 1 MIN / 2 MAX
 *   allgather_dev  bytes of device memory from every process, rank order, into d_out (world * bytes) */
typedef struct rfx_transport {
    void *user;
    int (*world_rank)(void *user, int *world, int *rank);
    int (*allgather_host)(void *user, const void *in, size_t bytes, void *out);
    int (*allreduce)(void *user, void *d_buf, int64_t n, int type, int op);
    int (*allgather_dev)(void *user, const void *d_in, size_t bytes, void *d_out);
} rfx_transport_t;
int rfx_exec_set_transport(rfx_exec_t *x, const rfx_transport_t *t); /* NULL: back to the default */
/* The inter-process side as the planner itself sees it (the host's transport, else the lead context's RCCL communicator unless that one is process-local):
 * how many processes share the table (1: no exchange), and `bytes` of host memory from each of them in rank order -- for a caller that must agree on a
 * small fact before it builds the query (rfx_select's reproducible sums: one scale for all ranks). */
int rfx_exec_ranks(rfx_exec_t *x, int *rank); /* (rank may be NULL) */
int rfx_exec_allgather_host(rfx_exec_t *x, const void *in, size_t bytes, void *out);

/* ---- the query ---- */
typedef struct rfx_qcol {
    const void *d[RFX_MAX_SHARDS]; /* the column's address on every shard; d[0] is what the descriptors name */
} rfx_qcol_t;

enum {
    RFX_Q_NO_SAMPLED_SCOPE = 1, /* always the exact key scope (index_scope_i64's full pass) */
    RFX_Q_REFUSE_NULL_KEY = 2,  /* a selected null group key ends the call with RFX_EXEC_NULL_KEY before anything is grouped (rfx_select hands
                                 * such queries to the host: the reference opens one group per null-key row, core/index.c:1808-1816) */
    RFX_Q_WANT_FIRST = 4,       /* rfx_groups_t.d_first is wanted (the groups' first rows: costs one more result column) */
    RFX_Q_NO_SMALL = 8,         /* never the one-launch rank + emit of small dense tables (tests) */
    RFX_Q_PROBE_FIRST = 16,     /* hashed path, one shard: also leave, per row, the first row of its group (rfx_groups_t.d_probe) */
    RFX_Q_SLICED = 32           /* the caller reads the result through rfx_exec_groups_fetch_all only: the planner may leave it as SLICES -- after the
                                 * merge every device holds the whole tables, ranks them (the same order everywhere) and emits only ITS range of the
                                 * groups; fetch_all copies every slice into the host columns from the owning shard's own thread, over its own PCIe
                                 * link (rfx_groups_t.nslices / slice[]).  Without the flag -- or with one device, or with FIRST aggregates -- the
                                 * whole result is on shard 0 as before (nslices == 1) */
};
#define RFX_EXEC_NULL_KEY 1 /* positive: not an error, see RFX_Q_REFUSE_NULL_KEY */

typedef struct rfx_query {
    const rfx_pred_t *preds; /* where: comparisons (flat / two-level / tree form of rfx_pred_t) ... */
    int32_t npred, logic;
    const int8_t *d_mask;    /* ... or a B8 selection mask evaluated by the caller (trees the fused form cannot carry): then npred == 0.  One shard only */
    const rfx_agg_t *aggs;   /* up to RFX_EXEC_MAX_AGGS */
    int32_t nagg;
    int32_t nkeys;           /* by: columns (0: scalar aggregates / where) */
    const void *const *d_keys; /* i64-like device columns */
    const int64_t *kxbar;    /* per key: > 0 = bucket width of (xbar key width); may be NULL */
    int64_t nrows;           /* rows of the whole table (all shards of this process) */
    const rfx_qcol_t *cols;  /* per-shard addresses; NULL with one shard */
    int32_t ncols;
    int32_t flags;           /* RFX_Q_* */
    const int64_t *key_scope; /* optional {min, max}: a scope of key 0 the caller remembers (a superset of any selection's): saves the scope pass when LDS-sized */
    int64_t row0;            /* rfx_exec_where only: the id of the table's row 0 (ids come out as row0 + row) */
    /* rfx_exec_filter_aggr only -- the selection as ROW IDS, shard by shard (a lazy MAPFILTER (values, ids) pair, core/filter.c:29-49): shard s
     * folds its columns gathered at the sel_count[s] GLOBAL row ids d_sel_ids[s] (on shard s's device; every one inside the shard's row range
     * -- the caller vouches for that; order kept: FIRST is the value at the first id of the first non-empty shard).  npred must be 0 */
    const int64_t *const *d_sel_ids;
    const int64_t *sel_count;
} rfx_query_t;

/* ---- scalar aggregates: select {aggs} from t where p ---- */
int rfx_exec_filter_aggr(rfx_exec_t *x, const rfx_query_t *q, rfx_value_t *values, int64_t *selected);

/* ---- where: ascending GLOBAL row ids, one run per shard (shard order = row order) ---- */
typedef struct rfx_ids {
    int32_t nshards;
    int64_t total;
    int64_t count[RFX_MAX_SHARDS];
    int64_t *d_ids[RFX_MAX_SHARDS]; /* on the shard's device; NULL when count == 0 */
} rfx_ids_t;
int rfx_exec_where(rfx_exec_t *x, const rfx_query_t *q, rfx_ids_t *out);
void rfx_exec_ids_free(rfx_exec_t *x, rfx_ids_t *ids);

/* ---- group-by: select {aggs} from t where p by keys ---- */
enum { RFX_PATH_NONE = 0, RFX_PATH_DENSE = 1, RFX_PATH_DENSE_SMALL = 2, RFX_PATH_HASH = 3, RFX_PATH_ROWHASH = 4 };
typedef struct rfx_groups {
    int64_t groups;
    int32_t path;                     /* RFX_PATH_* */
    int32_t nkeys, nagg;
    int64_t *d_keys;                  /* one key: the groups' keys.  Several: the composite key / the row hash (see d_keycols) */
    int64_t *d_keycols[RFX_MAX_KEYS]; /* several keys: the result's key columns */
    int64_t *d_first;                 /* global first row of every group (RFX_Q_WANT_FIRST, and always on the hashed paths) */
    void *d_results[RFX_EXEC_MAX_AGGS];
    int32_t result_type[RFX_EXEC_MAX_AGGS]; /* RFX_I64 | RFX_F64 */
    int64_t *d_probe;                 /* RFX_Q_PROBE_FIRST */
    int64_t capacity;                 /* hashed paths: the table size the query ended with */
    /* small dense tables: everything above points into ONE device block that is mirrored on the host -- read results through
     * rfx_exec_groups_fetch and they cost no further round trip */
    const char *d_block, *h_block;
    size_t block_bytes;
    /* device blocks to release, and the shard whose context each came from.  Worst case by construction: a result of RFX_MAX_SHARDS slices registers per
     * slice its first rows, RFX_MAX_KEYS key columns (twice on the row-hash route: the proof passes' blocks) and one result block per pass
     * (<= RFX_EXEC_MAX_AGGS passes) */
    void *own[RFX_GROUPS_OWN];
    int8_t own_shard[RFX_GROUPS_OWN];
    int32_t nown;
    /* RFX_Q_SLICED: column c of the result = the concatenation of slice[0..nslices)'s pieces; slice i holds the groups [g0, g0 + n) on shard
     * `shard`.  The column pointers above are slice 0's (with one slice: the whole columns, as without the flag) */
    int32_t nslices;
    struct rfx_gslice {
        int32_t shard;
        int64_t g0, n;
        int64_t *d_keys, *d_first;
        int64_t *d_keycols[RFX_MAX_KEYS];
        void *d_results[RFX_EXEC_MAX_AGGS];
    } slice[RFX_MAX_SHARDS];
} rfx_groups_t;
int rfx_exec_group_by(rfx_exec_t *x, const rfx_query_t *q, rfx_groups_t *out);
int rfx_exec_groups_fetch(rfx_exec_t *x, const rfx_groups_t *g, void *dst, const void *d_src, size_t bytes); /* device result -> host (syncs) */
/* n result columns to host memory in ONE call: dst[i] receives the whole column src[i] -- a column pointer out of `g` (d_keys, d_first,
 * d_keycols[k], d_results[a]), groups * 8 bytes.  Every slice is copied by the shard that owns it, on that shard's host thread and stream;
 * one wait per shard at the end instead of one per column.  Columns above 64 MB go through pinned staging with several host threads
 * writing the destination (first-touch page faults of a freshly allocated vector are taken in parallel). */
int rfx_exec_groups_fetch_all(rfx_exec_t *x, const rfx_groups_t *g, int n, const void *const *d_srcs, void *const *dsts);
/* The groups [g0, g0 + n) of a result as a result of its own: a VIEW on the same device blocks (it owns nothing: release the original, not the view), read
 * through rfx_exec_groups_fetch / _fetch_all like any other.  What one rank of several returns when every rank is to keep only ITS range of the answer
 * (rfx_ops_set_rank_slices).  Results of one slice only (RFX_EINVAL otherwise). */
int rfx_exec_groups_window(const rfx_groups_t *g, int64_t g0, int64_t n, rfx_groups_t *out);
void rfx_exec_groups_free(rfx_exec_t *x, rfx_groups_t *g);

/* ---- join index (index_left_join_obj, core/index.c:2886-2928): d_ids[i] = first right row whose key tuple equals left row i's, else null.
 * RFX_ESTATE with *collision = 1: two key tuples share one 64-bit row hash (nothing may be used).  One shard. */
int rfx_exec_join_index(rfx_exec_t *x, const void *const *d_left_keys, const void *const *d_right_keys, int nkeys, int64_t nleft, int64_t nright,
                        int64_t *d_ids, int *collision);
/* ... over SHARDS: a broadcast join.  The BUILD side's key columns (drk) WHOLE on shard `shard`'s device, dlk this shard's nl rows of the left keys;
 * d_ids (on that device) receives per left row the first right row with an equal tuple -- a GLOBAL right row id -- or null.  Every shard builds the same
 * table and probes its own rows: no exchange.  To be called on the shard's thread (rfx_exec_run). */
int rfx_exec_join_index_shard(rfx_exec_t *x, int shard, const void *const *dlk, const void *const *drk, int nk, int64_t nl, int64_t nr, int64_t *d_ids, int *collision);

/* ---- counters since rfx_exec_create ---- */
enum {
    RFX_XSTAT_SCOPE_SAMPLED = 0, /* group-bys that ran under a sampled key scope */
    RFX_XSTAT_SCOPE_RETRIED = 1, /* ... whose pass reported a key outside it: run again under the exact scope */
    RFX_XSTAT_SCOPE_REMEMBERED = 2, /* group-bys that took the caller's remembered scope */
    RFX_XSTAT_HASH_GROWN = 3,    /* hashed tables that reported full and were grown */
    RFX_XSTAT_MERGES_KERNEL = 4, /* table sets merged by the same-device kernel */
    RFX_XSTAT_MERGES_RCCL = 5,   /* fused RCCL exchanges issued (process-local communicators) */
    RFX_XSTAT_MERGES_TRANSPORT = 6, /* inter-process exchanges issued */
    RFX_XSTAT_QUERIES = 7,
    RFX_XSTAT_SLICED = 8,        /* group-by results left as more than one slice */
    /* per-phase wall time of the calling thread, nanoseconds, accumulated over the group-bys since rfx_exec_timing(x, 1) (which zeroes
     * them): scope (samples, exact scopes, the scope exchange), pass (tables + scatter / aggregate kernels on every shard, to the last
     * shard's stream idle), merge (kernel merges, the fused exchange, copy-back), rank (slot ranking incl. its one round trip), emit
     * (+ key columns, FIRST values), fetch (rfx_exec_groups_fetch / _fetch_all: device -> host result).  With timing on every phase
     * ends with its shards' streams idle (a one-shard query otherwise runs on in stream order), so the sum is a little above the
     * untimed query */
    RFX_XSTAT_NS_SCOPE = 9,
    RFX_XSTAT_NS_PASS = 10,
    RFX_XSTAT_NS_MERGE = 11,
    RFX_XSTAT_NS_RANK = 12,
    RFX_XSTAT_NS_EMIT = 13,
    RFX_XSTAT_NS_FETCH = 14,
    RFX_XSTAT_NS_TOTAL = 15,     /* rfx_exec_group_by entry to exit, + the fetches */
    RFX_XSTAT_N = 16
};
/* what ONE phase hand-over to nshards - 1 worker threads costs the calling thread (microseconds; a bare pool without devices, `reps` empty
 * phases) -- the planner's own overhead per phase of a sharded query, which a one-GPU box can measure */
double rfx_exec_probe_handover_us(int nshards, int reps);
/* on = 1: zero the RFX_XSTAT_NS_* counters and time the phases from now on (a sync per phase); on = 0: stop */
void rfx_exec_timing(rfx_exec_t *x, int on);
/* fn(arg, shard) on EVERY shard at once -- shard 0 on the calling thread, the others on the planner's own shard threads (each with its shard's
 * context bound) -- the first failure's code back.  How the operator layer uploads a column: every shard's row range through its own device's
 * copy engine and PCIe link at the same time (rfx_pin / first touch), as the reference maps every column file where it lies (core/io.c:1310-1364). */
int rfx_exec_run(rfx_exec_t *x, int (*fn)(void *arg, int shard), void *arg);
int64_t rfx_exec_stat(const rfx_exec_t *x, int which);
/* forget which key columns' sampled scopes were reported too small (the planner does not sample those again: tests start over with this) */
void rfx_exec_forget_scopes(rfx_exec_t *x);
const char *rfx_exec_last_error(const rfx_exec_t *x);

#ifndef __HIPCC_RTC__
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RFX_EXEC_H */
