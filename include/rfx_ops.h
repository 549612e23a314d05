/*
 * rfx_ops.h -- operator-level C ABI of librfx.so: the drop-in boundary for RayforceDB's select / where / by path.
 *
 * Every entry point has one of the reference's three operator shapes (core/ops.h:202-204)
 *
 *     obj_p f(obj_p)            unary_f          obj_p f(obj_p, obj_p)     binary_f          obj_p f(obj_p *, i64_t)   vary_f
 *
 * over the reference's 16-byte object header (include/rfx_abi.h), follows its ownership rule -- arguments are BORROWED,
 * the result is OWNED by the caller (core/eval.c:741-742,764-766,793-794) -- and reports errors by returning an object
 * of type TYPE_ERR obtained from the host's ray_err() (core/rayforce.h:292), never by longjmp / exit.  So a function
 * here can be bound with the reference's own plugin loader, unchanged:
 *
 *     (set gsel (loadfn "librfx.so" "rfx_select" 1))        ;; core/dynlib.c:153-216  dlopen(RTLD_NOW|RTLD_GLOBAL) + dlsym
 *     (gsel {s: (sum v) from: t where: (< a 100000) by: k})
 *
 * or linked in place of the reference's objects (Makefile:54-61 CORE_OBJECTS).  INTEGRATION.md shows both.
 *
 *   entry point            replaces (reference)                                  shape
 *   ---------------------  ----------------------------------------------------  -------
 *   rfx_select             ray_select            core/query.c:607-654            unary_f   (whole select dict)
 *   rfx_eq .. rfx_ge       ray_eq .. ray_ge      core/cmp.c:692-697              binary_f  -> B8 mask vector
 *   rfx_and, rfx_or        ray_and, ray_or       core/logic.c:262-264            vary_f    (over EVALUATED B8 masks: a loadfn
 *                                                                                           plugin is not a special form)
 *   rfx_where              ray_where             core/items.c:1366-1397          unary_f   -> ascending I64 row ids
 *   rfx_sum rfx_avg        ray_sum ray_avg       core/math.c:2388,2445-2526      unary_f   vector | MAPFILTER(val, ids)
 *   rfx_min rfx_max        ray_min ray_max       core/math.c:2428-2429           unary_f
 *   rfx_count rfx_first    ray_count ray_first   core/misc.c:43-60, core/items.c unary_f
 *   rfx_at                 at_ids via ray_at     core/rayforce.c:1100-1158       binary_f  (column, I64 ids) -> gathered column
 *   rfx_left_join          ray_left_join         core/join.c:158-198             vary_f    (key symbols, left table, right table)
 *   rfx_inner_join         ray_inner_join        core/join.c:200-298             vary_f
 *   rfx_add rfx_sub        ray_add ray_sub       core/math.c:2280-2345 (binop_map)   binary_f  vector (x) vector | atom -> vector
 *   rfx_mul rfx_div        ray_mul ray_fdiv (`div`)                                  binary_f  (i64 / f64, the reference's promotion)
 *   rfx_floordiv rfx_mod   ray_div (`/`: floor division, left operand's type) ray_mod (`%`)   binary_f  (core/math.c:1138-1364, 1449-1530)
 *
 * Everything below runs on the MI355X through the flat ABI of rfx_hip.h.  There is NO CPU implementation behind these
 * entry points: queries whose shape the GPU path does not cover are handed back to the host's own ray_* function when
 * the library runs as a plugin (the host exports them, rayforce.syms), and return an error object otherwise.
 *
 * Data residency: columns are host objects.  The first query that touches a column uploads it to HBM (PCIe) and keeps
 * it in a residency cache keyed by (payload pointer, length, type); an UNPINNED copy is proven current at every use -- by the
 * soft-dirty bits of the payload's pages where the kernel tracks them (O(pages)), else by a checksum of the whole payload -- and
 * refreshed when the host wrote into it; later queries run HBM-resident.  rfx_pin (trusted until rfx_invalidate / rfx_unpin)
 * and rfx_cache_clear make that explicit.
 */
#ifndef RFX_OPS_H
#define RFX_OPS_H

#include "rfx_abi.h"

#ifdef __cplusplus
extern "C" {
#endif
#ifndef __HIPCC_RTC__
#pragma GCC visibility push(default) /* librfx.so is built with -fvisibility=hidden: what the headers under include/ declare is its WHOLE dynamic surface (plugins are
                                      * dlopen'ed RTLD_GLOBAL, core/dynlib.c:131 -- internals must not land in the host's namespace) */
#endif

/* ---- host binding ------------------------------------------------------------------------------------------------
 * The shim needs the host's constructors (vector, table, i64, f64, b8, drop_obj, clone_obj, eval, ray_err,
 * symbols_intern, str_from_symbol: all in the reference's dynamic export list, rayforce.syms).  As a plugin it finds
 * them with dlsym(RTLD_DEFAULT) on first use.  Without a host process (tests, bench, the GPU box) it falls back to the
 * minimal object allocator in rfx_host.c.  Returns 1 = reference host bound, 0 = standalone host, <0 = error. */
int rfx_host_bind(void);
/* GPU ordinal used by the operator layer (default 0 or $RFX_DEVICE).  Call before the first operator. */
int rfx_ops_set_device(int device);
/* SHARDS: the operator layer plans through rfx_exec.h over one context per shard.  Before the first operator: the devices to use
 * (ndevices = 0: $RFX_DEVICES = "0,1,2" | "all", else the one device above) and how many shards in all (0: $RFX_SHARDS, else one per
 * device; more shards than devices share devices round robin).  With more than one shard every column is split row-range over the
 * shards when it is uploaded (rfx_pin / first touch), rfx_select answers from all of them -- each shard's pass on its own host thread,
 * the partial states merged by a device kernel (same device) and ONE fused RCCL exchange (across devices) before rank / emit, as
 * ray_select merges its pool workers' partials (core/query.c:607-654, core/aggr.c:163-181,375, core/pool.c:369-424) -- and the
 * operators that need a column whole on one device return an error object. */
int rfx_ops_set_shards(const int *devices, int ndevices, int nshards);
int rfx_ops_shards(void);
/* one process per device instead: the operator layer's context joins an RCCL communicator of `world` processes (id from rfx_dist_unique_id on
 * rank 0, shipped by the host); from then on rfx_select over this process' row range of the table answers for the WHOLE table */
int rfx_ops_dist_init(int world, int rank, const void *id128);
int rfx_ops_dist_finalize(void);
struct rfx_exec;
struct rfx_exec *rfx_ops_exec(void); /* the operator layer's planner (NULL before the first operator): counters, transport */
const char *rfx_ops_last_error(void);

/* ---- the operator surface ---------------------------------------------------------------------------------------- */
rfx_obj_p rfx_select(rfx_obj_p dict);

rfx_obj_p rfx_eq(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_ne(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_lt(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_gt(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_le(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_ge(rfx_obj_p x, rfx_obj_p y);

rfx_obj_p rfx_add(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_sub(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_mul(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_div(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_floordiv(rfx_obj_p x, rfx_obj_p y);
rfx_obj_p rfx_mod(rfx_obj_p x, rfx_obj_p y);

rfx_obj_p rfx_left_join(rfx_obj_p *x, int64_t n);  /* (keys symbol vector, left table, right table) */
rfx_obj_p rfx_inner_join(rfx_obj_p *x, int64_t n);

rfx_obj_p rfx_and(rfx_obj_p *x, int64_t n);
rfx_obj_p rfx_or(rfx_obj_p *x, int64_t n);
/* the same two as the SPECIAL FORMS the reference registers (FN_SPECIAL_FORM, core/env.c:224-225; logic_map evaluates its own arms,
 * core/logic.c:89-260): the arms arrive unevaluated.  Comparison trees over i64 / f64 vectors become one mask on the device; arms that are
 * B8 masks already take rfx_and / rfx_or; anything else is handed to the host's ray_and / ray_or.  These are the entry points to put in
 * place of ray_and / ray_or themselves (INTEGRATION.md sections 2 and 3). */
rfx_obj_p rfx_and_sf(rfx_obj_p *x, int64_t n);
rfx_obj_p rfx_or_sf(rfx_obj_p *x, int64_t n);
rfx_obj_p rfx_where(rfx_obj_p mask);
rfx_obj_p rfx_at(rfx_obj_p col, rfx_obj_p ids);

rfx_obj_p rfx_sum(rfx_obj_p x);
rfx_obj_p rfx_avg(rfx_obj_p x);
rfx_obj_p rfx_min(rfx_obj_p x);
rfx_obj_p rfx_max(rfx_obj_p x);
rfx_obj_p rfx_count(rfx_obj_p x);
rfx_obj_p rfx_first(rfx_obj_p x);

/* ---- residency ---------------------------------------------------------------------------------------------------- */
/* unary_f: (update {col: mapping ... from: t [where: p] [by: k]}) -- ray_update, core/update.c:936-1106: a NEW table whose named columns
 * carry the mapping's values at the selected rows (value i at row ids[i]; under by: every group's aggregate at all of its selected
 * rows; unknown names become new columns, null elsewhere).  `from:` must be a table value; the in-place form on a quoted global and
 * everything the device path does not cover go to the host's ray_update. */
rfx_obj_p rfx_update(rfx_obj_p update_dict);
/* unary_f: the reference's 7-slot group index of an i64 key column (index_group_i64_scoped, core/index.c:2002-2092), built on the
 * device: what a link-time replacement of index_group hands to the FN_AGGR built-ins inside a MAPGROUP pair.  rfx_sum .. rfx_first
 * accept such pairs (val, index) -- with indexes built here or by the reference -- besides vectors and MAPFILTER pairs. */
rfx_obj_p rfx_group(rfx_obj_p keys);
/* Residency (round 6): a cached device copy HOLDS A REFERENCE to its host vector (clone_obj / drop_obj, rayforce.syms:25-26).  By the reference's
 * own rule -- in-place writes only with rc == 1: cow_obj core/rayforce.c:3003-3026, core/math.c:2248,2310, every writer of core/update.c --
 * its cells cannot change and its address cannot be reused while the copy lives, so a later use is validated by ONE pointer compare, pinned or not,
 * and an UNPATCHED reference never sees a stale answer.  Entries whose vector nobody else refers to any more are released at the next operator call. */
rfx_obj_p rfx_pin(rfx_obj_p table_or_column);   /* unary_f: upload NOW + exempt from LRU eviction; returns a clone of its argument */
rfx_obj_p rfx_unpin(rfx_obj_p table_or_column); /* unary_f: drop the device copies (and the cache's references) */
/* unary_f: drop every cached device copy overlapping this vector / this table's columns.  Never needed under validation by ownership; in
 * checksum mode (below) it is what a host that writes PINNED vectors in place calls afterwards */
rfx_obj_p rfx_invalidate(rfx_obj_p table_or_column);
/* How cached copies are proven current: 0 = by ownership (default), 1 = by a checksum of the full payload on every use of an unpinned entry,
 * keyed by (payload address, length, type) -- for a host that writes payloads in place without looking at reference counts (raw views over the
 * standalone host's vectors).  Also RFX_VALIDATE=checksum in the environment.  Switching drops the cached copies. */
int rfx_ops_set_validation(int mode);
/* Reproducible grouped f64 sums (opt-in; also RFX_DETERMINISTIC=1): rfx_select runs every (sum x) / (avg x) over f64 under by: as an INTEGER sum over x
 * scaled by a power of two and rounded once per cell -- the same bits whatever order the rows reach their group in (the reference is bit-stable for a fixed
 * pool size, core/pool.c:415-424; the default path's f64 atomics are not).  mode 1: ONE i64 limb -- a cell is rounded to a multiple of 2^(e + b - 62)
 * (2^e > max |x|, 2^b >= rows): absolute, so a group of values far below the column's largest loses relative precision; mode 2 (RFX_DETERMINISTIC=2): a
 * SECOND limb adds up what the first one's cells rounded away, scaled by 2^(62 - b) more -- 2^(e + 2b - 124) per cell, below an f64 sum's own rounding for
 * any data whose magnitudes span less than ~2^60; one more i64 sum per aggregate.  The images of a resident column are made once and cached with it (8 bytes
 * of HBM per row and limb); a column holding a NaN or an infinity keeps the default path.  DESIGN.md section 4. */
int rfx_ops_set_deterministic(int mode);
/* One process per device (rfx_ops_dist_init, or a transport on the planner): 1 = a grouped rfx_select returns only THIS rank's range of the groups
 * (rfx_exec_split(groups, ranks, rank) in the answer's order: the ranks' tables end to end are the answer) instead of the whole answer on every rank; also
 * RFX_RANK_SLICES=1.  Nothing changes in a process of its own. */
int rfx_ops_set_rank_slices(int on);
/* unary_f: I64[17] counters since load: {selects on the GPU, selects delegated to the host, joins on the GPU, joins delegated,
 * uploads, cache hits, stale entries refreshed, operator calls, group scopes sampled, sampled scopes retried exactly,
 * materialised B8 mask passes (RFX_STAT_MASK_PASSES: a fused `where:` tree runs none), uses validated by soft-dirty page bits,
 * uses that cost a full-payload checksum (0 under ownership), uses validated by ownership, entries released because the cache held the
 * last reference, fixed-point images of resident f64 columns made for the reproducible sums, such images found again}; the argument is ignored */
rfx_obj_p rfx_stats(rfx_obj_p ignored);
void rfx_cache_clear(void);
int64_t rfx_cache_bytes(void);
/* statistics of the most recent rfx_select: 1 = ran on the GPU path, 0 = delegated to the host's ray_select */
int rfx_last_select_on_gpu(void);

/* ---- standalone host (rfx_host.c): just enough object model to build queries without the reference ---------------- */
rfx_obj_p rfx_host_vector(int8_t type, int64_t len);
/* A DEVICE column handle for standalone hosts that keep their columns in HBM already: a vector header whose payload is the cells' device
 * address instead of the cells.  nptrs = 1: one allocation (shards on the same device take their row ranges of it); nptrs = shards: one
 * address per shard (rows rfx_exec_split(len, shards, s)).  Types: I64 / F64 / SYMBOL / TIMESTAMP / B8.  The memory is borrowed: the
 * operators never upload, cache, validate or free it; rfx_host_drop frees the header only. */
rfx_obj_p rfx_host_device_vector(int8_t type, int64_t len, const void *const *d_ptrs, int nptrs);
rfx_obj_p rfx_host_i64(int64_t v);
rfx_obj_p rfx_host_f64(double v);
rfx_obj_p rfx_host_symbol(const char *name);             /* symbol atom */
rfx_obj_p rfx_host_list(int64_t len);                    /* LIST of `len` null slots; fill with RFX_AS_LIST */
rfx_obj_p rfx_host_table(rfx_obj_p keys, rfx_obj_p vals); /* takes ownership of both */
rfx_obj_p rfx_host_dict(rfx_obj_p keys, rfx_obj_p vals);
rfx_obj_p rfx_host_fn(const char *name);                 /* function object for "sum" "<" "and" ... bound to rfx_* */
rfx_obj_p rfx_host_clone(rfx_obj_p o);
void rfx_host_drop(rfx_obj_p o);
int64_t rfx_host_intern(const char *s, int64_t len);
const char *rfx_host_symbol_name(int64_t id);
const char *rfx_host_error_text(rfx_obj_p err);          /* message of an error object made by the standalone host */

#ifndef __HIPCC_RTC__
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RFX_OPS_H */
