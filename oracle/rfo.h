/*
 * oracle/rfo.h -- CPU restatement of the reference path.   *** TEST INFRASTRUCTURE ONLY ***
 * See rfo_core.c for the per-function reference citations.  Never linked into librfx.so.
 */
#ifndef RFO_H
#define RFO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
enum { RFO_I64 = 5, RFO_F64 = 10 };
enum { RFO_EQ = 0, RFO_NE, RFO_LT, RFO_GT, RFO_LE, RFO_GE };

void rfo_set_threads(int n); /* emulated pool executor count (reference: all cores, core/runtime.c:141-145) */
int rfo_get_threads(void);
int64_t rfo_pool_split_by_mem(int64_t input_len, int64_t groups_len, int64_t type_size);
int64_t rfo_pool_chunk_aligned(int64_t total_len, int64_t num_workers, int64_t elem_size);

void rfo_gen_i64(int64_t *out, int64_t n, uint64_t seed, int64_t row0, uint64_t mod);
void rfo_gen_f64(double *out, int64_t n, uint64_t seed, int64_t row0);

int rfo_cmp(int op, int ltype, const void *l, int l_atom, int rtype, const void *r, int r_atom, int64_t n, int8_t *out);
void rfo_logic(int is_or, int8_t *acc, const int8_t *next, int next_scalar, int64_t n);
int64_t rfo_where(const int8_t *mask, int64_t n, int64_t *ids);
void rfo_at_ids(const void *col, const int64_t *ids, int64_t m, void *out);

int64_t rfo_sum_i64(const int64_t *x, int64_t l);
double rfo_sum_f64(const double *x, int64_t l);
int64_t rfo_min_i64(const int64_t *x, int64_t l);
int64_t rfo_max_i64(const int64_t *x, int64_t l);
double rfo_min_f64(const double *x, int64_t l);
double rfo_max_f64(const double *x, int64_t l);
int64_t rfo_cnt_i64(const int64_t *x, int64_t l);
int64_t rfo_cnt_f64(const double *x, int64_t l);
double rfo_avg_i64(const int64_t *x, int64_t l);
double rfo_avg_f64(const double *x, int64_t l);

int rfo_binop(int op, int ltype, const void *l, int l_atom, int rtype, const void *r, int r_atom, int64_t n, void *out);
void rfo_aggr_fold(int fn, int type, const void *in, const int64_t *gids, int64_t len, int64_t groups, void *res);
void rfo_xbar_i64(const int64_t *x, int64_t n, int64_t y, int64_t *out);
void rfo_scope_i64(const int64_t *values, const int64_t *indices, int64_t len, int64_t *pmin, int64_t *pmax);
int64_t rfo_group_dense(const int64_t *values, const int64_t *indices, int64_t len, int64_t min, int64_t range, int64_t *hk,
                        int64_t *firsts, int64_t *gids);
int64_t rfo_group_sparse(const int64_t *values, const int64_t *indices, int64_t len, int64_t *gids, int64_t *firsts);
int rfo_composite_plan(const int64_t *mins, const int64_t *maxs, int ncols, int64_t *mults, int64_t *total_max);
void rfo_composite_key(const int64_t *const *cols, const int64_t *mins, const int64_t *mults, int ncols, const int64_t *indices,
                       int64_t len, int64_t *out);
uint64_t rfo_hash_fnv1a(int64_t key);
uint64_t rfo_hash_index_u64(uint64_t h, uint64_t k);

void rfo_aggr_sum_i64(const int64_t *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, int64_t *res);
void rfo_aggr_sum_f64(const double *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, double *res);
void rfo_aggr_min_i64(const int64_t *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, int64_t *res);
void rfo_aggr_max_i64(const int64_t *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, int64_t *res);
void rfo_aggr_min_f64(const double *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, double *res);
void rfo_aggr_max_f64(const double *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, double *res);
void rfo_aggr_count_i64(const int64_t *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, int64_t *res);
void rfo_aggr_count_f64(const double *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, int64_t *res);
void rfo_aggr_avg_i64(const int64_t *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, double *res);
void rfo_aggr_avg_f64(const double *in, const int64_t *gids, const int64_t *filter, int64_t len, int64_t groups, double *res);
void rfo_aggr_first(const void *in, const int64_t *firsts, const int64_t *filter, int64_t groups, void *out);
#ifdef __cplusplus
}
#endif
#endif
