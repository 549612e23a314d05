// rfx_multikey.hip -- several `by:` columns folded into ONE dense i64 key.
//
// Reference: index_group_list_perfect (core/index.c:2308-2424).  When the product of the per-column key ranges fits a signed
// 64-bit integer, every row gets the composite key  sum_c (col_c[row] - min_c) * mult_c  (mult_0 = 1, mult_c = mult_{c-1} *
// range_{c-1}) and the single-key index (index_group_i64_scoped with the forced scope {0, max}) does the grouping; the
// result's key columns are the source columns at the groups' first rows (core/query.c:93-135).
//
// Here: rfx_composite_plan is the host-side multiplier / overflow logic (:2364-2383), k_composite_key the one streaming
// pass that writes the composite column (NK x 8 B/row in, 8 B/row out; the reference materialises the same column, :2386),
// k_composite_decode turns the emitted composite keys back into one key column (groups elements; replaces the reference's
// at_ids gather over first rows, which on several GPUs would need rows another rank owns).
#include "rfx_group_common.hpp"

struct KeyParts {
    int n;
    const u64 *col[RFX_MAX_KEYS];
    u64 min[RFX_MAX_KEYS];
    u64 mult[RFX_MAX_KEYS];
};

extern "C" int rfx_composite_plan(const int64_t *mins, const int64_t *maxs, int nkeys, int64_t *mults, int64_t *total_max) {
    RFX_REQUIRE(mins && maxs && mults && total_max, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nkeys >= 1 && nkeys <= RFX_MAX_KEYS, RFX_ELIMIT, "1..RFX_MAX_KEYS key columns");
    u64 product = 1, smax = 0;
    for (int i = 0; i < nkeys; i++) {
        const i64 range = (i64)((u64)maxs[i] - (u64)mins[i] + 1u);
        // a null key (INT64_MIN) makes the range wrap: the reference gives up on the perfect path there (row-hash path)
        if (range <= 0 || product > (u64)INT64_MAX / (u64)range) {
            rfx_set_error("rfx_composite_plan: key ranges overflow a 64-bit composite key (column %d)", i);
            return RFX_ELIMIT;
        }
        mults[i] = (i64)product;
        product *= (u64)range;
        const u64 delta = (u64)(maxs[i] - mins[i]) * (u64)mults[i];
        if (smax > (u64)INT64_MAX - delta) {
            rfx_set_error("rfx_composite_plan: composite key maximum overflows (column %d)", i);
            return RFX_ELIMIT;
        }
        smax += delta;
    }
    *total_max = (i64)smax;
    return RFX_OK;
}

// A wave owns 512 consecutive rows per step (as k_cmp_mask): lane l takes rows 2l, 2l+1 of each 128-row group -- one
// 16-byte load per key column and group (1 KB contiguous per wave instruction) and one 16-byte store of the two composites.
template <int NK>
__global__ __launch_bounds__(RFX_BLOCK) void k_composite_key(const KeyParts K, i64 nrows, u64 *__restrict__ out) {
    u64 mn[NK], mu[NK];
#pragma unroll
    for (int c = 0; c < NK; c++) {
        mn[c] = K.min[c];
        mu[c] = K.mult[c];
    }
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nfull = nrows / 512;
    for (i64 q = wave_id; q < nfull; q += nwaves) {
        const i64 base = q * 512 + lane * 2;
        u64 k[8];
#pragma unroll
        for (int e = 0; e < 8; e++) k[e] = 0;
#pragma unroll
        for (int c = 0; c < NK; c++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u64x2 t = rfx_ld2(K.col[c] + base + j * 128);
                k[2 * j] += (t.x - mn[c]) * mu[c];
                k[2 * j + 1] += (t.y - mn[c]) * mu[c];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u64x2 o;
            o.x = k[2 * j];
            o.y = k[2 * j + 1];
            *(u64x2 *)(out + base + j * 128) = o;
        }
    }
    if (blockIdx.x == 0) {
        for (i64 r = nfull * 512 + threadIdx.x; r < nrows; r += RFX_BLOCK) {
            u64 k = 0;
#pragma unroll
            for (int c = 0; c < NK; c++) k += (K.col[c][r] - mn[c]) * mu[c];
            out[r] = k;
        }
    }
}

extern "C" int rfx_hip_composite_key(rfx_ctx_t *c, const void *const *d_cols, const int64_t *mins, const int64_t *mults, int nkeys,
                                     int64_t nrows, int64_t *d_out) {
    RFX_REQUIRE(c && d_cols && mins && mults, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nkeys >= 1 && nkeys <= RFX_MAX_KEYS, RFX_ELIMIT, "1..RFX_MAX_KEYS key columns");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_out && ((uintptr_t)d_out & 15) == 0, RFX_EINVAL, "output must be a 16-byte aligned device buffer");
    KeyParts K;
    memset(&K, 0, sizeof(K));
    K.n = nkeys;
    for (int i = 0; i < nkeys; i++) {
        RFX_REQUIRE(d_cols[i] && ((uintptr_t)d_cols[i] & 15) == 0, RFX_EINVAL, "key columns must be 16-byte aligned device buffers");
        K.col[i] = (const u64 *)d_cols[i];
        K.min[i] = (u64)mins[i];
        K.mult[i] = (u64)mults[i];
    }
    const int grid = c->num_cus * 8;
    RFX_KERNEL_BEGIN(c);
#define RFX_CK(N) case N: hipLaunchKernelGGL((k_composite_key<N>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, K, (i64)nrows, (u64 *)d_out); break
    switch (nkeys) {
        RFX_CK(1);
        RFX_CK(2);
        RFX_CK(3);
        RFX_CK(4);
        RFX_CK(5);
        RFX_CK(6);
        RFX_CK(7);
        default: RFX_CK(8);
    }
#undef RFX_CK
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

__global__ __launch_bounds__(RFX_BLOCK) void k_composite_decode(const u64 *__restrict__ comp, i64 n, u64 mn, u64 mult, u64 range, u64 *__restrict__ out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) out[i] = mn + (comp[i] / mult) % range;
}

extern "C" int rfx_hip_composite_decode(rfx_ctx_t *c, const int64_t *d_comp, int64_t n, int64_t min, int64_t mult, int64_t range, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_comp && d_out, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(mult >= 1 && range >= 1, RFX_EINVAL, "multiplier and range must be positive");
    i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    int grid = rfx_grid(c);
    if (blocks < grid) grid = (int)blocks;
    hipLaunchKernelGGL(k_composite_decode, dim3(grid), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_comp, (i64)n, (u64)min, (u64)mult, (u64)range, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- time / value bucketing as a group key: (xbar col width), XBARI64 core/ops.h:192-193, ray_xbar_partial core/math.c:1635 ----
// out = null when x is null, else ((x < 0 ? x + 1 - w : x) / w) * w -- floor to a multiple of w for w > 0 (C division truncates
// towards zero, hence the shift for negative x).  Only w > 0 is taken here; the adjustment wraps like the reference's.
__global__ __launch_bounds__(RFX_BLOCK) void k_xbar_i64(const u64 *__restrict__ in, i64 nrows, i64 w, u64 *__restrict__ out) {
    const i64 npairs = nrows / 2;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < npairs; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64x2 t = rfx_ld2(in + 2 * i);
        u64x2 o;
        const i64 x0 = (i64)t.x, x1 = (i64)t.y;
        o.x = (x0 == RFX_NULL_I64_D) ? t.x : (u64)(((x0 < 0) ? (i64)((u64)x0 + 1u - (u64)w) : x0) / w * w);
        o.y = (x1 == RFX_NULL_I64_D) ? t.y : (u64)(((x1 < 0) ? (i64)((u64)x1 + 1u - (u64)w) : x1) / w * w);
        *(u64x2 *)(out + 2 * i) = o;
    }
    if ((nrows & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const i64 x = (i64)in[nrows - 1];
        out[nrows - 1] = (x == RFX_NULL_I64_D) ? (u64)x : (u64)(((x < 0) ? (i64)((u64)x + 1u - (u64)w) : x) / w * w);
    }
}

extern "C" int rfx_hip_xbar_i64(rfx_ctx_t *c, const int64_t *d_col, int64_t nrows, int64_t width, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    RFX_REQUIRE(width > 0, RFX_EINVAL, "xbar width must be positive on this path");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_col && d_out && ((uintptr_t)d_col & 15) == 0 && ((uintptr_t)d_out & 15) == 0, RFX_EINVAL, "16-byte aligned device buffers expected");
    hipLaunchKernelGGL(k_xbar_i64, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_col, (i64)nrows, (i64)width, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- several key columns whose ranges do NOT multiply into 64 bits: the row-hash path (index_group_list, core/index.c:2731-2790) ----
// The reference hashes every row's key tuple -- h = U64_HASH_SEED, then column by column h = hash_index_u64(h, col_c[row]) for
// unfiltered i64-like columns (hash_index_i64_batch, core/hash.h:130-143) but hash_index_u64(col_c[row], h) -- arguments the
// other way round -- under a filter and for f64 columns (index_hash_obj_partial, core/index.c:155-175): `value_first` picks --
// and groups rows by hash with a full tuple comparison on collision (__index_list_cmp_row, :59-104).  k_row_hash writes that same hash as one i64 column
// (NK x 8 B/row in, 8 B/row out); the sparse-key group-by then runs on it, and the caller proves the absence of collisions
// from per-group min == max of every key column (Engine._group_by_row_hash).  With the reference's own hash the radix order
// of its multi-threaded path (hash & 1023, then first occurrence; :2465-2729) can be reproduced exactly.
template <int NK>
__global__ __launch_bounds__(RFX_BLOCK) void k_row_hash(const KeyParts K, i64 nrows, int value_first, u64 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * (RFX_BLOCK / RFX_WAVE) + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (RFX_BLOCK / RFX_WAVE);
    const i64 nfull = nrows / 512;
    for (i64 q = wave_id; q < nfull; q += nwaves) {
        const i64 base = q * 512 + lane * 2;
        u64 h[8];
#pragma unroll
        for (int e = 0; e < 8; e++) h[e] = RFX_U64_HASH_SEED;
#pragma unroll
        for (int c = 0; c < NK; c++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u64x2 t = rfx_ld2(K.col[c] + base + j * 128);
                h[2 * j] = value_first ? rfx_hash_index_u64(t.x, h[2 * j]) : rfx_hash_index_u64(h[2 * j], t.x);
                h[2 * j + 1] = value_first ? rfx_hash_index_u64(t.y, h[2 * j + 1]) : rfx_hash_index_u64(h[2 * j + 1], t.y);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u64x2 o;
            o.x = h[2 * j];
            o.y = h[2 * j + 1];
            *(u64x2 *)(out + base + j * 128) = o;
        }
    }
    if (blockIdx.x == 0) {
        for (i64 r = nfull * 512 + threadIdx.x; r < nrows; r += RFX_BLOCK) {
            u64 h = RFX_U64_HASH_SEED;
#pragma unroll
            for (int c = 0; c < NK; c++) h = value_first ? rfx_hash_index_u64(K.col[c][r], h) : rfx_hash_index_u64(h, K.col[c][r]);
            out[r] = h;
        }
    }
}

extern "C" int rfx_hip_row_hash(rfx_ctx_t *c, const void *const *d_cols, int nkeys, int64_t nrows, int value_first, int64_t *d_out) {
    RFX_REQUIRE(c && d_cols, RFX_EINVAL, "NULL argument");
    RFX_REQUIRE(nkeys >= 1 && nkeys <= RFX_MAX_KEYS, RFX_ELIMIT, "1..RFX_MAX_KEYS key columns");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_out && ((uintptr_t)d_out & 15) == 0, RFX_EINVAL, "output must be a 16-byte aligned device buffer");
    KeyParts K;
    memset(&K, 0, sizeof(K));
    K.n = nkeys;
    for (int i = 0; i < nkeys; i++) {
        RFX_REQUIRE(d_cols[i] && ((uintptr_t)d_cols[i] & 15) == 0, RFX_EINVAL, "key columns must be 16-byte aligned device buffers");
        K.col[i] = (const u64 *)d_cols[i];
    }
    const int grid = c->num_cus * 8;
    RFX_KERNEL_BEGIN(c);
#define RFX_RH(N) case N: hipLaunchKernelGGL((k_row_hash<N>), dim3(grid), dim3(RFX_BLOCK), 0, c->stream, K, (i64)nrows, value_first, (u64 *)d_out); break
    switch (nkeys) {
        RFX_RH(1);
        RFX_RH(2);
        RFX_RH(3);
        RFX_RH(4);
        RFX_RH(5);
        RFX_RH(6);
        RFX_RH(7);
        default: RFX_RH(8);
    }
#undef RFX_RH
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// out = (x is null) ? repl : x -- lets min / max (which skip nulls) see a null key as a value of its own in the collision proof
__global__ __launch_bounds__(RFX_BLOCK) void k_replace_null_i64(const u64 *__restrict__ in, i64 nrows, u64 repl, u64 *__restrict__ out) {
    const i64 npairs = nrows / 2;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < npairs; i += (i64)gridDim.x * RFX_BLOCK) {
        u64x2 t = rfx_ld2(in + 2 * i);
        t.x = ((i64)t.x == RFX_NULL_I64_D) ? repl : t.x;
        t.y = ((i64)t.y == RFX_NULL_I64_D) ? repl : t.y;
        *(u64x2 *)(out + 2 * i) = t;
    }
    if ((nrows & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[nrows - 1] = ((i64)in[nrows - 1] == RFX_NULL_I64_D) ? repl : in[nrows - 1];
}

extern "C" int rfx_hip_replace_null_i64(rfx_ctx_t *c, const int64_t *d_col, int64_t nrows, int64_t repl, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_col && d_out && ((uintptr_t)d_col & 15) == 0 && ((uintptr_t)d_out & 15) == 0, RFX_EINVAL, "16-byte aligned device buffers expected");
    hipLaunchKernelGGL(k_replace_null_i64, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_col, (i64)nrows, (u64)repl, (u64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// out = (x == from) ? to : x, any alignment (the key columns of a sharded row-hash result: a null key rides through the MIN / MAX proof
// as a value no key has and comes back as the null)
// (no __restrict__: d_out may BE d_col -- the row-hash proof passes replace in place)
__global__ __launch_bounds__(RFX_BLOCK) void k_replace_i64(const i64 *in, i64 nrows, i64 from, i64 to, i64 *out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < nrows; i += (i64)gridDim.x * RFX_BLOCK) {
        const i64 t = in[i];
        out[i] = t == from ? to : t;
    }
}
extern "C" int rfx_hip_replace_i64(rfx_ctx_t *c, const int64_t *d_col, int64_t nrows, int64_t from, int64_t to, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (nrows <= 0) return RFX_OK;
    RFX_REQUIRE(d_col && d_out, RFX_EINVAL, "device buffers expected");
    const i64 want = (nrows + RFX_BLOCK - 1) / RFX_BLOCK;
    hipLaunchKernelGGL(k_replace_i64, dim3((unsigned)(want < c->num_cus * 16 ? want : c->num_cus * 16)), dim3(RFX_BLOCK), 0, c->stream, (const i64 *)d_col, (i64)nrows, (i64)from, (i64)to,
                       (i64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ---- reproducible grouped f64 sums (round 6, opt-in: RFX_DETERMINISTIC / rfx_ops_set_deterministic) ----
// Grouped f64 sums reach their accumulators through LDS / device atomics in whatever order the waves run: the last bits differ from run to run.  Integer
// addition is associative: a column scaled by a power of two and rounded to i64 ONCE per cell sums to the same bits in any order.  Two small kernels make
// that column; the planner then runs an ordinary i64 SUM over it (rfx_ops_select.c: det_rewrite).
//   k_absmax_f64   max |x| over the column as ordered bits (|x| >= 0: the bit pattern of a non-negative double orders like an integer) + a flag for NaN / inf
//   k_fix_f64      out[i] = llrint(x[i] * 2^k)   (in place allowed)
__global__ __launch_bounds__(RFX_BLOCK) void k_absmax_f64(const u64 *__restrict__ in, i64 n, unsigned long long *__restrict__ out /* [0] max bits, [1] non-finite seen */) {
    u64 mx = 0, bad = 0;
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const u64 a = in[i] & 0x7FFFFFFFFFFFFFFFULL;
        if (a >= 0x7FF0000000000000ULL) bad = 1;
        else mx = a > mx ? a : mx;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const u64 o = rfx_shfl_xor_u64(mx, m);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&out[0], (unsigned long long)mx);
    if (bad) atomicOr(&out[1], 1ULL);
}
__global__ __launch_bounds__(RFX_BLOCK) void k_fix_f64(const double *in, i64 n, int k, i64 *out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) out[i] = (i64)__builtin_llrint(__builtin_ldexp(in[i], k));
}
extern "C" int rfx_hip_absmax_f64(rfx_ctx_t *c, const double *d_in, int64_t n, double *absmax, int *nonfinite) {
    RFX_REQUIRE(c && absmax && nonfinite, RFX_EINVAL, "NULL argument");
    *absmax = 0.0;
    *nonfinite = 0;
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in, RFX_EINVAL, "NULL argument");
    int rc = rfx_ws_reserve(c, 256);
    if (rc != RFX_OK) return rc;
    unsigned long long *o = (unsigned long long *)c->d_ws;
    RFX_HIP_CHECK(hipMemsetAsync(o, 0, 16, c->stream));
    hipLaunchKernelGGL(k_absmax_f64, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, (const u64 *)d_in, (i64)n, o);
    RFX_HIP_CHECK(hipGetLastError());
    unsigned long long *h = (unsigned long long *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(h, o, 16, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(absmax, &h[0], 8);
    *nonfinite = h[1] != 0;
    return RFX_OK;
}
// two limbs: y = x * 2^k (exact), h = rint(y) -- limb 0 -- and limb 1 = llrint((y - h) * 2^m): the part of the cell the first limb rounded away (y - h is exact)
__global__ __launch_bounds__(RFX_BLOCK) void k_fix_f64_low(const double *in, i64 n, int k, int m, i64 *out) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        const double y = __builtin_ldexp(in[i], k);
        out[i] = (i64)__builtin_llrint(__builtin_ldexp(y - __builtin_rint(y), m));
    }
}
extern "C" int rfx_hip_fix_f64_low(rfx_ctx_t *c, const double *d_in, int64_t n, int k, int m, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "NULL argument");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in && d_out && k > -1100 && k < 1100 && m >= 0 && m <= 62, RFX_EINVAL, "bad argument");
    hipLaunchKernelGGL(k_fix_f64_low, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, d_in, (i64)n, k, m, (i64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
// ... and back: a result column of integer sums (hi, optionally lo) as f64 in place -- (double)hi * sc + (double)lo * sc2 (powers of two: the products are
// exact), divided by the group's row count for an average.  No contraction (-ffp-contract=off): one rounding per operation, on every device alike.
__global__ __launch_bounds__(RFX_BLOCK) void k_unfix_f64(i64 *hi_io, const i64 *lo, const i64 *cnt, i64 n, double sc, double sc2) {
    for (i64 i = blockIdx.x * (i64)RFX_BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * RFX_BLOCK) {
        double v = (double)hi_io[i] * sc;
        if (lo) v = v + (double)lo[i] * sc2;
        if (cnt) v = cnt[i] ? v / (double)cnt[i] : rfx_as_f64(RFX_NAN_BITS);
        ((double *)hi_io)[i] = v;
    }
}
extern "C" int rfx_hip_unfix_f64(rfx_ctx_t *c, int64_t *d_hi_io, const int64_t *d_lo, const int64_t *d_cnt, int64_t n, double sc, double sc2) {
    RFX_REQUIRE(c, RFX_EINVAL, "NULL argument");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_hi_io, RFX_EINVAL, "NULL argument");
    const i64 blocks = (n + RFX_BLOCK - 1) / RFX_BLOCK;
    hipLaunchKernelGGL(k_unfix_f64, dim3((unsigned)(blocks < (i64)c->num_cus * 8 ? blocks : (i64)c->num_cus * 8)), dim3(RFX_BLOCK), 0, c->stream, (i64 *)d_hi_io, (const i64 *)d_lo,
                       (const i64 *)d_cnt, (i64)n, sc, sc2);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
extern "C" int rfx_hip_fix_f64(rfx_ctx_t *c, const double *d_in, int64_t n, int k, int64_t *d_out) {
    RFX_REQUIRE(c, RFX_EINVAL, "ctx is NULL");
    if (n <= 0) return RFX_OK;
    RFX_REQUIRE(d_in && d_out && k > -1100 && k < 1100, RFX_EINVAL, "bad argument");
    hipLaunchKernelGGL(k_fix_f64, dim3(c->num_cus * 8), dim3(RFX_BLOCK), 0, c->stream, d_in, (i64)n, k, (i64 *)d_out);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
