// rfx_group_common.hpp -- internal: cell rules shared by the dense (rfx_group.hip), partitioned
// (rfx_group_part.hip) and hashed (rfx_hash.hip) group-by paths.
#pragma once
#include "rfx_scalar_kernel.hpp"

#define RFX_CHUNK 512
int rfx_scan_counts(rfx_ctx *c, i64 *d_cnt, i64 n, i64 *d_total, const i64 *d_n_eff = nullptr); // d_n_eff: a device-side bound on n (entries beyond it are zero, untouched)
int rfx_rank_slots(rfx_ctx *c, const u64 *d_first, i64 slots, i64 row_base, i64 total_rows, i64 *ngroups);
int rfx_fill_u64(rfx_ctx *c, void *p, i64 n, u64 val);
// partitioned form of the sparse-key (hashed) group-by, rfx_group_part.hip.  RFX_ESTATE: not applicable.
int rfx_group_part_hash_accumulate(rfx_ctx *c, const struct Plan &P, int key_idx, const struct HashArgs &H, int *d_overflow);

#define RFX_U64_HASH_SEED 0x9ddfea08eb382d69ULL /* core/hash.h:35 */

// hash_index_u64 -- core/hash.h:86-97
__device__ __host__ __forceinline__ u64 rfx_hash_index_u64(u64 h, u64 k) {
    const u64 s = RFX_U64_HASH_SEED;
    u64 a = (h ^ k) * s;
    a ^= (a >> 47);
    u64 b = (((k << 31) | (k >> 33)) ^ a) * s;
    b ^= (b >> 47);
    b *= s;
    return b;
}

struct HashArgs {
    i64 capacity;
    int key_idx;
    int nagg;
    u64 *keys;
    u64 *first;
    u64 *acc[RFX_MAX_AGGS];
    u64 *cnt[RFX_MAX_AGGS];
};

// find-or-insert in the device-wide open-addressed table; returns the slot, or -1 when RFX_HASH_MAX_PROBES consecutive
// slots are taken by other keys (the caller reports "table full": capacity is meant to be >= 2x the distinct keys, where
// runs of that length do not occur).
#define RFX_HASH_MAX_PROBES 2048
__device__ __forceinline__ i64 hash_slot(u64 *keys, i64 capacity, u64 key) {
    if ((i64)key == RFX_NULL_I64_D) return capacity;
    const u64 mask = (u64)capacity - 1;
    u64 s = rfx_hash_index_u64(RFX_U64_HASH_SEED, key) & mask;
    const i64 bound = capacity < RFX_HASH_MAX_PROBES ? capacity : RFX_HASH_MAX_PROBES;
    for (i64 probe = 0; probe < bound; probe++) {
        u64 k = keys[s];
        if (k == key) return (i64)s;
        if ((i64)k == RFX_NULL_I64_D) {
            u64 old = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key);
            if ((i64)old == RFX_NULL_I64_D || old == key) return (i64)s;
        }
        s = (s + 1) & mask;
    }
    return -1; // table (locally) full
}

// the same, counting the slots this call claimed (the device-wide kernel stops a launch once the table is 3/4 full: long before
// probe runs reach RFX_HASH_MAX_PROBES, which is what made an undersized first attempt as slow as the final one)
__device__ __forceinline__ i64 hash_slot_ins(u64 *keys, i64 capacity, u64 key, unsigned &inserted) {
    if ((i64)key == RFX_NULL_I64_D) return capacity;
    const u64 mask = (u64)capacity - 1;
    u64 s = rfx_hash_index_u64(RFX_U64_HASH_SEED, key) & mask;
    const i64 bound = capacity < RFX_HASH_MAX_PROBES ? capacity : RFX_HASH_MAX_PROBES;
    for (i64 probe = 0; probe < bound; probe++) {
        u64 k = keys[s];
        if (k == key) return (i64)s;
        if ((i64)k == RFX_NULL_I64_D) {
            u64 old = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key);
            if ((i64)old == RFX_NULL_I64_D) {
                inserted++;
                return (i64)s;
            }
            if (old == key) return (i64)s;
        }
        s = (s + 1) & mask;
    }
    return -1;
}

struct GroupArgs {
    i64 kmin;
    i64 range;
    int key_idx; // column index of the key inside Plan::cols
    int nagg;
    // several key columns folded on the fly (index_group_list_perfect_partial, core/index.c:2238-2305): nkeys >= 2,
    // slot = sum_i (col[kidx[i]] - kmn[i]) * kmul[i]; kmin is 0 then
    int nkeys;
    int kidx[RFX_MAX_KEYS];
    u64 kmn[RFX_MAX_KEYS];
    u64 kmul[RFX_MAX_KEYS];
    int rep_shift; // TINY form: log2 of the number of lane-private table replicas
    u64 *first;
    u64 *acc[RFX_MAX_AGGS];
    u64 *cnt[RFX_MAX_AGGS];
    // a selected row whose key lies outside the agreed scope (possible when the scope was SAMPLED, rfx_hip_scope_sample_i64): counted
    // here instead of being dropped silently; several keys are checked column by column (krng: each column's range)
    unsigned *oob;
    u64 krng[RFX_MAX_KEYS];
};

#define RFX_FEW_MAX_GROUPS 8
#ifndef __HIPCC_RTC__
int rfx_rtc_group_few(rfx_ctx *c, const Plan &P, const GroupArgs &G, int grid); // rfx_rtc.hip; RFX_ESTATE: no run-time compiler / sources
#endif

// identity element of an accumulator cell
__device__ __host__ __forceinline__ u64 acc_identity(int kind, int f64) {
    (void)f64;
    if (kind == RFX_AGG_MIN) return (u64)RFX_INF_I64_D;
    if (kind == RFX_AGG_MAX) return (u64)RFX_NULL_I64_D;
    return 0ULL; // SUM/AVG/COUNT/FIRST: 0 (== +0.0)
}
__host__ __device__ __forceinline__ bool agg_has_cnt(int kind, int f64) {
    return kind == RFX_AGG_AVG || (kind == RFX_AGG_SUM && !f64);
}

// Apply one selected row to the tables.  Works on LDS or global cells (the compiler resolves the address space).
// Grouped rules (core/aggr.c): sum is null-STICKY (ADDI64/ADDF64, :1088-1092) -> i64: count nulls aside, f64: IEEE NaN
// propagates by itself; min/max skip nulls (:1152-1315); count counts every row (:1317-1453); avg casts to f64
// and skips nulls (:1455-1540).
// skip = 1: the aggregate's argument is an expression, which the reference folds group by group with its SCALAR rules
// (FOLD_ADD* skip nulls, core/ops.h:156-158): an f64 sum then skips NaN instead of letting it poison the group.
template <typename P64>
__device__ __forceinline__ void group_apply(P64 acc, P64 cnt, int kind, int f64, u64 x, int skip = 0) {
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) {
                if (!skip || !rfx_isnan_bits(x)) unsafeAtomicAdd((double *)acc, rfx_as_f64(x));
            }
            else if ((i64)x == RFX_NULL_I64_D) atomicAdd((unsigned long long *)cnt, 1ULL);
            else atomicAdd((unsigned long long *)acc, (unsigned long long)x);
            break;
        case RFX_AGG_AVG:
            if (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D)) {
                unsafeAtomicAdd((double *)acc, f64 ? rfx_as_f64(x) : (double)(i64)x);
                atomicAdd((unsigned long long *)cnt, 1ULL);
            }
            break;
        case RFX_AGG_MIN:
            if (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D)) atomicMin((long long *)acc, f64 ? rfx_f64_to_ord(x) : (i64)x);
            break;
        case RFX_AGG_MAX:
            if (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D)) atomicMax((long long *)acc, f64 ? rfx_f64_to_ord(x) : (i64)x);
            break;
        case RFX_AGG_COUNT:
            atomicAdd((unsigned long long *)acc, 1ULL);
            break;
        default: // FIRST is resolved at emit time from d_first
            break;
    }
}

// LDS form with a 32-bit count cell (compact tables of k_group_dense).
__device__ __forceinline__ void group_apply(u64 *acc, unsigned *cnt, int kind, int f64, u64 x, int skip = 0) {
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) {
                if (!skip || !rfx_isnan_bits(x)) unsafeAtomicAdd((double *)acc, rfx_as_f64(x));
            } else if ((i64)x == RFX_NULL_I64_D) atomicAdd(cnt, 1u);
            else atomicAdd((unsigned long long *)acc, (unsigned long long)x);
            break;
        case RFX_AGG_AVG:
            if (f64 ? !rfx_isnan_bits(x) : ((i64)x != RFX_NULL_I64_D)) {
                unsafeAtomicAdd((double *)acc, f64 ? rfx_as_f64(x) : (double)(i64)x);
                atomicAdd(cnt, 1u);
            }
            break;
        default:
            group_apply(acc, (u64 *)0, kind, f64, x, skip); // MIN / MAX / COUNT never touch the count cell
            break;
    }
}

// ... the same without atomics, for a cell only this workgroup writes during the launch
__device__ __forceinline__ void group_merge_cell_plain(u64 *gacc, u64 *gcnt, int kind, int f64, u64 a, u64 c) {
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) { if (a != 0ULL) *gacc = rfx_as_u64(rfx_as_f64(*gacc) + rfx_as_f64(a)); }
            else {
                if (a) *gacc += a;
                if (c) *gcnt += c;
            }
            break;
        case RFX_AGG_AVG:
            if (c) {
                *gacc = rfx_as_u64(rfx_as_f64(*gacc) + rfx_as_f64(a));
                *gcnt += c;
            }
            break;
        case RFX_AGG_MIN:
            if ((i64)a != RFX_INF_I64_D && (i64)a < (i64)*gacc) *gacc = a;
            break;
        case RFX_AGG_MAX:
            if ((i64)a != RFX_NULL_I64_D && (i64)a > (i64)*gacc) *gacc = a;
            break;
        case RFX_AGG_COUNT:
            if (a) *gacc += a;
            break;
        default:
            break;
    }
}

// Merge one LDS cell into the global tables.
__device__ __forceinline__ void group_merge_cell(u64 *gacc, u64 *gcnt, int kind, int f64, u64 a, u64 c) {
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) { if (a != 0ULL) unsafeAtomicAdd((double *)gacc, rfx_as_f64(a)); }
            else {
                if (a) atomicAdd((unsigned long long *)gacc, (unsigned long long)a);
                if (c) atomicAdd((unsigned long long *)gcnt, (unsigned long long)c);
            }
            break;
        case RFX_AGG_AVG:
            if (c) {
                unsafeAtomicAdd((double *)gacc, rfx_as_f64(a));
                atomicAdd((unsigned long long *)gcnt, (unsigned long long)c);
            }
            break;
        case RFX_AGG_MIN:
            if ((i64)a != RFX_INF_I64_D) atomicMin((long long *)gacc, (i64)a);
            break;
        case RFX_AGG_MAX:
            if ((i64)a != RFX_NULL_I64_D) atomicMax((long long *)gacc, (i64)a);
            break;
        case RFX_AGG_COUNT:
            if (a) atomicAdd((unsigned long long *)gacc, (unsigned long long)a);
            break;
        default:
            break;
    }
}

struct EmitArgs {
    i64 kmin;
    i64 slots;
    int nagg;
    int kinds[RFX_MAX_AGGS];
    int f64s[RFX_MAX_AGGS];
    int skips[RFX_MAX_AGGS]; // scalar (null-skipping) finalisation: expression aggregates
    const u64 *first;
    const u64 *keys; // hashed tables: explicit key per slot ; dense: NULL (key = kmin + slot)
    const u64 *acc[RFX_MAX_AGGS];
    const u64 *cnt[RFX_MAX_AGGS];
    const u64 *col[RFX_MAX_AGGS]; // FIRST: source column (local rows), may be NULL
    i64 row0;                     // FIRST: global id of col[0]
    i64 nloc;                     // FIRST: rows col[] holds (0 = unbounded): a first row another GPU owns reads as 0 here, the owner supplies it
    i64 *out_keys;
    i64 *out_first;
    u64 *out[RFX_MAX_AGGS];
    i64 g0, gn; // emit WINDOW (rfx_hip_ctx_emit_window): only the groups [g0, g0 + gn) are written, group g at out[g - g0] -- one slice of a result
                // whose other slices other devices emit (each reads its own back over its own PCIe link)
};

// Final grouped value of one cell -- core/aggr.c rules, see DESIGN.md "NULL semantics".
__device__ __forceinline__ u64 group_final(int kind, int f64, u64 a, u64 c, int skip = 0) {
    if (skip) { // scalar rules per group (core/math.c folds): sum of no value = 0, min / max of no value = null
        switch (kind) {
            case RFX_AGG_SUM: return a;
            case RFX_AGG_MIN:
                if ((i64)a == RFX_INF_I64_D) return f64 ? RFX_NAN_BITS : (u64)RFX_NULL_I64_D;
                return f64 ? rfx_ord_to_f64((i64)a) : a;
            case RFX_AGG_MAX:
                if ((i64)a == RFX_NULL_I64_D) return f64 ? RFX_NAN_BITS : a;
                return f64 ? rfx_ord_to_f64((i64)a) : a;
            default: break; // AVG: same rule either way
        }
    }
    switch (kind) {
        case RFX_AGG_SUM:
            if (f64) return rfx_isnan_bits(a) ? RFX_NAN_BITS : a;
            return c ? (u64)RFX_NULL_I64_D : a; // any null input poisons the group (ADDI64, core/ops.h:154)
        case RFX_AGG_AVG:
            return c ? rfx_as_u64(rfx_as_f64(a) / (double)(i64)c) : RFX_NAN_BITS; // core/aggr.c:2060
        case RFX_AGG_MIN:
            if (f64) return ((i64)a == RFX_INF_I64_D) ? RFX_PINF_BITS : rfx_ord_to_f64((i64)a); // all-null group -> +inf (core/aggr.c:1250)
            return a;                                                                             // all-null group -> INF_I64 (core/aggr.c:1246)
        case RFX_AGG_MAX:
            if (f64) return ((i64)a == RFX_NULL_I64_D) ? RFX_NAN_BITS : rfx_ord_to_f64((i64)a); // all-null group -> null
            return a;
        case RFX_AGG_COUNT:
            return a;
        default:
            return a;
    }
}


int rfx_emit_slots(rfx_ctx *c, const EmitArgs &A);
int rfx_rank_emit(rfx_ctx *c, const EmitArgs &A, i64 total_rows, int nsl, int si, i64 out_cap, i64 *ngroups); // rank -> emit, no round trip between (rfx_group.hip)
