// rfx_group_plane.hip -- one-pass radix partitioning into PLANES: the dense group-by whose tables do not fit one workgroup's LDS
// (BASELINE C3 / C3w: 1e9 rows, 1e6 i64 keys, sum(f64); the H2O shapes with two or three value columns), moved with 8 bytes per value
// + 4 bytes of meta per record instead of 16 / 32-byte records.
//
// What the reference does here: index_group_i64_scoped (core/index.c:2002-2092) walks the key column once, single-threaded, and
// AGGR_ITER (core/aggr.c:73-181) folds every value column through the group ids.  rfx_group_chunk.hip partitions 16-byte records
// {row:32 | key >> 8, value} (one value column only): 16 B read + 16 written + 16 read per row, with a scatter that spends a third of its
// time in workgroup barriers.  Measured on this part (tools/write_probe.hip, profiles/r03_write_probe.jsonl): a scatter keeps up with HBM
// only when every store instruction covers >= 128 contiguous bytes per stream with 8..16 bytes per lane (256 B: 4.9-5.2 TB/s of mixed
// traffic, 128 B: 4.7-4.9, 64 B: 3.7, 32 B: 3.1, 16 B: 2.5) -- so the record is split into planes that each leave in whole lines:
//
//   value plane(s)  8 B per record (one plane per distinct value column, up to three)
//   meta plane      4 B per record {slot-in-partition : 14 | row - block_base : 18}
//
//   k_plane_scatter   one 1024-lane workgroup per ROW BLOCK of 2^18 rows (thousands of blocks: the dispatcher balances them).  Waves run
//                     DECOUPLED, no workgroup barrier in the loop: a selected row takes the next place of its partition's LDS ring with
//                     one returning LDS atomic, writes its values + meta there, and counts itself into the ring half's arrival word; the
//                     lane that completes a half (a "group": 32 or 16 records) queues it, and its wave stores queued groups as whole lines
//                     (16 B per lane, non-temporal) before it goes on.  A (block, partition) region has a FIXED place and size (no
//                     allocator): too many records for a region (skew the sample did not show) or a stuck ring raise a flag and the caller
//                     takes the chunk / column paths.  Side product, as before: min / max / count of the selected keys (index_scope_i64).
//   k_plane_aggregate one LDS table set per partition (first row as 32 bits, accumulators 64, counts 32); every wave streams its own
//                     regions with the next batch's loads in flight; merged into the global tables.  Several value planes: one pass per
//                     set of aggregates whose tables fit a CU's LDS (each reads the meta plane + its own value planes).
//
// Bytes per row (C3): 16 read + 12 written + 12 read = 40 (was 48).  Both kernels turned out INSTRUCTION-bound, not byte-bound (rocprofv3
// SQ counters, profiles/r03_pmc_plane.txt: ~200 wave instructions per 64 rows in the scatter): a 2-byte meta plane for the rows that cannot
// be a group's first one (36 B/row) was built, measured at the same 7.6 ms per 1e9 rows, and taken out again (git d52b3d2).
#include "rfx_part_common.hpp"
#include <stdlib.h>

#define PL_T 1024
#define PL_WAVES (PL_T / RFX_WAVE)
#define PL_WROWS 512         /* rows per wave step: 8 per lane, four 16-byte loads per lane and column */
#define PL_BLOCK_ROWS (1 << 18)
#define PL_DELTA_BITS 18
#define PL_SLOT_BITS 14
#define PL_QCAP 320          /* per-wave queue of completed groups */
#define PL_QFLUSH 24         /* ... stored once this many are queued (and at the end of every step) */
#define PL_SPIN_LIMIT (1u << 14)
#define PL_MAX_NV 3
#ifndef PL_META64_PLAIN
#define PL_META64_PLAIN 1 /* ordinary (not non-temporal) stores for the 64-byte meta groups of 16-record groups too: A/B at 1e9 rows, 1e6 keys -- `sum v1, avg v3`
                           * 13.85 -> 12.34 ms, `sum v1,v2,v3` 18.46 -> 17.84, `avg v1,v2,v3` 19.20 -> 18.75 (round 5; half and quarter lines written non-temporally
                           * bypass the L2 that would have merged them with their neighbours) */
#endif

struct PlaneArgs {
    int nblk;          // row blocks
    int nv;            // value planes
    unsigned pmask;    // plan columns the predicates read (loaded first and whole; the others only where a row is selected)
    i64 block_rows;
    unsigned c0;       // records per (block, partition) region, a multiple of 32
    u64 *vals[PL_MAX_NV];
    unsigned *meta;    // region (b, p) starts at entry ((b << PBITS) + p) * c0 of every plane
    unsigned *cnt;     // [nblk << PBITS] records per region
    unsigned *ctl;     // 256 bytes, zeroed before the launch.  [1]: some region overflowed / a ring did not drain -> the caller falls back;
                       // bytes 64..87: the scope of all selected keys {~image(min), image(max), count} (see the kernel's end)
};
struct PlSh {
    unsigned dead, _pad[3];
    i64 red[3][PL_WAVES];
};
// A group is 2^VGL records (32: two 128-byte lines per value plane and one of meta; 16: one line and half a line), a partition's ring two
// groups per plane.
// PARTITIONS: 2^PBITS -- except the sparse-key form <PBITS 8, VGL 4, HK>: 192 partitions by multiply-shift of the hash (round 4).  192 x 16-record
// rings of {key, value, meta} fill 125 KB of LDS, and 1e6 keys / 192 = 5 208 keys of 20 bytes fit ONE workgroup's LDS table in the aggregate
// pass: every record is streamed once there, where 128 partitions needed two workgroups per partition that each streamed all of it
// (42.8 of the 88 GB the query moved).
template <int PBITS, int VGL, bool HK>
__host__ __device__ constexpr int pl_parts() { return (HK && PBITS == 8 && VGL == 4) ? 192 : (1 << PBITS); }
template <int NV, int PBITS, int VGL, bool HK = false>
__host__ __device__ constexpr size_t pl_lds_bytes() {
    return (size_t)pl_parts<PBITS, VGL, HK>() * ((size_t)(2 << VGL) * (NV * 8 + 4) + 8 + 4) + (size_t)PL_WAVES * PL_QCAP * 4 + sizeof(PlSh) + 64;
}
typedef u64 pl_v2 __attribute__((ext_vector_type(2)));
typedef unsigned pl_m2 __attribute__((ext_vector_type(2)));
// LDS words other waves write: relaxed workgroup-scope atomics keep the LDS address space (a volatile cast turns them into FLAT
// accesses that wait for every global load in flight)
#define PL_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define PL_ST(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// HK: sparse keys -- the partition is the top PBITS bits of hash_index_u64(key) and the meta word carries the next 14 bits of the hash
// (where the aggregate pass starts probing its LDS table) instead of the slot inside the partition; plane 0 is then the key column itself.
template <int NC, int NP, int NV, int PBITS, int VGL, bool HK = false>
__global__ __launch_bounds__(PL_T) void k_plane_scatter(const Plan P, const PlaneArgs A) {
    constexpr int PARTS = pl_parts<PBITS, VGL, HK>();
    constexpr bool P192 = PARTS != (1 << PBITS); // multiply-shift partitions (not a power of two): regions at (block * PARTS + p)
    constexpr unsigned VG = 1u << VGL, RING = 2u << VGL; // records per group / per ring
    // KVI (round 5, sparse keys over 256 partitions): 8-record groups are 64-byte pieces per plane -- half lines, which cost the scatter 13.3 ms
    // where 128-byte lines cost 7.6 (profiles/r03_write_probe.jsonl: 64 B 3.7 TB/s, 128 B 4.7).  The key group and the value group of the SAME
    // eight records are therefore laid side by side in ONE plane of 16-byte-per-record regions: [keys of group g: 64 B][values of group g: 64 B],
    // stored by eight neighbouring lanes of one instruction as one whole 128-byte line.
    constexpr bool KVI = HK && NV == 2 && PBITS == 8 && VGL == 3;
    constexpr unsigned LPD = VG / 2;                     // lanes that store one value group (16 bytes = two records each); meta: half of them
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    u64 *vring = (u64 *)pl_smem;                                       // [NV][PARTS][RING]
    unsigned *mring = (unsigned *)(vring + (size_t)NV * PARTS * RING); // [PARTS][RING]
    unsigned *words = mring + (size_t)PARTS * RING;                    // [PARTS][2] ring halves: generation << 8 | arrivals
    unsigned *tail = words + PARTS * 2;                                // [PARTS] records handed out
    unsigned *fq = tail + PARTS;                                       // [PL_WAVES][PL_QCAP]
    PlSh &L = *(PlSh *)(fq + PL_WAVES * PL_QCAP);
    PredSet<NP> S;
    predset_load<NP>(P, S);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned *myq = fq + wv * PL_QCAP;
    const i64 r0 = (i64)blockIdx.x * A.block_rows;
    const i64 r1 = (r0 + A.block_rows < P.nrows) ? r0 + A.block_rows : P.nrows;
    const u64 region = ((u64)blockIdx.x * (u64)PARTS) * A.c0;
    for (int i = tid; i < PARTS; i += PL_T) {
        tail[i] = 0;
        words[2 * i] = 0;
        words[2 * i + 1] = 0;
    }
    if (tid == 0) L.dead = 0;
    __syncthreads();
    i64 mn = RFX_INF_I64_D, mx = RFX_NULL_I64_D; // over ALL selected keys: a null key (INT64_MIN) shows as the minimum, as for index_scope_i64
    unsigned nsel = 0;                           // selected rows of this lane
    int qn = 0;                                  // wave-uniform

    // completed groups of this wave -> global memory as whole lines, then their ring halves are released
    auto flush_queue = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        for (int i0 = 0; i0 < qn; i0 += 64 / (int)LPD) {
            const int idx = i0 + (lane / (int)LPD);
            if (idx < qn) {
                const unsigned d = myq[idx];
                const unsigned p = d & 0xFFu, gi = (d >> 8) & 0x3FFFFFu, pl = d >> 30, sub = lane & (LPD - 1u);
                if (pl < 3) {
                    const pl_v2 x = *(const pl_v2 *)(vring + ((size_t)pl * PARTS + p) * RING + (gi & 1u) * VG + sub * 2);
                    if constexpr (KVI) {
                        __builtin_nontemporal_store(x, (pl_v2 *)(A.vals[0] + 2 * (region + (u64)p * A.c0 + (u64)gi * VG) + (u64)pl * VG + sub * 2));
                    } else {
                        u64 *dst = (NV == 1 ? A.vals[0] : (pl == 0 ? A.vals[0] : (pl == 1 ? A.vals[1] : A.vals[2])));
                        __builtin_nontemporal_store(x, (pl_v2 *)(dst + region + (u64)p * A.c0 + (u64)gi * VG + sub * 2));
                    }
                } else if (sub < LPD / 2) {
                    const pl_v2 x = *(const pl_v2 *)(mring + (size_t)p * RING + (gi & 1u) * VG + sub * 4);
                    // (KVI: a meta group is 32 bytes -- a quarter line; non-temporal stores of that size are the slowest thing the write probe found
                    //  (1.5 TB/s), ordinary ones meet their neighbours in L2)
                    if constexpr (KVI || (VGL == 4 && PL_META64_PLAIN)) *(pl_v2 *)(A.meta + region + (u64)p * A.c0 + (u64)gi * VG + sub * 4) = x;
                    else __builtin_nontemporal_store(x, (pl_v2 *)(A.meta + region + (u64)p * A.c0 + (u64)gi * VG + sub * 4));
                }
            }
        }
        asm volatile("" ::: "memory"); // ring reads above, releases below
        for (int i = lane; i < qn; i += 64) {
            const unsigned d = myq[i];
            if ((d >> 30) == 3u) atomicAdd(&words[2 * (d & 0xFFu) + ((d >> 8) & 1u)], 256u - VG); // VG arrivals -> 0, generation + 1
        }
        qn = 0;
    };

    // one selected row per lane (or none): takes its place in its partition's ring, writes, counts itself in, queues completed groups
    auto place = [&](const bool on, const u64 key, const u64 (&val)[NV], const unsigned delta) __attribute__((always_inline)) {
        const u64 kh = HK ? rfx_hash_index_u64(RFX_U64_HASH_SEED, key) : key;
        const unsigned p = P192 ? (unsigned)(((kh >> 32) * (u64)PARTS) >> 32) : (HK ? (unsigned)(kh >> (64 - PBITS)) : ((unsigned)key & (unsigned)(PARTS - 1)));
        unsigned seq = 0;
        if (on) {
            const i64 k = (i64)key;
            mn = k < mn ? k : mn;
            mx = k > mx ? k : mx;
            seq = atomicAdd(&tail[p], 1u);
            if (seq >= A.c0) PL_ST(&L.dead, 1u); // the region is full (skew / selectivity the sample did not show): the caller falls back
        }
        const unsigned g = seq >> VGL;
        unsigned *w = &words[2 * p + (g & 1u)];
        bool todo = on && seq < A.c0;
        // A record goes into its ring place once the group that used the place before has left (generation check); almost always at
        // once.  When not, the lanes that CAN write do so first (the awaited group may be waiting for exactly them), the wave stores what
        // it owes (nobody may wait while holding completed groups), and only then polls.
        for (unsigned it = 0;; it++) {
            bool ready = false;
            if (todo) ready = (PL_LD(w) >> 8) == (g >> 1);
            bool comp = false;
            if (ready) {
                const unsigned at = p * RING + (seq & (RING - 1u));
#pragma unroll
                for (int j = 0; j < NV; j++) vring[(size_t)j * PARTS * RING + at] = val[j];
                mring[at] = ((unsigned)(P192 ? (kh >> 18) : (HK ? (kh >> (64 - PBITS - PL_SLOT_BITS)) : (key >> PBITS))) & ((1u << PL_SLOT_BITS) - 1u)) | (delta << PL_SLOT_BITS);
                asm volatile("" ::: "memory"); // the record is in the ring before it is counted (LDS executes a wave's operations in order)
                comp = (atomicAdd(w, 1u) & 0xFFu) == VG - 1u;
                todo = false;
            }
            const u64 cb = __ballot(comp);
            if (cb) {
                if (comp) {
                    const unsigned at = (unsigned)qn + (unsigned)(NV + 1) * __builtin_amdgcn_mbcnt_hi((unsigned)(cb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cb, 0u));
#pragma unroll
                    for (int j = 0; j < NV; j++) myq[at + j] = p | (g << 8) | ((unsigned)j << 30);
                    myq[at + NV] = p | (g << 8) | (3u << 30); // the meta plane last: its descriptor also releases the ring half
                }
                qn += (NV + 1) * __popcll(cb);
            }
            if (!__any(todo)) break;
            flush_queue();
            if (PL_LD(&L.dead) || it > PL_SPIN_LIMIT) {
                PL_ST(&L.dead, 1u);
                break;
            }
            if (it) __builtin_amdgcn_s_sleep(4);
        }
        if (qn >= PL_QFLUSH) flush_queue();
    };

    bool alive = true;
    // `m`: the selected rows of this lane (bit e = row e): every valid row without a filter, else what the predicates keep
    auto step = [&](const u64 (&v)[NC][8], const unsigned m, const i64 sbase) __attribute__((always_inline)) {
        const unsigned dbase = (unsigned)(sbase - r0);
        const unsigned valid = m;
        if constexpr (NP == 0) {
            nsel += (unsigned)__popc(valid);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                u64 val[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) val[j] = v[1 + j][e]; // plane j = plan column 1 + j
                place((valid >> e) & 1u, v[0][e], val, dbase + (unsigned)((e >> 1) * 128 + (e & 1)));
            }
        } else {
            // Under a filter a lane keeps few of its eight rows (10 %: none 43 %, one 38 %, two 15 %): instead of eight passes over mostly
            // idle lanes, every pass takes each lane's NEXT selected row -- as many passes as the busiest lane has rows (three or four).
            unsigned rem = m;
            nsel += (unsigned)__popc(rem);
            // (the rows as opaque register values: left as elements of the tile, the select chain below is folded back into ONE load with a
            //  per-lane index, and a tile beyond 128 bytes -- two or three value planes -- then lives in scratch memory: 208-272 bytes per
            //  lane written and gathered back every step)
            u64 t[1 + NV][8];
#pragma unroll
            for (int c = 0; c < 1 + NV; c++) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    t[c][q] = v[c][q];
                    if constexpr (NV > 1) asm volatile("" : "+v"(t[c][q]));
                }
            }
            while (__any(rem != 0u)) {
                const bool on = rem != 0u;
                const unsigned e = (unsigned)__builtin_ctz(rem | 0x100u);
                rem &= rem - 1u;
                u64 key = t[0][0], val[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) val[j] = t[1 + j][0];
#pragma unroll
                for (int q = 1; q < 8; q++) {
                    const bool is = e == (unsigned)q;
                    key = is ? t[0][q] : key;
#pragma unroll
                    for (int j = 0; j < NV; j++) val[j] = is ? t[1 + j][q] : val[j];
                }
                place(on, key, val, dbase + (e >> 1) * 128u + (e & 1u));
            }
        }
        flush_queue();
        if (PL_LD(&L.dead)) alive = false;
    };

    const i64 nsteps = (r1 - r0 + PL_WROWS - 1) / PL_WROWS; // of this block
    const i64 nfull = (r1 - r0) / PL_WROWS;
    {
        u64 va[NC][8];
        for (i64 s = wv; s < nfull && alive; s += PL_WAVES) {
            const i64 base = r0 + s * PL_WROWS + lane * 2;
            if constexpr (NP == 0) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u64x2 t = rfx_ld2(P.cols[c] + base + j * 128);
                        va[c][2 * j] = t.x;
                        va[c][2 * j + 1] = t.y;
                    }
                }
                step(va, 0xffu, base);
            } else {
                // Under a filter: FIRST the columns the predicates read (A.pmask), whole; the selection; THEN the other columns, and of those
                // only the 16-byte pairs that hold a selected row.  At 10 % selectivity a 64-byte piece of the key / value columns holds a
                // selected row with probability 1 - 0.9^8 = 57 %: the rest is never fetched (C3w: 24 -> ~17 B/row of HBM reads).
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if ((A.pmask >> c) & 1u) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const u64x2 t = rfx_ld2(P.cols[c] + base + j * 128);
                            va[c][2 * j] = t.x;
                            va[c][2 * j + 1] = t.y;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) va[c][e] = 0;
                    }
                }
                const unsigned m = eval_preds<NC, 8, NP>(S, va, 0xffu);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (!((A.pmask >> c) & 1u)) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if ((m >> (2 * j)) & 3u) {
                                const u64x2 t = rfx_ld2(P.cols[c] + base + j * 128);
                                va[c][2 * j] = t.x;
                                va[c][2 * j + 1] = t.y;
                            }
                        }
                    }
                }
                step(va, m, base);
            }
        }
    }
    if (nfull < nsteps && (nfull % PL_WAVES) == wv && alive) { // the ragged last step of the block
        u64 va[NC][8];
        const i64 base = r0 + nfull * PL_WROWS + lane * 2;
        unsigned valid = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const i64 row = base + (e >> 1) * 128 + (e & 1);
            const bool in = row < r1;
            valid |= (unsigned)in << e;
#pragma unroll
            for (int c = 0; c < NC; c++) va[c][e] = in ? P.cols[c][row] : 0ULL;
        }
        step(va, (NP == 0) ? valid : eval_preds<NC, 8, NP>(S, va, valid), base);
    }
    __syncthreads();
    if (L.dead) {
        if (tid == 0) atomicExch(&A.ctl[1], 1u);
        return;
    }
    // the last, partly filled group of every plane of every partition (the region has room: c0 is a multiple of the group; the record count
    // ends before the padding)
    for (int i = tid; i < PARTS * (NV + 1) * (int)LPD; i += PL_T) {
        const unsigned sub = (unsigned)i & (LPD - 1u), p = ((unsigned)i / LPD) % (unsigned)PARTS, pl = ((unsigned)i / LPD) / (unsigned)PARTS;
        const unsigned t = tail[p];
        if (!(t & (VG - 1u))) continue;
        const unsigned gi = t >> VGL;
        if (pl < (unsigned)NV) {
            const pl_v2 x = *(const pl_v2 *)(vring + ((size_t)pl * PARTS + p) * RING + (gi & 1u) * VG + sub * 2);
            if constexpr (KVI) {
                *(pl_v2 *)(A.vals[0] + 2 * (region + (u64)p * A.c0 + (u64)gi * VG) + (u64)pl * VG + sub * 2) = x;
            } else {
                u64 *dst = (NV == 1 ? A.vals[0] : (pl == 0 ? A.vals[0] : (pl == 1 ? A.vals[1] : A.vals[2])));
                *(pl_v2 *)(dst + region + (u64)p * A.c0 + (u64)gi * VG + sub * 2) = x;
            }
        } else if (sub < LPD / 2) {
            const pl_v2 x = *(const pl_v2 *)(mring + (size_t)p * RING + (gi & 1u) * VG + sub * 4);
            *(pl_v2 *)(A.meta + region + (u64)p * A.c0 + (u64)gi * VG + sub * 4) = x;
        }
    }
    for (int i = tid; i < PARTS; i += PL_T) A.cnt[(size_t)blockIdx.x * PARTS + i] = tail[i];
    i64 sel = (i64)nsel;
    for (int s2 = 32; s2 >= 1; s2 >>= 1) {
        const i64 omn = (i64)rfx_shfl_xor_u64((u64)mn, s2), omx = (i64)rfx_shfl_xor_u64((u64)mx, s2);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
        sel += (i64)rfx_shfl_xor_u64((u64)sel, s2);
    }
    if (lane == 0) {
        L.red[0][wv] = mn;
        L.red[1][wv] = mx;
        L.red[2][wv] = sel;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w2 = 1; w2 < PL_WAVES; w2++) {
            mn = L.red[0][w2] < mn ? L.red[0][w2] : mn;
            mx = L.red[1][w2] > mx ? L.red[1][w2] : mx;
            sel += L.red[2][w2];
        }
        // the whole launch's scope, folded with three atomics per block into cells that start at zero: the minimum as the maximum of the
        // inverted order-preserving image (0 = +inf), the maximum as the maximum of the image (0 = INT64_MIN = "no key")
        unsigned long long *g = (unsigned long long *)A.ctl + 8;
        if (sel > 0) {
            atomicMax(&g[0], ~((unsigned long long)mn ^ 0x8000000000000000ULL));
            atomicMax(&g[1], (unsigned long long)mx ^ 0x8000000000000000ULL);
            atomicAdd(&g[2], (unsigned long long)sel);
        }
    }
}

// ---- per-partition LDS aggregation over the regions of one partition ----
struct PlaneAggArgs {
    i64 kmin, range;
    i64 local; // table cells per partition: slots congruent to one residue mod 2^pbits
    int split; // workgroups per partition
    int excl;  // split == 1: this workgroup alone writes its partition's slots during the launch -- the write-out needs no atomics
    int first_read; // A/B (RFX_PL_FIRST_READ=1): rounds 3-5's first-row update -- read the word, compare, atomic only when smaller -- instead of one no-return ds_min_u32
    int nblk, pbits;
    int agg_pl[RFX_MAX_AGGS]; // loaded plane of aggregate a (-1: none: COUNT / FIRST)
    i64 block_rows;
    unsigned c0;
    const u64 *vals[PL_MAX_NV]; // the planes this pass loads
    const unsigned *meta;
    const unsigned *cnt;
    u64 *first;
    u64 *acc[RFX_MAX_AGGS];
    u64 *cntt[RFX_MAX_AGGS];
};
#define PL_FIRST_NONE 0xFFFFFFFFu

// LDS: [nagg accumulators u64 x local][first u32 x local][counts u32 x local each][count windows 64 x u32 per wave]
// FAST: exactly one aggregate, a plain f64 sum (C3 / C3w): first row + one ds_add_f64 per record, no per-record dispatch on the aggregate kinds
template <int THREADS, int NVL, bool FAST>
__global__ __launch_bounds__(THREADS) void k_plane_aggregate(const Plan P, const PlaneAggArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 pl_agg_smem[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x / A.split, s = blockIdx.x % A.split;
    const i64 local = A.local;
    int kind[RFX_MAX_AGGS], f64[RFX_MAX_AGGS], skip[RFX_MAX_AGGS], cnt_of[RFX_MAX_AGGS], apl[RFX_MAX_AGGS];
    int ncnt = 0;
#pragma unroll
    for (int a = 0; a < RFX_MAX_AGGS; a++) {
        kind[a] = (a < P.nagg) ? P.aggs[a].kind : -1;
        f64[a] = (a < P.nagg) ? P.aggs[a].f64 : 0;
        skip[a] = (a < P.nagg) ? P.aggs[a].skipnull : 0;
        apl[a] = (a < P.nagg) ? A.agg_pl[a] : -1;
        cnt_of[a] = -1;
        if (kind[a] >= 0 && agg_has_cnt(kind[a], f64[a])) cnt_of[a] = ncnt++;
    }
    u64 *accs = pl_agg_smem;                                   // [nagg][local]
    unsigned *first = (unsigned *)(accs + (i64)P.nagg * local); // [local]
    unsigned *cnts = first + local;                            // [ncnt][local]
    for (i64 i = tid; i < local; i += THREADS) first[i] = PL_FIRST_NONE;
#pragma unroll
    for (int a = 0; a < RFX_MAX_AGGS; a++) {
        if (kind[a] < 0) continue;
        const u64 id = acc_identity(kind[a], f64[a]);
        for (i64 i = tid; i < local; i += THREADS) accs[(i64)a * local + i] = id;
        if (cnt_of[a] >= 0) {
            for (i64 i = tid; i < local; i += THREADS) cnts[(i64)cnt_of[a] * local + i] = 0u;
        }
    }
    __syncthreads();
    const unsigned shift = (unsigned)(((i64)p - A.kmin) >> A.pbits); // slot = (key - kmin) >> pbits = (key >> pbits) + floor((p - kmin) / 2^pbits), modular
    auto apply = [&](unsigned mm, i64 rbase, const u64 (&x)[NVL]) __attribute__((always_inline)) {
        const unsigned mask = (1u << PL_SLOT_BITS) - 1u;
        const unsigned slot = ((mm & mask) + shift) & mask;
        if ((i64)slot >= local) return; // a key outside the scope the tables were sized for: not ours
        const unsigned row = (unsigned)(rbase + (i64)(mm >> PL_SLOT_BITS));
        // (round 6 A/B, profiles/r06_first_read.txt: one no-return ds_min_u32 per record instead of read + compare + rare atomic changes nothing here --
        //  C3 7.42 against 7.28 ms, c3w 4.38 against 4.37 -- the read-first form stays)
        if (A.first_read) {
            if (row < first[slot]) atomicMin(&first[slot], row);
        } else __hip_atomic_fetch_min(&first[slot], row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (FAST) {
            unsafeAtomicAdd((double *)&accs[slot], rfx_as_f64(x[0]));
            return;
        }
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            if (kind[a] < 0) continue;
            u64 xv = 0;
#pragma unroll
            for (int j = 0; j < NVL; j++)
                if (apl[a] == j) xv = x[j];
            group_apply(&accs[(i64)a * local + slot], cnt_of[a] >= 0 ? &cnts[(i64)cnt_of[a] * local + slot] : (unsigned *)0, kind[a], f64[a], xv, skip[a]);
        }
    };
    // Every WAVE streams its own regions (blocks q, q + Q, ... of this partition): a region holds a few hundred to a few thousand
    // records, a 64-lane wave keeps its lanes busy on that where a whole workgroup would not.  One "batch" = up to two record pairs per
    // lane (256 records) of one region; the next batch's loads are issued before this one is applied.
    const int lane = tid & 63;
    constexpr int NW = THREADS / 64;
    const int Q = A.split * NW;
    struct Batch {
        pl_v2 val[NVL][2];
        pl_m2 m[2];
        unsigned n, i0;
        int b;
    };
    // Loads are UNCONDITIONAL: a fixed number per call whatever the region's size (a pair past the region's end is clamped to its last
    // allocated pair and ignored).  With a load inside a condition the compiler cannot count what is in flight and waits with vmcnt(0):
    // consuming one batch would wait for the other batch's loads too.
    auto load = [&](bool live, int b, unsigned i0, unsigned n, Batch &B) __attribute__((always_inline)) {
        B.b = live ? b : 0;
        B.i0 = live ? i0 : 0u;
        B.n = live ? n : 0u;
        const u64 base = (((u64)B.b << A.pbits) + (u64)p) * A.c0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            unsigned i = B.i0 + ((unsigned)k * 64u + (unsigned)lane) * 2u;
            i = i < A.c0 - 2u ? i : A.c0 - 2u;
#pragma unroll
            for (int j = 0; j < NVL; j++) B.val[j][k] = __builtin_nontemporal_load((const pl_v2 *)(A.vals[j] + base + i));
            B.m[k] = __builtin_nontemporal_load((const pl_m2 *)(A.meta + base + i));
        }
    };
    auto consume = [&](const Batch &B) __attribute__((always_inline)) {
        const i64 rbase = (i64)B.b * A.block_rows;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned i = B.i0 + ((unsigned)k * 64u + (unsigned)lane) * 2u;
            if (i < B.n) {
                u64 x[NVL];
#pragma unroll
                for (int j = 0; j < NVL; j++) x[j] = B.val[j][k].x;
                apply(B.m[k].x, rbase, x);
                if (i + 1 < B.n) {
#pragma unroll
                    for (int j = 0; j < NVL; j++) x[j] = B.val[j][k].y;
                    apply(B.m[k].y, rbase, x);
                }
            }
        }
    };
    // Cursor over (region, batch within the region) -- wave-uniform.  This wave's regions are blocks q, q + Q, ...; their record counts come
    // 64 at a time through a wave-private LDS window (one vector load + ds_write per 64 regions, waited for on the spot -- rare; the count
    // of the next region is then an LDS read).  A global load per region whose result is needed at once would be the YOUNGEST load in
    // flight: waiting for it waits for every batch load before it.
    const int q = __builtin_amdgcn_readfirstlane(s * NW + (tid >> 6));
    const int nreg = q < A.nblk ? (A.nblk - q + Q - 1) / Q : 0; // regions of this wave
    unsigned *wcnt = cnts + (i64)ncnt * local + (tid >> 6) * 64; // [NW][64] behind the tables
    int r = -1;
    int b = q;
    unsigned i0 = 0, n = 0;
    auto advance = [&]() __attribute__((always_inline)) { // to the next non-empty batch position; false at the end
        for (;;) {
            if (r >= 0 && i0 < n) return true;
            r++;
            if (r >= nreg) return false;
            if ((r & 63) == 0) {
                const int r0 = r + lane;
                wcnt[lane] = (r0 < nreg) ? A.cnt[((size_t)(q + r0 * Q) << A.pbits) + p] : 0u;
            }
            n = (unsigned)__builtin_amdgcn_readfirstlane((int)wcnt[r & 63]);
            b = q + r * Q;
            i0 = 0;
        }
    };
    // (Round 4 tried FOUR batches in flight per wave -- under a selective filter a region is one batch and a wave walks ~120 of them, the
    // suspicion was one region per memory round trip.  C3w: 0.435 -> 0.424 ms, i.e. not it, and the general (non-FAST) form spilled:
    // `avg v1, v2, v3` at 1e6 keys 18.9 -> 32.1 ms.  Two it stays.  Nor is it the final merge below (a strided read and two device atomics
    // per cell from two workgroups per partition): the tables written out as they are, whole lines, and folded into the device-wide tables
    // by a transposing kernel of 31 us -- built, the whole suite green, C3w 4.74 against 4.70 ms with the atomics, C3 7.57 against 7.62:
    // nothing, taken out again (profiles/r04_plane_combine_ab.txt).)
    Batch B0, B1;
    bool h0 = advance();
    load(h0, b, i0, n, B0);
    i0 += 256u;
    while (h0) {
        const bool h1 = advance();
        load(h1, b, i0, n, B1);
        i0 += 256u;
        consume(B0);
        if (!h1) break;
        h0 = advance();
        load(h0, b, i0, n, B0);
        i0 += 256u;
        consume(B1);
    }
    __syncthreads();
    for (i64 i = tid; i < local; i += THREADS) {
        const unsigned f = first[i];
        if (f == PL_FIRST_NONE) continue;
        const i64 g = (i << A.pbits) | (i64)(((u64)p - (u64)A.kmin) & (u64)((1 << A.pbits) - 1));
        if (g >= A.range) continue;
        const u64 fr = (u64)P.row0 + (u64)f;
        if (A.excl) { // (round 5) nobody else touches slot g while this launch runs: plain read-modify-write instead of two to four device atomics a slot
            if (fr < A.first[g]) A.first[g] = fr;
#pragma unroll
            for (int a = 0; a < RFX_MAX_AGGS; a++) {
                if (kind[a] < 0) continue;
                const bool hc = cnt_of[a] >= 0;
                group_merge_cell_plain(&A.acc[a][g], hc ? &A.cntt[a][g] : (u64 *)0, kind[a], f64[a], accs[(i64)a * local + i], hc ? (u64)cnts[(i64)cnt_of[a] * local + i] : 0ULL);
            }
            continue;
        }
        if (fr < A.first[g]) atomicMin((unsigned long long *)&A.first[g], (unsigned long long)fr);
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            if (kind[a] < 0) continue;
            const bool hc = cnt_of[a] >= 0;
            group_merge_cell(&A.acc[a][g], hc ? &A.cntt[a][g] : (u64 *)0, kind[a], f64[a], accs[(i64)a * local + i], hc ? (u64)cnts[(i64)cnt_of[a] * local + i] : 0ULL);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int rfx_chunk_reserve(rfx_ctx *c, size_t bytes);
static size_t pl_al256(size_t x) { return (x + 255) & ~(size_t)255; }

// scratch block: [ctl 256 B][region counts][meta plane][value planes]
static void plane_layout(rfx_ctx *c, int nblk, int pbits, unsigned c0, int nv, PlaneArgs *A, size_t *total, int parts = 0) {
    const size_t regions = parts ? (size_t)nblk * (size_t)parts : (size_t)nblk << pbits;
    const size_t o_cnt = 256;
    const size_t o_meta = o_cnt + pl_al256(regions * 4), o_val = o_meta + pl_al256(regions * c0 * 4);
    const size_t plane = pl_al256(regions * c0 * 8);
    if (total) *total = o_val + plane * nv;
    char *w = (char *)c->d_chunk;
    A->nblk = nblk;
    A->nv = nv;
    A->c0 = c0;
    A->ctl = (unsigned *)w;
    A->cnt = (unsigned *)(w + o_cnt);
    A->meta = (unsigned *)(w + o_meta);
    for (int j = 0; j < nv; j++) A->vals[j] = (u64 *)(w + o_val + plane * j);
}

template <int NC, int NP, int NV, int PBITS, int VGL, bool HK = false>
static int launch_plane_scatter_inst(rfx_ctx *c, const Plan &P, const PlaneArgs &A) {
    constexpr size_t lds = pl_lds_bytes<NV, PBITS, VGL, HK>();
    static_assert(lds <= 160 * 1024, "rings beyond a CU's LDS");
    static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */ // per instantiation
    if (!((attr_set >> (c->device & 63)) & 1ull)) {
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_scatter<NC, NP, NV, PBITS, VGL, HK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);
    }
    hipLaunchKernelGGL((k_plane_scatter<NC, NP, NV, PBITS, VGL, HK>), dim3(A.nblk), dim3(PL_T), lds, c->stream, P, A);
    return RFX_OK;
}
template <int NC, int NV, int PBITS, int VGL, bool HK = false>
static int launch_plane_scatter_np(rfx_ctx *c, const Plan &P, const PlaneArgs &A) {
    if (P.npred == 0) return launch_plane_scatter_inst<NC, 0, NV, PBITS, VGL, HK>(c, P, A);
    if (P.npred == 1) return launch_plane_scatter_inst<NC, 1, NV, PBITS, VGL, HK>(c, P, A);
    if (P.npred <= 3) return launch_plane_scatter_inst<NC, 3, NV, PBITS, VGL, HK>(c, P, A);
    return launch_plane_scatter_inst<NC, RFX_MAX_PREDS, NV, PBITS, VGL, HK>(c, P, A);
}
// One value plane: 128 partitions with 32-record groups (256-byte value lines, 128 bytes of meta) when a partition twice as wide still
// fits the aggregate pass's LDS, else 256 partitions with 16-record groups.  Two and three planes: 128 partitions, 16-record groups (the
// rings of every plane share the CU's LDS).
static int launch_plane_scatter(rfx_ctx *c, const Plan &P, const PlaneArgs &A, int pbits) {
    const int extra = P.ncols - 1 - A.nv; // predicate columns beside the key and the value planes
    if (extra < 0 || extra > 2) return RFX_ESTATE;
    if (A.nv == 1) {
        if (pbits == 7) return extra == 0 ? launch_plane_scatter_np<2, 1, 7, 5>(c, P, A) : (extra == 1 ? launch_plane_scatter_np<3, 1, 7, 5>(c, P, A) : launch_plane_scatter_np<4, 1, 7, 5>(c, P, A));
        return extra == 0 ? launch_plane_scatter_np<2, 1, 8, 4>(c, P, A) : (extra == 1 ? launch_plane_scatter_np<3, 1, 8, 4>(c, P, A) : launch_plane_scatter_np<4, 1, 8, 4>(c, P, A));
    }
    if (pbits != 7 || extra > 1) return RFX_ESTATE;
    if (A.nv == 2) return extra == 0 ? launch_plane_scatter_np<3, 2, 7, 4>(c, P, A) : launch_plane_scatter_np<4, 2, 7, 4>(c, P, A);
    return extra == 0 ? launch_plane_scatter_np<4, 3, 7, 4>(c, P, A) : launch_plane_scatter_np<5, 3, 7, 4>(c, P, A);
}

// the value planes of a plan: distinct plain columns the aggregates read (COUNT / FIRST read none).  -1: expressions / more than PL_MAX_NV
static int plane_value_cols(const Plan &P, int *vcol, int *agg_plane) {
    int nv = 0;
    for (int a = 0; a < P.nagg; a++) {
        const PlanAgg &ag = P.aggs[a];
        agg_plane[a] = -1;
        if (ag.kind == RFX_AGG_COUNT || ag.kind == RFX_AGG_FIRST || ag.col < 0) continue;
        if (ag.col >= RFX_XCOL) return -1;
        int j = 0;
        for (; j < nv; j++)
            if (vcol[j] == ag.col) break;
        if (j == nv) {
            if (nv >= PL_MAX_NV) return -1;
            vcol[nv++] = ag.col;
        }
        agg_plane[a] = j;
    }
    return nv;
}
// LDS bytes of the aggregate pass for the aggregates [a0, a1)
static size_t plane_agg_lds(const Plan &P, int a0, int a1, i64 local) {
    size_t cells8 = 0, cells4 = 1;
    for (int a = a0; a < a1; a++) {
        cells8++;
        if (agg_has_cnt(P.aggs[a].kind, P.aggs[a].f64)) cells4++;
    }
    return (size_t)local * (cells8 * 8 + cells4 * 4) + 16 * 64 * 4 + 64; // tables + the waves' count windows
}
#define PL_AGG_LDS_MAX (150 * 1024)
// can every aggregate be served by SOME pass?  (each alone must fit; passes then take as many as fit)
static bool plane_agg_fits(const Plan &P, i64 local) {
    for (int a = 0; a < P.nagg; a++)
        if (plane_agg_lds(P, a, a + 1, local) > PL_AGG_LDS_MAX) return false;
    return P.nagg >= 1;
}

struct PlaneState { // what rfx_plane_scope leaves for rfx_plane_accumulate (lives in the context: ext_p[2])
    int valid, npred, logic, nblk, pbits, nv;
    unsigned c0;
    const void *key;
    const void *val[PL_MAX_NV];
    i64 nrows;
    i64 seen; // records in the planes (the selected rows)
    u64 sig[RFX_MAX_PREDS][6];
};
static PlaneState *plane_state(rfx_ctx *c) {
    if (!c->ext_p[2]) c->ext_p[2] = calloc(1, sizeof(PlaneState));
    return (PlaneState *)c->ext_p[2];
}
void rfx_plane_invalidate(rfx_ctx *c) {
    if (c->ext_p[2]) ((PlaneState *)c->ext_p[2])->valid = 0;
}
void rfx_plane_release(rfx_ctx *c) {
    free(c->ext_p[2]);
    c->ext_p[2] = NULL;
}
static void plane_pred_sig(const Plan &P, u64 (*sig)[6]) {
    for (int i = 0; i < P.npred; i++) {
        const PlanPred &q = P.preds[i];
        sig[i][0] = (u64)(uintptr_t)P.cols[q.col];
        sig[i][1] = q.rhs_col >= 0 ? (u64)(uintptr_t)P.cols[q.rhs_col] : 0;
        sig[i][2] = (u64)q.op;
        sig[i][3] = (u64)(q.dom_f64 | (q.lhs_cvt << 1) | (q.rhs_cvt << 2) | (q.more << 3) | ((u64)q.tree << 4));
        sig[i][4] = q.rhs_bits;
        sig[i][5] = 0;
    }
}

// Scope pass that also partitions into planes.  P: the plan as built (key at key_idx); est_range / frac: the sample's guesses.
// RFX_ESTATE: not applicable / gave up (nothing is left behind), the caller goes on with the chunk kernels or the plain scope pass.
int rfx_plane_scope(rfx_ctx *c, const Plan &P, int key_idx, const void *d_key, int npred, int logic, unsigned long long est_range, double frac,
                    i64 *kmin, i64 *kmax, i64 *seen) {
    PlaneState *st = plane_state(c);
    if (!st) return RFX_ESTATE;
    st->valid = 0;
    if (c->flags & RFX_TUNE_NO_PLANE) return RFX_ESTATE;
    int vcol[PL_MAX_NV], agg_plane[RFX_MAX_AGGS];
    const int nv = plane_value_cols(P, vcol, agg_plane);
    if (nv < 1 || P.nx > 0) return RFX_ESTATE;
    // 128 partitions when a partition twice as wide still fits the aggregate pass's LDS with a sixteenth to spare (the sampled range can
    // only be too small; a range that turns out wider costs the scatter and falls back), else -- one plane only -- 256
    int pbits = 7;
    {
        static const char *force = getenv("RFX_PLANE_PBITS"); // (A/B)
        const i64 l7 = (i64)((est_range + 127) >> 7);
        if (l7 + l7 / 16 > (1 << PL_SLOT_BITS) || !plane_agg_fits(P, l7 + l7 / 16)) pbits = 8;
        if (force && nv == 1) pbits = atoi(force) == 7 ? 7 : 8;
    }
    if (pbits == 8 && nv > 1) return RFX_ESTATE;
    const i64 est_local = (i64)((est_range + (1ULL << pbits) - 1) >> pbits);
    if (est_local > (1 << PL_SLOT_BITS) || !plane_agg_fits(P, est_local)) return RFX_ESTATE;
    const i64 nrows = P.nrows;
    const i64 nblk64 = (nrows + PL_BLOCK_ROWS - 1) / PL_BLOCK_ROWS;
    if (nblk64 > (1 << 20) || nrows >= 0xFFFFFFF0LL) return RFX_ESTATE;
    // region size: the expected share of a partition plus room for its spread (uniform keys: sigma = sqrt(share)), whole groups
    double share = (double)PL_BLOCK_ROWS / (double)(1 << pbits);
    if (npred > 0) share *= (frac * 1.3 + 0.01 < 1.0 ? frac * 1.3 + 0.01 : 1.0);
    unsigned c0 = (unsigned)(share * 1.25 + 160.0);
    c0 = (c0 + 127u) & ~127u;
    PlaneArgs A;
    memset(&A, 0, sizeof(A));
    size_t need = 0;
    plane_layout(c, (int)nblk64, pbits, c0, nv, &A, &need);
    if (rfx_chunk_reserve(c, need) != RFX_OK) return RFX_ESTATE;
    plane_layout(c, (int)nblk64, pbits, c0, nv, &A, NULL);
    A.block_rows = PL_BLOCK_ROWS;
    // key -> column 0, value plane j -> column 1 + j (the kernel reads them without a run-time select; a value column that IS the key
    // column is listed a second time), the predicates' other columns behind them
    Plan Pc = P;
    {
        int perm[RFX_MAX_COLS + PL_MAX_NV], inv[RFX_MAX_COLS], n2 = 0;
        for (int i = 0; i < P.ncols; i++) inv[i] = -1;
        perm[n2++] = key_idx;
        inv[key_idx] = 0;
        for (int j = 0; j < nv; j++) {
            if (inv[vcol[j]] < 0) inv[vcol[j]] = n2;
            perm[n2++] = vcol[j];
        }
        for (int i = 0; i < P.ncols; i++)
            if (inv[i] < 0) {
                inv[i] = n2;
                perm[n2++] = i;
            }
        if (n2 > 1 + nv + 2 || n2 > RFX_MAX_COLS) return RFX_ESTATE;
        Pc.ncols = n2;
        for (int i = 0; i < n2; i++) Pc.cols[i] = P.cols[perm[i]];
        for (int i = 0; i < P.npred; i++) {
            Pc.preds[i].col = inv[P.preds[i].col];
            if (P.preds[i].rhs_col >= 0) Pc.preds[i].rhs_col = inv[P.preds[i].rhs_col];
        }
    }
    for (int i = 0; i < Pc.npred; i++) {
        A.pmask |= 1u << Pc.preds[i].col;
        if (Pc.preds[i].rhs_col >= 0) A.pmask |= 1u << Pc.preds[i].rhs_col;
    }
    if (getenv("RFX_PLANE_LOAD_ALL")) A.pmask = 0xFFu; // (A/B: every column whole, as before)
    RFX_HIP_CHECK(hipMemsetAsync(A.ctl, 0, 256, c->stream));
    c->ext_i[3 + RFX_STAT_PLANE_SCATTER]++;
    RFX_KERNEL_BEGIN(c);
    int rc = launch_plane_scatter(c, Pc, A, pbits);
    RFX_KERNEL_END(c);
    if (rc != RFX_OK) return rc;
    RFX_HIP_CHECK(hipGetLastError());
    unsigned *hctl = (unsigned *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(hctl, A.ctl, 256, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (hctl[1]) { // a region overflowed (skew or selectivity the sample did not show): the chunk kernels take over
        c->ext_i[3 + RFX_STAT_PLANE_FALLBACK]++;
        return RFX_ESTATE;
    }
    const u64 *hs = (const u64 *)hctl + 8;
    *seen = (i64)hs[2];
    plane_state(c)->seen = *seen;
    *kmin = (i64)(~hs[0] ^ 0x8000000000000000ULL);
    *kmax = (i64)(hs[1] ^ 0x8000000000000000ULL);
    if (*seen > 0 && *kmin == RFX_NULL_I64_D) return RFX_ESTATE; // a null key among the selected rows: the exact scope pass counts them
    if (*seen == 0) return RFX_OK;
    st->valid = 1;
    st->key = d_key;
    for (int j = 0; j < nv; j++) st->val[j] = (const void *)P.cols[vcol[j]];
    st->nrows = nrows;
    st->npred = npred;
    st->logic = logic;
    st->nblk = (int)nblk64;
    st->pbits = pbits;
    st->nv = nv;
    st->c0 = c0;
    plane_pred_sig(P, st->sig);
    return RFX_OK;
}

template <int THREADS, int NVL, bool FAST>
static int launch_plane_aggregate_inst(rfx_ctx *c, const Plan &P, const PlaneAggArgs &G, int grid, size_t lds, int lds_max) {
    static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */ // per instantiation
    if (!((attr_set >> (c->device & 63)) & 1ull)) {
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_aggregate<THREADS, NVL, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);
    }
    hipLaunchKernelGGL((k_plane_aggregate<THREADS, NVL, FAST>), dim3(grid), dim3(THREADS), lds, c->stream, P, G);
    return RFX_OK;
}
template <int THREADS>
static int launch_plane_aggregate(rfx_ctx *c, const Plan &P, const PlaneAggArgs &G, int nvl, bool fast, int grid, size_t lds, int lds_max) {
    if (nvl <= 1) return fast ? launch_plane_aggregate_inst<THREADS, 1, true>(c, P, G, grid, lds, lds_max) : launch_plane_aggregate_inst<THREADS, 1, false>(c, P, G, grid, lds, lds_max);
    if (nvl == 2) return launch_plane_aggregate_inst<THREADS, 2, false>(c, P, G, grid, lds, lds_max);
    return launch_plane_aggregate_inst<THREADS, 3, false>(c, P, G, grid, lds, lds_max);
}

// Pass 2 over the planes rfx_plane_scope left, if they are the planes of exactly this plan.  RFX_ESTATE: not so.
int rfx_plane_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const rfx_group_tables_t *t) {
    PlaneState *st = (PlaneState *)c->ext_p[2];
    if (!st || !st->valid) return RFX_ESTATE;
    int vcol[PL_MAX_NV], agg_plane[RFX_MAX_AGGS];
    bool ok = st->key == (const void *)P.cols[key_idx] && st->nrows == P.nrows && st->npred == P.npred && st->logic == P.logic && P.nx == 0;
    const int nv = ok ? plane_value_cols(P, vcol, agg_plane) : -1;
    ok = ok && nv == st->nv;
    for (int j = 0; ok && j < nv; j++) ok = st->val[j] == (const void *)P.cols[vcol[j]];
    if (ok && P.npred > 0) {
        u64 sig[RFX_MAX_PREDS][6];
        plane_pred_sig(P, sig);
        ok = memcmp(sig, st->sig, sizeof(u64) * 6 * (size_t)P.npred) == 0;
    }
    st->valid = 0; // consumed (or stale) either way
    if (!ok || t->range <= (1 << st->pbits)) return RFX_ESTATE;
    const i64 local = (t->range + (1 << st->pbits) - 1) >> st->pbits;
    if (local > (1 << PL_SLOT_BITS) || !plane_agg_fits(P, local)) return RFX_ESTATE;
    PlaneArgs A;
    memset(&A, 0, sizeof(A));
    plane_layout(c, st->nblk, st->pbits, st->c0, st->nv, &A, NULL);
    const int nparts = 1 << st->pbits;
    RFX_KERNEL_BEGIN(c);
    int rc = RFX_OK;
    // one pass per run of aggregates whose tables fit the LDS together
    for (int a0 = 0; a0 < P.nagg && rc == RFX_OK;) {
        int a1 = a0 + 1;
        while (a1 < P.nagg && plane_agg_lds(P, a0, a1 + 1, local) <= PL_AGG_LDS_MAX) a1++;
        Plan Ps = P;
        Ps.nagg = a1 - a0;
        PlaneAggArgs G;
        memset(&G, 0, sizeof(G));
        G.kmin = t->kmin;
        G.range = t->range;
        G.local = local;
        {
            static const char *fr = getenv("RFX_PL_FIRST_READ"); // (0: one no-return ds_min_u32 per record instead -- measured: C3 7.42 against 7.28 ms, c3w 4.38 / 4.37: no gain here)
            G.first_read = fr ? atoi(fr) : 1;
        }
        G.nblk = st->nblk;
        G.pbits = st->pbits;
        G.block_rows = PL_BLOCK_ROWS;
        G.c0 = st->c0;
        G.meta = A.meta;
        G.cnt = A.cnt;
        G.first = (u64 *)t->d_first;
        int used[PL_MAX_NV] = {-1, -1, -1}, nvl = 0; // scatter planes this pass loads
        for (int a = a0; a < a1; a++) {
            Ps.aggs[a - a0] = P.aggs[a];
            G.acc[a - a0] = (u64 *)t->d_acc[a];
            G.cntt[a - a0] = (u64 *)t->d_cnt[a];
            G.agg_pl[a - a0] = -1;
            if (agg_plane[a] < 0) continue;
            int j = 0;
            for (; j < nvl; j++)
                if (used[j] == agg_plane[a]) break;
            if (j == nvl) {
                used[nvl] = agg_plane[a];
                G.vals[nvl] = A.vals[agg_plane[a]];
                nvl++;
            }
            G.agg_pl[a - a0] = j;
        }
        for (int a = a1 - a0; a < RFX_MAX_AGGS; a++) {
            Ps.aggs[a].kind = -1;
            G.agg_pl[a] = -1;
        }
        if (nvl == 0) { // COUNT / FIRST only: any plane keeps the loads uniform
            G.vals[0] = A.vals[0];
            nvl = 1;
        }
        const size_t lds = plane_agg_lds(P, a0, a1, local);
        const bool fast = Ps.nagg == 1 && Ps.aggs[0].kind == RFX_AGG_SUM && Ps.aggs[0].f64 && !Ps.aggs[0].skipnull && G.agg_pl[0] == 0;
        c->ext_i[3 + RFX_STAT_PLANE_AGGREGATE]++;
        if (lds > 52 * 1024) {
            // big tables: one 1024-lane workgroup per CU
            G.split = (c->num_cus + nparts - 1) / nparts;
            if (G.split < 1) G.split = 1;
            // A SHORT pass (round 5): two workgroups per partition each fold their LDS table into the device tables by atomics -- 4e6 of them for
            // 1e6 slots, 0.16 ms whatever the record count (the per-device pass of an 8-way split, a 1e8-row table: more than the records cost).
            // One workgroup per partition owns its slots for the launch and writes them plainly; its record loop takes twice as long, which
            // pays below ~4e7 records.  RFX_PLANE_AGG_SPLIT=<n> forces the split (A/B).
            {
                static const char *fs = getenv("RFX_PLANE_AGG_SPLIT");
                if (fs && atoi(fs) >= 1) G.split = atoi(fs);
                else if (st->seen > 0 && st->seen <= 40000000) G.split = 1;
            }
            G.excl = G.split == 1;
            rc = launch_plane_aggregate<1024>(c, Ps, G, nvl, fast, nparts * G.split, lds, 160 * 1024);
        } else {
            const int per_cu = (int)((156 * 1024) / lds) < 4 ? (int)((156 * 1024) / lds) : 4; // 512-lane workgroups a CU can hold: 3 at 51 KB
            G.split = (per_cu * c->num_cus + nparts - 1) / nparts;
            if (G.split < 1) G.split = 1;
            rc = launch_plane_aggregate<512>(c, Ps, G, nvl, fast, nparts * G.split, lds, 64 * 1024);
        }
        a0 = a1;
    }
    RFX_KERNEL_END(c);
    if (rc != RFX_OK) return rc;
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}

// ------------------------------------------------------------------------------------------------
// Sparse keys (K9) through the same planes: partition = top 7 bits of hash_index_u64(key), planes {key 8 B, value 8 B, meta 4 B
// {14 more hash bits | row - block base}}; every partition is then aggregated in LDS OPEN-ADDRESSED tables.
//
// The reference: index_group_distribute (core/index.c:1777-1911) -- one ht_oa table per CPU chunk (core/hash.c:35-148), merged one after
// the other.  Round 1's device form (rfx_group_part.hip): histogram pass -> exact-offset scatter of {row, key, value} as three
// tile-sorted planes (15 ms per 1e9 rows) -> one LDS hash table per partition (10 ms): 29 ms, 86 GB moved for 16 algorithmic.  Here
// the scatter is k_plane_scatter itself (no histogram pass, barrier-free rings, whole lines), and because 128 partitions of 1e6 keys
// hold 7 800 keys each -- more than one CU's LDS takes at 20 bytes an entry -- every partition is aggregated by `halves` workgroups
// that all stream the partition's records and keep those whose next hash bits are theirs.
//   k_plane_hash_aggregate  one 1024-lane workgroup per (partition, half): LDS table {key 64, first row 32, accumulators 64, counts 32}
//                           probed from (hash bits x capacity) >> bits, ds_cmpst insert; a key that finds no room (or the null key) goes
//                           to the caller's device-wide table directly; at the end one insert per distinct key into that table.
// ------------------------------------------------------------------------------------------------
struct PlaneHashArgs {
    HashArgs H;
    int nblk, pbits, hbits; // hbits: log2(workgroups per partition)
    int parts;              // partitions (2^pbits, or 192): region (b, p) sits at entry (b * parts + p) * c0
    i64 block_rows;
    unsigned c0;
    unsigned lcap; // LDS table entries
    int narr;      // 8-byte arrays per entry beside the key (accumulators, wide counts)
    int agg_pl[RFX_MAX_AGGS]; // 1: the value plane, -1: none (COUNT / FIRST)
    const u64 *keys, *vals;
    const unsigned *meta, *cnt;
    int *overflow;
    int dbg; // RFX_PLH_DBG (ablation, WRONG answers): 1 no first-row update, 2 + no probing (the start slot is taken), 3 loads only
    int kvi; // the key and value planes are ONE plane of interleaved 8-record groups (k_plane_scatter's KVI form: 256 partitions)
};
#define PLH_T 1024
#define PLH_PROBES 512

// FAST: exactly one aggregate, a plain f64 sum over the value plane (the K9 shape): no per-record dispatch on the aggregate kinds
// VAR 0: a lane's four records settle their slots one after the other (slot-by-slot linear probing), the spill path inline.  VAR 1: the same with the spill
// path as ONE cold loop behind the batch (the kernel's ISA: 24 800 -> ~10 000 lines).  VAR 2: the key table as buckets of four keys (below).  The first-row
// update is one no-return ds_min_u32 per record (RFX_PLH_DBG=5: round 5's read-first form).  Measured and withdrawn: LOCKSTEP probing (every round reads the
// four records' candidate slots back to back and settles them together: max-over-records rounds instead of their sum) -- 23.5 ms per query against 17.4: the
// unconditional reads of settled records and the per-round ballots cost more than the shorter dependency chain saves (git 37867ef has the kernel).
template <bool FAST, int VAR>
__global__ __launch_bounds__(PLH_T) void k_plane_hash_aggregate(const Plan P, const PlaneHashArgs X) {
    extern __shared__ __attribute__((aligned(16))) u64 plh_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = blockIdx.x >> X.hbits, half = blockIdx.x & ((1 << X.hbits) - 1);
    const unsigned C = X.lcap;
    u64 *lkey = plh_smem;                                                       // [C]
    u64 *larr = plh_smem + C;                                                   // [narr][C]
    unsigned *lfirst = (unsigned *)(plh_smem + (size_t)(1 + X.narr) * C);       // [C] local row of the first occurrence
    unsigned *wcnt = lfirst + C;                                                // [16][64] region counts, a window per wave
    int kind[RFX_MAX_AGGS], f64[RFX_MAX_AGGS], apl[RFX_MAX_AGGS], arr_of[RFX_MAX_AGGS], skip[RFX_MAX_AGGS];
    {
        int arr = 0;
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            kind[a] = (a < P.nagg) ? P.aggs[a].kind : -1;
            f64[a] = (a < P.nagg) ? P.aggs[a].f64 : 0;
            skip[a] = (a < P.nagg) ? P.aggs[a].skipnull : 0;
            apl[a] = (a < P.nagg) ? X.agg_pl[a] : -1;
            arr_of[a] = arr;
            if (kind[a] >= 0) arr += agg_has_cnt(kind[a], f64[a]) ? 2 : 1;
        }
    }
    for (unsigned i = tid; i < C; i += PLH_T) {
        lkey[i] = (u64)RFX_NULL_I64_D;
        lfirst[i] = 0xffffffffu;
    }
#pragma unroll
    for (int a = 0; a < RFX_MAX_AGGS; a++) {
        if (kind[a] < 0) continue;
        const u64 id = acc_identity(kind[a], f64[a]);
        for (unsigned i = tid; i < C; i += PLH_T) larr[(size_t)arr_of[a] * C + i] = id;
        if (agg_has_cnt(kind[a], f64[a]))
            for (unsigned i = tid; i < C; i += PLH_T) larr[(size_t)(arr_of[a] + 1) * C + i] = 0;
    }
    __syncthreads();
    const unsigned sbits = PL_SLOT_BITS - (unsigned)X.hbits; // hash bits left for the place in the table
    // where a record starts probing: from the meta word's hash bits
    auto start_of = [&](const unsigned mm) __attribute__((always_inline)) {
        const unsigned hb = mm & ((1u << PL_SLOT_BITS) - 1u);
        return (unsigned)(((u64)(hb & ((1u << sbits) - 1u)) * (u64)C) >> sbits);
    };
    auto mine_of = [&](const unsigned mm) __attribute__((always_inline)) { return (int)((mm & ((1u << PL_SLOT_BITS) - 1u)) >> sbits) == half; };
    // the slot of `key` (find or insert), the FIRST probe already read: k0 = lkey[s] as it stood a moment ago (a slot never changes once it
    // holds a key; an empty one is claimed by ds_cmpst, whose answer says who got it).  -1: no room / the null key
    auto slot_of = [&](const u64 key, unsigned s, u64 k0) __attribute__((always_inline)) {
        if ((i64)key == RFX_NULL_I64_D) return -1;
        for (int probe = 0; probe < PLH_PROBES; probe++) {
            const u64 k = probe ? lkey[s] : k0;
            if (k == key) return (int)s;
            if ((i64)k == RFX_NULL_I64_D) {
                const u64 old = atomicCAS((unsigned long long *)&lkey[s], (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key);
                if ((i64)old == RFX_NULL_I64_D || old == key) return (int)s;
            }
            s = (s + 1 == C) ? 0 : s + 1;
        }
        return -1;
    };
    auto fold = [&](const int idx, const unsigned lrow, const unsigned f0, const u64 key, const u64 val) __attribute__((always_inline)) {
        if (idx >= 0) {
            if (X.dbg == 5) { // (A/B: round 5's form -- the word read first, the atomic only when the row is smaller)
                if (lrow < f0) atomicMin(&lfirst[idx], lrow);
            } else if (X.dbg != 1) __hip_atomic_fetch_min(&lfirst[idx], lrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // one no-return ds_min_u32 per record
            if constexpr (FAST) {
                unsafeAtomicAdd((double *)&larr[idx], rfx_as_f64(val));
                return;
            }
#pragma unroll
            for (int a = 0; a < RFX_MAX_AGGS; a++) {
                if (kind[a] < 0) continue;
                group_apply(&larr[(size_t)arr_of[a] * C + idx], &larr[(size_t)(arr_of[a] + 1) * C + idx], kind[a], f64[a], apl[a] == 1 ? val : 0ULL, skip[a]);
            }
        } else { // no room in LDS for this key (or the null key): straight to the device-wide table
            const i64 g = hash_slot(X.H.keys, X.H.capacity, key);
            if (g < 0) {
                atomicExch(X.overflow, 1);
                return;
            }
            const u64 row = (u64)P.row0 + lrow;
            if (row < X.H.first[g]) atomicMin((unsigned long long *)&X.H.first[g], (unsigned long long)row);
#pragma unroll
            for (int a = 0; a < RFX_MAX_AGGS; a++) {
                if (kind[a] < 0) continue;
                group_apply(&X.H.acc[a][g], X.H.cnt[a] ? &X.H.cnt[a][g] : (u64 *)0, kind[a], f64[a], apl[a] == 1 ? val : 0ULL, skip[a]);
            }
        }
    };
    // every WAVE streams its own regions (blocks q, q + 16, ...) of the partition, two record pairs per lane and plane per batch, the
    // next batch's loads in flight while this one is applied (the structure of k_plane_aggregate; unconditional, clamped loads)
    constexpr int NW = PLH_T / 64;
    struct Batch {
        pl_v2 key[2], val[2];
        pl_m2 m[2];
        unsigned n, i0;
        int b;
    };
    auto load = [&](bool live, int b, unsigned i0, unsigned n, Batch &B) __attribute__((always_inline)) {
        B.b = live ? b : 0;
        B.i0 = live ? i0 : 0u;
        B.n = live ? n : 0u;
        const u64 base = ((u64)B.b * (u64)X.parts + (u64)p) * X.c0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            unsigned i = B.i0 + ((unsigned)k * 64u + (unsigned)lane) * 2u;
            i = i < X.c0 - 2u ? i : X.c0 - 2u;
            // (one load each, the address computed: two alternative loads under a uniform condition would be issued both, with a vmcnt(0) between)
            const u64 ko = X.kvi ? 2 * (base + (u64)(i & ~7u)) + (u64)(i & 7u) : base + (u64)i;
            const u64 *vb = X.kvi ? X.keys : X.vals;
            const u64 vo = X.kvi ? ko + 8 : base + (u64)i;
            B.key[k] = __builtin_nontemporal_load((const pl_v2 *)(X.keys + ko));
            B.val[k] = __builtin_nontemporal_load((const pl_v2 *)(vb + vo));
            B.m[k] = __builtin_nontemporal_load((const pl_m2 *)(X.meta + base + i));
        }
    };
    // A lane's FOUR records of a batch go through the table together (round 5): the four first probes are read back to back, then the four
    // slots are settled (the rare longer chains and the inserts one by one), then the four first-row words are read back to back, then the
    // folds -- four LDS round trips in flight where the record-by-record form waited for each in turn (the pass is bound by exactly that
    // latency: with the probes switched off it runs at the speed of its loads, tools/k9_ablate.py).
    auto consume = [&](const Batch &B) __attribute__((always_inline)) {
        const i64 rbase = (i64)B.b * X.block_rows;
        unsigned mm[4], st[4], lrow[4], f0[4];
        u64 key[4], val[4], k0[4];
        bool on[4];
        int idx[4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned i = B.i0 + ((unsigned)k * 64u + (unsigned)lane) * 2u;
            mm[2 * k] = B.m[k].x, mm[2 * k + 1] = B.m[k].y;
            key[2 * k] = B.key[k].x, key[2 * k + 1] = B.key[k].y;
            val[2 * k] = B.val[k].x, val[2 * k + 1] = B.val[k].y;
            on[2 * k] = i < B.n && mine_of(mm[2 * k]);
            on[2 * k + 1] = i + 1 < B.n && mine_of(mm[2 * k + 1]);
        }
        if (X.dbg == 2 || X.dbg == 3) { // ablations (WRONG answers): 3 = the loads only, 2 = + one f64 add at the start slot, no probing
#pragma unroll
            for (int j = 0; j < 4; j++) {
                asm volatile("" ::"v"(key[j]), "v"(val[j]), "v"(mm[j]));
                if (X.dbg == 2 && on[j]) unsafeAtomicAdd((double *)&larr[start_of(mm[j])], rfx_as_f64(val[j]));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            st[j] = start_of(mm[j]);
            k0[j] = lkey[st[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) idx[j] = on[j] ? slot_of(key[j], st[j], k0[j]) : -1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lrow[j] = (unsigned)(rbase + (i64)(mm[j] >> PL_SLOT_BITS));
            f0[j] = X.dbg == 5 ? lfirst[idx[j] >= 0 ? idx[j] : 0] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (on[j]) fold(idx[j], lrow[j], f0[j], key[j], val[j]);
    };
    // VAR 1: the same probing, the no-room / null-key records in ONE cold loop behind the batch instead of inline between the LDS atomics of their
    // neighbours (the kernel's ISA: 24 800 -> ~10 000 lines)
    auto consume1 = [&](const Batch &B) __attribute__((always_inline)) {
        const i64 rbase = (i64)B.b * X.block_rows;
        unsigned mm[4], st[4], lrow[4];
        u64 key[4], val[4], k0[4];
        bool on[4], cold[4];
        int idx[4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned i = B.i0 + ((unsigned)k * 64u + (unsigned)lane) * 2u;
            mm[2 * k] = B.m[k].x, mm[2 * k + 1] = B.m[k].y;
            key[2 * k] = B.key[k].x, key[2 * k + 1] = B.key[k].y;
            val[2 * k] = B.val[k].x, val[2 * k + 1] = B.val[k].y;
            on[2 * k] = i < B.n && mine_of(mm[2 * k]);
            on[2 * k + 1] = i + 1 < B.n && mine_of(mm[2 * k + 1]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            st[j] = start_of(mm[j]);
            k0[j] = lkey[st[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) idx[j] = on[j] ? slot_of(key[j], st[j], k0[j]) : -1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lrow[j] = (unsigned)(rbase + (i64)(mm[j] >> PL_SLOT_BITS));
            cold[j] = on[j] && idx[j] < 0;
            if (!on[j] || idx[j] < 0) continue;
            __hip_atomic_fetch_min(&lfirst[idx[j]], lrow[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (FAST) unsafeAtomicAdd((double *)&larr[idx[j]], rfx_as_f64(val[j]));
            else {
#pragma unroll
                for (int a = 0; a < RFX_MAX_AGGS; a++) {
                    if (kind[a] < 0) continue;
                    group_apply(&larr[(size_t)arr_of[a] * C + idx[j]], &larr[(size_t)(arr_of[a] + 1) * C + idx[j]], kind[a], f64[a], apl[a] == 1 ? val[j] : 0ULL, skip[a]);
                }
            }
        }
        while (__builtin_expect(__builtin_amdgcn_ballot_w64(cold[0] | cold[1] | cold[2] | cold[3]) != 0, 0)) {
            const bool any = cold[0] | cold[1] | cold[2] | cold[3];
            const int pick = cold[0] ? 0 : cold[1] ? 1 : cold[2] ? 2 : 3;
            const u64 ck = pick == 0 ? key[0] : pick == 1 ? key[1] : pick == 2 ? key[2] : key[3];
            const u64 cv = pick == 0 ? val[0] : pick == 1 ? val[1] : pick == 2 ? val[2] : val[3];
            const unsigned cr = pick == 0 ? lrow[0] : pick == 1 ? lrow[1] : pick == 2 ? lrow[2] : lrow[3];
            cold[0] = cold[0] && pick != 0;
            cold[1] = cold[1] && pick != 1;
            cold[2] = cold[2] && pick != 2;
            cold[3] = cold[3] && pick != 3;
            if (any) fold(-1, cr, 0u, ck, cv);
        }
    };
    // VAR 2 (round 6): the key table as BUCKETS of four keys -- one lookup reads a whole bucket (two ds_read_b128) and compares its four keys at once; a full
    // bucket sends the key to the next one.  At load 0.53 a bucket of four overflows with probability ~6 %, so a batch settles in 2-3 rounds where slot-by-slot
    // linear probing needs the LONGEST of 64 lanes' chains (5-6 rounds of read -> wait -> compare -> branch per record: the 3.5 ms the probing costs on top of
    // the loads, tools/k9_ablate.py).  Same arrays, same slot numbering (slot = bucket * 4 + place): first rows / accumulators / the final walk do not change.
    auto bucket_find = [&](const u64 key, unsigned b, pl_v2 q0, pl_v2 q1) __attribute__((always_inline)) {
        if ((i64)key == RFX_NULL_I64_D) return -1;
        const unsigned nb = C >> 2;
        unsigned steps = 0;
        bool fresh = true; // q0 / q1 hold bucket b as it stood a moment ago
        for (;;) {
            if (!fresh) {
                q0 = *(const pl_v2 *)&lkey[(size_t)b * 4];
                q1 = *(const pl_v2 *)&lkey[(size_t)b * 4 + 2];
            }
            fresh = false;
            const int hit = q0.x == key ? 0 : q0.y == key ? 1 : q1.x == key ? 2 : q1.y == key ? 3 : -1;
            if (hit >= 0) return (int)(b * 4u) + hit;
            const int e = (i64)q0.x == RFX_NULL_I64_D ? 0 : (i64)q0.y == RFX_NULL_I64_D ? 1 : (i64)q1.x == RFX_NULL_I64_D ? 2 : (i64)q1.y == RFX_NULL_I64_D ? 3 : -1;
            if (e >= 0) { // room here: claim the first empty place (places only ever fill up; a lost race re-reads the SAME bucket: at most four times)
                const u64 old = atomicCAS((unsigned long long *)&lkey[(size_t)b * 4 + e], (unsigned long long)RFX_NULL_I64_D, (unsigned long long)key);
                if ((i64)old == RFX_NULL_I64_D || old == key) return (int)(b * 4u) + e;
                continue;
            }
            if (++steps >= nb) break; // every bucket is full: no room in this table
            b = (b + 1 == nb) ? 0 : b + 1;
        }
        return -1;
    };
    auto consume2 = [&](const Batch &B) __attribute__((always_inline)) {
        const i64 rbase = (i64)B.b * X.block_rows;
        unsigned mm[4], sb[4], lrow[4];
        u64 key[4], val[4];
        pl_v2 q0[4], q1[4];
        bool on[4], cold[4];
        int idx[4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned i = B.i0 + ((unsigned)k * 64u + (unsigned)lane) * 2u;
            mm[2 * k] = B.m[k].x, mm[2 * k + 1] = B.m[k].y;
            key[2 * k] = B.key[k].x, key[2 * k + 1] = B.key[k].y;
            val[2 * k] = B.val[k].x, val[2 * k + 1] = B.val[k].y;
            on[2 * k] = i < B.n && mine_of(mm[2 * k]);
            on[2 * k + 1] = i + 1 < B.n && mine_of(mm[2 * k + 1]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { // the four first buckets back to back: eight 16-byte LDS reads in flight
            sb[j] = start_of(mm[j]) >> 2;
            q0[j] = *(const pl_v2 *)&lkey[(size_t)sb[j] * 4];
            q1[j] = *(const pl_v2 *)&lkey[(size_t)sb[j] * 4 + 2];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) idx[j] = on[j] ? bucket_find(key[j], sb[j], q0[j], q1[j]) : -1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lrow[j] = (unsigned)(rbase + (i64)(mm[j] >> PL_SLOT_BITS));
            cold[j] = on[j] && idx[j] < 0;
            if (!on[j] || idx[j] < 0) continue;
            __hip_atomic_fetch_min(&lfirst[idx[j]], lrow[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (FAST) unsafeAtomicAdd((double *)&larr[idx[j]], rfx_as_f64(val[j]));
            else {
#pragma unroll
                for (int a = 0; a < RFX_MAX_AGGS; a++) {
                    if (kind[a] < 0) continue;
                    group_apply(&larr[(size_t)arr_of[a] * C + idx[j]], &larr[(size_t)(arr_of[a] + 1) * C + idx[j]], kind[a], f64[a], apl[a] == 1 ? val[j] : 0ULL, skip[a]);
                }
            }
        }
        while (__builtin_expect(__builtin_amdgcn_ballot_w64(cold[0] | cold[1] | cold[2] | cold[3]) != 0, 0)) {
            const bool any = cold[0] | cold[1] | cold[2] | cold[3];
            const int pick = cold[0] ? 0 : cold[1] ? 1 : cold[2] ? 2 : 3;
            const u64 ck = pick == 0 ? key[0] : pick == 1 ? key[1] : pick == 2 ? key[2] : key[3];
            const u64 cv = pick == 0 ? val[0] : pick == 1 ? val[1] : pick == 2 ? val[2] : val[3];
            const unsigned cr = pick == 0 ? lrow[0] : pick == 1 ? lrow[1] : pick == 2 ? lrow[2] : lrow[3];
            cold[0] = cold[0] && pick != 0;
            cold[1] = cold[1] && pick != 1;
            cold[2] = cold[2] && pick != 2;
            cold[3] = cold[3] && pick != 3;
            if (any) fold(-1, cr, 0u, ck, cv);
        }
    };
    auto consume_v = [&](const Batch &B) __attribute__((always_inline)) {
        if constexpr (VAR == 1) consume1(B);
        else if constexpr (VAR == 2) consume2(B);
        else consume(B);
    };
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nreg = q < X.nblk ? (X.nblk - q + NW - 1) / NW : 0;
    unsigned *mywcnt = wcnt + (tid >> 6) * 64;
    int r = -1;
    int b = q;
    unsigned i0 = 0, n = 0;
    auto advance = [&]() __attribute__((always_inline)) {
        for (;;) {
            if (r >= 0 && i0 < n) return true;
            r++;
            if (r >= nreg) return false;
            if ((r & 63) == 0) {
                const int r0 = r + lane;
                mywcnt[lane] = (r0 < nreg) ? X.cnt[(size_t)(q + r0 * NW) * (size_t)X.parts + p] : 0u;
            }
            n = (unsigned)__builtin_amdgcn_readfirstlane((int)mywcnt[r & 63]);
            b = q + r * NW;
            i0 = 0;
        }
    };
    Batch B0, B1;
    bool h0 = advance();
    load(h0, b, i0, n, B0);
    i0 += 256u;
    while (h0) {
        const bool h1 = advance();
        load(h1, b, i0, n, B1);
        i0 += 256u;
        consume_v(B0);
        if (!h1) break;
        h0 = advance();
        load(h0, b, i0, n, B0);
        i0 += 256u;
        consume_v(B1);
    }
    __syncthreads();
    // the partition's groups into the device-wide table: one insert per distinct key
    for (unsigned i = tid; i < C; i += PLH_T) {
        const u64 key = lkey[i];
        if ((i64)key == RFX_NULL_I64_D) continue;
        const i64 g = hash_slot(X.H.keys, X.H.capacity, key);
        if (g < 0) {
            atomicExch(X.overflow, 1);
            continue;
        }
        const u64 row = (u64)P.row0 + lfirst[i];
        if (row < X.H.first[g]) atomicMin((unsigned long long *)&X.H.first[g], (unsigned long long)row);
#pragma unroll
        for (int a = 0; a < RFX_MAX_AGGS; a++) {
            if (kind[a] < 0) continue;
            const bool hc = agg_has_cnt(kind[a], f64[a]);
            group_merge_cell(&X.H.acc[a][g], hc ? &X.H.cnt[a][g] : (u64 *)0, kind[a], f64[a], larr[(size_t)arr_of[a] * C + i],
                             hc ? larr[(size_t)(arr_of[a] + 1) * C + i] : 0ULL);
        }
    }
}

// which probing form k_plane_hash_aggregate runs: 2 = four-key buckets (default since round 6: 16.3 -> 13.2 ms per 1e9 rows / 1e6 keys, profiles/r06_k9_ab.txt),
// 0 = slot-by-slot linear probing (rounds 3-5), 1 = 0 with the spill path as a cold loop.  RFX_PLH_VAR for A/B.
static int plh_var() {
    static const char *var_env = getenv("RFX_PLH_VAR");
    return var_env ? atoi(var_env) : 2;
}
// Sparse-key group-by through the planes.  est = sampled distinct-key estimate.  RFX_ESTATE: not applicable / gave up BEFORE anything
// touched the caller's tables (the caller takes round 1's kernels); otherwise the tables hold the answer unless *d_overflow is set.
int rfx_plane_hash_accumulate(rfx_ctx *c, const Plan &P, int key_idx, const HashArgs &H, double est, int *d_overflow) {
    if (c->flags & (RFX_TUNE_NO_PLANE | RFX_TUNE_NO_PARTITION)) return RFX_ESTATE;
    const i64 nrows = P.nrows;
    if (nrows < ((c->flags & RFX_TUNE_CHUNK_SMALL) ? (1LL << 16) : (1LL << 22)) || nrows >= 0xFFFFFFF0LL || P.nx > 0 || P.nagg < 1) return RFX_ESTATE;
    int vcol[PL_MAX_NV], agg_plane[RFX_MAX_AGGS];
    const int nv = plane_value_cols(P, vcol, agg_plane);
    if (nv < 0 || nv > 1) return RFX_ESTATE; // the records carry the key and ONE value
    int narr = 0;
    for (int a = 0; a < P.nagg; a++) narr += agg_has_cnt(P.aggs[a].kind, P.aggs[a].f64) ? 2 : 1;
    // LDS: entry = key 8 + narr x 8 + first 4; beside it the waves' count windows
    const size_t entry = 8 + (size_t)narr * 8 + 4;
    const unsigned lcap = (unsigned)((((size_t)150 * 1024) - 16 * 64 * 4) / entry) & ~63u;
    if (est < 4096.0) return RFX_ESTATE; // a few keys do not spread over 128 partitions: one of them would overflow its regions
    // 128 partitions (16-record store groups: 128-byte value lines), shared by 2 / 4 workgroups when a partition's keys overflow one LDS
    // table; 256 partitions (8-record groups: 64-byte lines) only beyond that.  Measured at 1e6 keys: 128 x 2 workgroups 8.1 + 12.6 ms,
    // 256 x 1 13.3 + 8.2 ms -- the short lines cost the scatter more than the single stream saves the aggregate.
    // headroom of the LDS tables over the estimated keys of a partition: slot-by-slot linear probing wants load <= 0.6 (chains grow fast beyond), four-key
    // buckets hold up to ~0.8 (a bucket overflows into the next one): 1.3 lets sum + count (28-byte entries, 5 248 per table) keep ONE workgroup per
    // partition at 1e6 keys where 1.6 split every partition between two workgroups that each streamed all of its records (30 ms)
    const double HEAD = plh_var() == 2 ? 1.3 : 1.6;
    int pbits = 7, hbits = 0, parts = 128;
    static const char *force = getenv("RFX_PLANE_HASH_PARTS"); // (A/B: 192 = the one-workgroup-per-partition form)
    if (force && atoi(force) == 192 && est * HEAD / 128.0 > (double)lcap && est * 1.2 / 192.0 <= (double)lcap) {
        // round 4 experiment, OPT-IN (RFX_PLANE_HASH_PARTS=192): 192 partitions, ONE workgroup each -- every record streamed once (traffic
        // 80 -> 58 GB).  Measured at 1e9 rows / 1e6 keys: scatter 7.8 ms (as with 128 partitions: same 16-record lines), aggregate 27 ms against
        // 12.6 -- the aggregate is bound by LDS probing per record (192 CUs at load 0.70 instead of 256 at 0.53), not by the bytes it streams:
        // the double stream was never the bottleneck.  Kept for the record and for tables that would otherwise overflow to the device-wide one.
        pbits = 8;
        parts = 192;
    } else {
        if (est * HEAD / (128.0 * 4.0) > (double)lcap) pbits = 8;
        // round 5: keys that overflow ONE LDS table per partition at 128 partitions take 256 partitions with ONE workgroup each -- every record
        // streamed once (the two workgroups per partition of round 3 each streamed all of it: 42.8 of the query's 80 GB) -- now that the
        // 256-partition scatter stores whole 128-byte lines (KVI).  RFX_PLANE_HASH_PARTS=128 keeps the two-workgroup form (A/B).
        if (est * HEAD / 128.0 > (double)lcap && est * HEAD / 256.0 <= (double)lcap && !(force && atoi(force) == 128)) pbits = 8;
        while (hbits < 2 && est * HEAD / (double)((1 << pbits) << hbits) > (double)lcap) hbits++; // (every workgroup of a partition streams ALL its records: 4 is where that stops paying)
        if (est * HEAD / (double)((1 << pbits) << hbits) > (double)lcap) return RFX_ESTATE; // more keys than 512 LDS tables hold: round 1's kernels / the device-wide table
        parts = 1 << pbits;
    }
    const i64 nblk64 = (nrows + PL_BLOCK_ROWS - 1) / PL_BLOCK_ROWS;
    if (nblk64 > (1 << 20)) return RFX_ESTATE;
    // key -> columns 0 and 1 (plane 0 IS the key), the value column -> 2 (plane 1; without one, the key again), the predicates' columns behind
    Plan Pc = P;
    {
        int perm[RFX_MAX_COLS + 2], inv[RFX_MAX_COLS], n2 = 0;
        for (int i = 0; i < P.ncols; i++) inv[i] = -1;
        perm[n2++] = key_idx;
        inv[key_idx] = 0;
        perm[n2++] = key_idx;
        perm[n2++] = nv == 1 ? vcol[0] : key_idx;
        if (nv == 1 && inv[vcol[0]] < 0) inv[vcol[0]] = 2;
        for (int i = 0; i < P.ncols; i++)
            if (inv[i] < 0) {
                inv[i] = n2;
                perm[n2++] = i;
            }
        if (n2 > 4 || n2 > RFX_MAX_COLS) return RFX_ESTATE; // one predicate column beside key and value
        Pc.ncols = n2;
        for (int i = 0; i < n2; i++) Pc.cols[i] = P.cols[perm[i]];
        for (int i = 0; i < P.npred; i++) {
            Pc.preds[i].col = inv[P.preds[i].col];
            if (P.preds[i].rhs_col >= 0) Pc.preds[i].rhs_col = inv[P.preds[i].rhs_col];
        }
    }
    // selectivity: unknown here -- regions sized for every row (a selective filter only leaves them emptier)
    double share = (double)PL_BLOCK_ROWS / (double)parts;
    unsigned c0 = (unsigned)(share * 1.25 + 160.0);
    c0 = (c0 + 127u) & ~127u;
    PlaneArgs A;
    memset(&A, 0, sizeof(A));
    size_t need = 0;
    plane_layout(c, (int)nblk64, pbits, c0, 2, &A, &need, parts);
    if (rfx_chunk_reserve(c, need) != RFX_OK) return RFX_ESTATE;
    plane_layout(c, (int)nblk64, pbits, c0, 2, &A, NULL, parts);
    A.block_rows = PL_BLOCK_ROWS;
    for (int i = 0; i < Pc.npred; i++) {
        A.pmask |= 1u << Pc.preds[i].col;
        if (Pc.preds[i].rhs_col >= 0) A.pmask |= 1u << Pc.preds[i].rhs_col;
    }
    c->ck_valid = 0;
    rfx_plane_invalidate(c);
    RFX_HIP_CHECK(hipMemsetAsync(A.ctl, 0, 256, c->stream));
    c->ext_i[3 + RFX_STAT_PLANE_SCATTER]++;
    RFX_KERNEL_BEGIN(c);
    int rc;
    if (parts == 192) rc = Pc.ncols == 3 ? launch_plane_scatter_np<3, 2, 8, 4, true>(c, Pc, A) : launch_plane_scatter_np<4, 2, 8, 4, true>(c, Pc, A);
    else if (pbits == 7) rc = Pc.ncols == 3 ? launch_plane_scatter_np<3, 2, 7, 4, true>(c, Pc, A) : launch_plane_scatter_np<4, 2, 7, 4, true>(c, Pc, A);
    else rc = Pc.ncols == 3 ? launch_plane_scatter_np<3, 2, 8, 3, true>(c, Pc, A) : launch_plane_scatter_np<4, 2, 8, 3, true>(c, Pc, A);
    if (rc != RFX_OK) return rc;
    RFX_HIP_CHECK(hipGetLastError());
    unsigned *hctl = (unsigned *)c->h_pin;
    RFX_HIP_CHECK(hipMemcpyAsync(hctl, A.ctl, 256, hipMemcpyDeviceToHost, c->stream));
    RFX_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (hctl[1]) { // a region overflowed (one key takes a large share of the rows): round 1's kernels
        c->ext_i[3 + RFX_STAT_PLANE_FALLBACK]++;
        return RFX_ESTATE;
    }
    PlaneHashArgs X;
    memset(&X, 0, sizeof(X));
    X.H = H;
    X.nblk = (int)nblk64;
    X.pbits = pbits;
    X.hbits = hbits;
    X.parts = parts;
    X.block_rows = PL_BLOCK_ROWS;
    X.c0 = c0;
    X.lcap = lcap;
    X.narr = narr;
    for (int a = 0; a < RFX_MAX_AGGS; a++) X.agg_pl[a] = (a < P.nagg && agg_plane[a] == 0) ? 1 : -1;
    X.keys = A.vals[0];
    X.vals = A.vals[1];
    X.kvi = (parts == 256); // (<.., 2, 8, 3, true>: the interleaved key | value plane)
    {
        static const char *dbg = getenv("RFX_PLH_DBG");
        X.dbg = dbg ? atoi(dbg) : 0;
    }
    X.meta = A.meta;
    X.cnt = A.cnt;
    X.overflow = d_overflow;
    const size_t lds = (size_t)lcap * entry + 16 * 64 * 4 + 64;
    static unsigned long long attr_set = 0; /* one bit per device: function attributes are per device */
    if (!((attr_set >> (c->device & 63)) & 1ull)) {
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_hash_aggregate<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_hash_aggregate<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_hash_aggregate<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_hash_aggregate<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_hash_aggregate<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RFX_HIP_CHECK(hipFuncSetAttribute((const void *)k_plane_hash_aggregate<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        __atomic_fetch_or(&attr_set, 1ull << (c->device & 63), __ATOMIC_RELAXED);
    }
    c->ext_i[3 + RFX_STAT_PLANE_AGGREGATE]++;
    const bool fast = P.nagg == 1 && P.aggs[0].kind == RFX_AGG_SUM && P.aggs[0].f64 && !P.aggs[0].skipnull && X.agg_pl[0] == 1;
    const int var = plh_var();
    if (var == 0) {
        if (fast) hipLaunchKernelGGL((k_plane_hash_aggregate<true, 0>), dim3(parts << hbits), dim3(PLH_T), lds, c->stream, Pc, X);
        else hipLaunchKernelGGL((k_plane_hash_aggregate<false, 0>), dim3(parts << hbits), dim3(PLH_T), lds, c->stream, Pc, X);
    } else if (var == 2) {
        if (fast) hipLaunchKernelGGL((k_plane_hash_aggregate<true, 2>), dim3(parts << hbits), dim3(PLH_T), lds, c->stream, Pc, X);
        else hipLaunchKernelGGL((k_plane_hash_aggregate<false, 2>), dim3(parts << hbits), dim3(PLH_T), lds, c->stream, Pc, X);
    } else {
        if (fast) hipLaunchKernelGGL((k_plane_hash_aggregate<true, 1>), dim3(parts << hbits), dim3(PLH_T), lds, c->stream, Pc, X);
        else hipLaunchKernelGGL((k_plane_hash_aggregate<false, 1>), dim3(parts << hbits), dim3(PLH_T), lds, c->stream, Pc, X);
    }
    RFX_KERNEL_END(c);
    RFX_HIP_CHECK(hipGetLastError());
    return RFX_OK;
}
