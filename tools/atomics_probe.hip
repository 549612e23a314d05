// tools/atomics_probe.hip -- can a 1e6-key f64 group-by be ONE 16 B/row pass over device-memory atomics on MI355X?
// Not part of the product library.  Build: hipcc --offload-arch=gfx950 -O3 tools/atomics_probe.hip -o tools/atomics_probe
// Run:   tools/atomics_probe [rows_log2=27] > profiles/atomics_r02.jsonl      (one JSON line per configuration)
//
// Sweep: no-return atomics  {f64 add, u64 add, u64 min, f32 add, u32 add}
//        x scope            {agent, workgroup}
//        x table bytes      {64 KB .. 64 MB}   (uniformly random slots, one table per device or one PRIVATE table per XCD)
//        x allocation       {hipMalloc (coarse-grained), hipExtMallocWithFlags(hipDeviceMallocFinegrained)}
//        x pre-sorting      {none, keys sorted inside each 64-lane wave}
// Each launch streams `rows` 8-byte keys (coalesced, like the group-by's key column) and issues one atomic per row.
// The counters that say WHERE the atomics execute (TCC_ATOMIC / TCC_HIT / TCC_EA0_ATOMIC ...) are collected by running this
// binary under `rocprofv3 --pmc ...` (tools/atomics_pmc.sh); kernel names carry the configuration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned long long u64;
typedef long long i64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __host__ inline u64 mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ void gen_keys(u64 *out, i64 n, u64 seed, u64 mod) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        out[i] = mix(seed + (u64)(i + 1) * 0x9E3779B97F4A7C15ULL) % mod;
}

enum { OP_F64ADD = 0, OP_U64ADD = 1, OP_U64MIN = 2, OP_F32ADD = 3, OP_U32ADD = 4 };
static const char *op_name[] = {"f64add", "u64add", "u64min", "f32add", "u32add"};

// SCOPE: 0 agent, 1 workgroup.  PERXCD: every XCD updates its own private copy of the table (block b runs on XCD b % 8).
template <int OP, int SCOPE, int PERXCD, int SORTED>
__global__ __launch_bounds__(256) void k_atomics(const u64 *__restrict__ keys, i64 n, void *table, u64 slots) {
    constexpr int scope = SCOPE ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
    const u64 esz = (OP >= OP_F32ADD) ? 4 : 8;
    char *tab = (char *)table + (PERXCD ? (u64)(blockIdx.x & 7) * slots * esz : 0);
    for (i64 i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        u64 k = keys[i];
        if (SORTED) { // bitonic sort of the 64 keys of a wave (so neighbouring lanes hit neighbouring slots)
            for (int sz = 2; sz <= 64; sz <<= 1)
                for (int st = sz >> 1; st > 0; st >>= 1) {
                    const int lane = threadIdx.x & 63;
                    unsigned lo = __shfl_xor((unsigned)k, st, 64), hi = __shfl_xor((unsigned)(k >> 32), st, 64);
                    const u64 o = ((u64)hi << 32) | lo;
                    const bool up = ((lane & sz) == 0), lower = ((lane & st) == 0);
                    const bool take_min = (up == lower);
                    k = take_min ? (k < o ? k : o) : (k > o ? k : o);
                }
        }
        if (OP == OP_F64ADD) __hip_atomic_fetch_add((double *)tab + k, 1.0, __ATOMIC_RELAXED, scope);
        if (OP == OP_U64ADD) __hip_atomic_fetch_add((u64 *)tab + k, (u64)i, __ATOMIC_RELAXED, scope);
        if (OP == OP_U64MIN) __hip_atomic_fetch_min((u64 *)tab + k, (u64)i, __ATOMIC_RELAXED, scope);
        if (OP == OP_F32ADD) __hip_atomic_fetch_add((float *)tab + k, 1.0f, __ATOMIC_RELAXED, scope);
        if (OP == OP_U32ADD) __hip_atomic_fetch_add((unsigned *)tab + k, 1u, __ATOMIC_RELAXED, scope);
    }
}

// the bound every variant is compared with: the same key stream without the atomic
__global__ __launch_bounds__(256) void k_stream_only(const u64 *__restrict__ keys, i64 n, u64 *out) {
    u64 acc = 0;
    for (i64 i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (i64)gridDim.x * 256) acc += keys[i];
    if (acc == 0x1234567ULL) out[0] = acc;
}

template <int OP, int SCOPE, int PERXCD, int SORTED>
static void run(const u64 *d_keys, i64 n, void *d_table, u64 slots, const char *alloc, int grid, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const u64 esz = (OP >= OP_F32ADD) ? 4 : 8;
    CK(hipMemsetAsync(d_table, 0, slots * esz * (PERXCD ? 8 : 1), st));
    hipLaunchKernelGGL((k_atomics<OP, SCOPE, PERXCD, SORTED>), dim3(grid), dim3(256), 0, st, d_keys, n, d_table, slots); // warm
    CK(hipEventRecord(e0, st));
    const int reps = 3;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_atomics<OP, SCOPE, PERXCD, SORTED>), dim3(grid), dim3(256), 0, st, d_keys, n, d_table, slots);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("{\"probe\":\"atomics\",\"op\":\"%s\",\"scope\":\"%s\",\"table\":\"%s\",\"alloc\":\"%s\",\"sorted_in_wave\":%d,\"slots\":%llu,\"table_bytes\":%llu,"
           "\"rows\":%lld,\"ms\":%.4f,\"Grows_per_s\":%.3f,\"ms_per_1e9_rows\":%.2f}\n",
           op_name[OP], SCOPE ? "workgroup" : "agent", PERXCD ? "private per XCD" : "one", alloc, SORTED, slots, slots * esz, (long long)n, ms,
           n / (ms * 1e-3) / 1e9, ms * 1e9 / n);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 27;
    const i64 n = 1LL << lg;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 8;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    u64 *d_keys;
    CK(hipMalloc(&d_keys, n * 8));
    const u64 max_table = 64ULL << 20;
    void *d_coarse, *d_fine;
    CK(hipMalloc(&d_coarse, max_table * 8));
    CK(hipExtMallocWithFlags(&d_fine, max_table, hipDeviceMallocFinegrained));
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"rows\":%lld}\n", prop.name, prop.multiProcessorCount, (long long)n);
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        u64 *d_out;
        CK(hipMalloc(&d_out, 8));
        hipLaunchKernelGGL(gen_keys, dim3(grid), dim3(256), 0, st, d_keys, n, 7ULL, 1000000ULL);
        hipLaunchKernelGGL(k_stream_only, dim3(grid), dim3(256), 0, st, d_keys, n, d_out);
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_stream_only, dim3(grid), dim3(256), 0, st, d_keys, n, d_out);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"probe\":\"stream_only\",\"rows\":%lld,\"ms\":%.4f,\"Grows_per_s\":%.3f}\n", (long long)n, ms / 5, n / (ms / 5 * 1e-3) / 1e9);
    }
    const u64 table_bytes[] = {64ULL << 10, 512ULL << 10, 2ULL << 20, 8ULL << 20, 64ULL << 20};
    for (u64 tb : table_bytes) {
        const u64 slots = tb / 8;
        hipLaunchKernelGGL(gen_keys, dim3(grid), dim3(256), 0, st, d_keys, n, 7ULL, slots);
        CK(hipStreamSynchronize(st));
        // device scope, one table
        run<OP_F64ADD, 0, 0, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_U64ADD, 0, 0, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_U64MIN, 0, 0, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_F32ADD, 0, 0, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_U32ADD, 0, 0, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_F64ADD, 0, 0, 1>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        // workgroup scope (the ISA form without sc1: may execute in the XCD's own L2), one table and private per-XCD tables
        run<OP_F64ADD, 1, 0, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_F64ADD, 1, 1, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_U64MIN, 1, 1, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        run<OP_F64ADD, 0, 1, 0>(d_keys, n, d_coarse, slots, "hipMalloc", grid, st);
        if (tb <= (8ULL << 20)) {
            run<OP_F64ADD, 0, 0, 0>(d_keys, n, d_fine, slots, "fine-grained", grid, st);
            run<OP_F64ADD, 1, 1, 0>(d_keys, n, d_fine, slots, "fine-grained", grid, st);
        }
    }
    return 0;
}
