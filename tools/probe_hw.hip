// tools/probe_hw.hip -- design-decision micro-probes for the MI355X group-by / streaming kernels.
// Not part of the product library.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_hw.hip -o tools/probe_hw
// Prints one JSON line per measurement.  Questions it answers (DESIGN.md cites the numbers):
//   Q1  what read bandwidth does a plain streaming reduction reach (unroll, nt-loads, blocks per CU)?
//   Q2  how fast are device-scope f64 / u64 atomics into a table of R slots with uniformly random keys?
//   Q3  are workgroup-scope (L2-executed) atomics into a per-XCD private table faster?
//   Q4  how fast is an LDS-privatised table for small R?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
typedef long long i64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __host__ inline u64 mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void gen_i64(i64 *out, i64 n, u64 seed, u64 mod) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        out[i] = (i64)(mix(seed + (u64)(i + 1) * 0x9E3779B97F4A7C15ULL) % mod);
}
__global__ void gen_f64(double *out, i64 n, u64 seed) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        out[i] = (double)(mix(seed + (u64)(i + 1) * 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53;
}

// ---------------- Q1: streaming read ----------------
typedef u64 v2u64 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void stream_sum(const u64 *__restrict__ in, i64 n, u64 *out) {
    const i64 tile = 256 * 2 * U;
    i64 ntiles = n / tile;
    u64 acc = 0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 *p = in + t * tile + threadIdx.x * 2;
        v2u64 v[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            if (NT) v[j] = __builtin_nontemporal_load((const v2u64 *)(p + j * 512));
            else v[j] = *(const v2u64 *)(p + j * 512);
        }
#pragma unroll
        for (int j = 0; j < U; j++) acc += (v[j].x < 100000 ? v[j].x : 0) + (v[j].y < 100000 ? v[j].y : 0);
    }
    // crude block reduce
    for (int m = 32; m; m >>= 1) {
        unsigned lo = __shfl_xor((unsigned)acc, m, 64), hi = __shfl_xor((unsigned)(acc >> 32), m, 64);
        acc += ((u64)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

template <int U, bool NT>
float run_stream(const u64 *d_in, i64 n, u64 *d_out, int grid, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((stream_sum<U, NT>), dim3(grid), dim3(256), 0, 0, d_in, n, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((stream_sum<U, NT>), dim3(grid), dim3(256), 0, 0, d_in, n, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

// ---------------- Q2/Q3: global atomics ----------------
__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// mode 0: device-scope f64 add ; 1: device-scope u64 min ; 2: both ; 3: workgroup-scope f64 add into per-XCD table
// 4: device-scope i64 add (integer)
template <int MODE, bool READ>
__global__ __launch_bounds__(256) void scatter_probe(const i64 *__restrict__ keys, const double *__restrict__ vals, i64 n, u64 R,
                                                       double *tab, u64 *first) {
    unsigned x = (MODE == 3) ? xcc_id() : 0;
    double *mytab = tab + (size_t)x * R;
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 k; double v;
        if (READ) { k = (u64)keys[i]; v = vals[i]; }
        else { k = mix((u64)i * 0x9E3779B97F4A7C15ULL) % R; v = 1.0; }
        if (MODE == 0 || MODE == 2) unsafeAtomicAdd(&tab[k], v);
        if (MODE == 1 || MODE == 2) __hip_atomic_fetch_min(&first[k], (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) __hip_atomic_fetch_add(&mytab[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) __hip_atomic_fetch_add((u64 *)&tab[k], (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------- Q4: LDS-privatised table (R small) ----------------
template <int RL>
__global__ __launch_bounds__(256) void lds_probe(const i64 *__restrict__ keys, const double *__restrict__ vals, i64 n, double *tab) {
    __shared__ double lt[RL];
    for (int i = threadIdx.x; i < RL; i += 256) lt[i] = 0.0;
    __syncthreads();
    const i64 tile = 256 * 2 * 4;
    i64 ntiles = n / tile;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 *pk = (const u64 *)keys + t * tile + threadIdx.x * 2;
        const u64 *pv = (const u64 *)vals + t * tile + threadIdx.x * 2;
        v2u64 k[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { k[j] = __builtin_nontemporal_load((const v2u64 *)(pk + j * 512)); v[j] = __builtin_nontemporal_load((const v2u64 *)(pv + j * 512)); }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsafeAtomicAdd(&lt[k[j].x & (RL - 1)], __longlong_as_double((i64)v[j].x));
            unsafeAtomicAdd(&lt[k[j].y & (RL - 1)], __longlong_as_double((i64)v[j].y));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RL; i += 256) unsafeAtomicAdd(&tab[i], lt[i]);
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    i64 n = (argc > 1) ? atoll(argv[1]) : (1LL << 28);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_khz\":%d,\"mem_gb\":%.1f}\n", prop.name, cus, prop.clockRate, prop.totalGlobalMem / 1e9);
    i64 *d_k; double *d_v; u64 *d_out;
    CK(hipMalloc(&d_k, n * 8)); CK(hipMalloc(&d_v, n * 8)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_out, 0, 64));

    // Q1
    hipLaunchKernelGGL(gen_i64, dim3(cus * 8), dim3(256), 0, 0, d_k, n, 2ULL, 1000000ULL);
    hipLaunchKernelGGL(gen_f64, dim3(cus * 8), dim3(256), 0, 0, d_v, n, 5ULL);
    CK(hipDeviceSynchronize());
    int bpcs[] = {2, 4, 8, 16};
    for (int bi = 0; bi < 4; bi++) {
        int grid = cus * bpcs[bi];
        struct { const char *name; float ms; } r[] = {
            {"U2", run_stream<2, false>((u64 *)d_k, n, d_out, grid, 10)}, {"U2nt", run_stream<2, true>((u64 *)d_k, n, d_out, grid, 10)},
            {"U4", run_stream<4, false>((u64 *)d_k, n, d_out, grid, 10)}, {"U4nt", run_stream<4, true>((u64 *)d_k, n, d_out, grid, 10)},
            {"U8", run_stream<8, false>((u64 *)d_k, n, d_out, grid, 10)}, {"U8nt", run_stream<8, true>((u64 *)d_k, n, d_out, grid, 10)},
        };
        for (auto &x : r)
            printf("{\"probe\":\"stream_read\",\"variant\":\"%s\",\"blocks_per_cu\":%d,\"n\":%lld,\"ms\":%.4f,\"GBps\":%.1f}\n", x.name, bpcs[bi], n, x.ms, n * 8 / x.ms / 1e6);
    }
    fflush(stdout);

    // Q2/Q3
    u64 Rs[] = {1000ULL, 100000ULL, 1000000ULL, 10000000ULL};
    for (u64 R : Rs) {
        double *d_tab; u64 *d_first;
        CK(hipMalloc(&d_tab, R * 8 * 8)); CK(hipMalloc(&d_first, R * 8));
        CK(hipMemset(d_tab, 0, R * 8 * 8)); CK(hipMemset(d_first, 0xff, R * 8));
        hipLaunchKernelGGL(gen_i64, dim3(cus * 8), dim3(256), 0, 0, d_k, n, 4ULL, R);
        CK(hipDeviceSynchronize());
        int grid = cus * 8;
        i64 nn = n / 4; // atomics are slow: keep each probe short
        float t0 = timeit([&] { hipLaunchKernelGGL((scatter_probe<0, false>), dim3(grid), dim3(256), 0, 0, d_k, d_v, nn, R, d_tab, d_first); }, 3);
        float t1 = timeit([&] { hipLaunchKernelGGL((scatter_probe<1, false>), dim3(grid), dim3(256), 0, 0, d_k, d_v, nn, R, d_tab, d_first); }, 3);
        float t2 = timeit([&] { hipLaunchKernelGGL((scatter_probe<2, true>), dim3(grid), dim3(256), 0, 0, d_k, d_v, nn, R, d_tab, d_first); }, 3);
        float t3 = timeit([&] { hipLaunchKernelGGL((scatter_probe<3, false>), dim3(grid), dim3(256), 0, 0, d_k, d_v, nn, R, d_tab, d_first); }, 3);
        float t4 = timeit([&] { hipLaunchKernelGGL((scatter_probe<4, false>), dim3(grid), dim3(256), 0, 0, d_k, d_v, nn, R, d_tab, d_first); }, 3);
        float t5 = timeit([&] { hipLaunchKernelGGL((scatter_probe<0, true>), dim3(grid), dim3(256), 0, 0, d_k, d_v, nn, R, d_tab, d_first); }, 3);
        printf("{\"probe\":\"atomics\",\"R\":%llu,\"rows\":%lld,\"f64add_agent_Grows_s\":%.3f,\"u64min_agent_Grows_s\":%.3f,\"read_add_min_Grows_s\":%.3f,"
               "\"f64add_wg_perxcd_Grows_s\":%.3f,\"i64add_agent_Grows_s\":%.3f,\"read_f64add_Grows_s\":%.3f}\n",
               R, nn, nn / t0 / 1e6, nn / t1 / 1e6, nn / t2 / 1e6, nn / t3 / 1e6, nn / t4 / 1e6, nn / t5 / 1e6);
        fflush(stdout);
        CK(hipFree(d_tab)); CK(hipFree(d_first));
    }

    // Q4
    {
        double *d_tab; CK(hipMalloc(&d_tab, 16384 * 8)); CK(hipMemset(d_tab, 0, 16384 * 8));
        hipLaunchKernelGGL(gen_i64, dim3(cus * 8), dim3(256), 0, 0, d_k, n, 4ULL, 1000000ULL);
        CK(hipDeviceSynchronize());
        float a = timeit([&] { hipLaunchKernelGGL((lds_probe<1024>), dim3(cus * 8), dim3(256), 0, 0, d_k, d_v, n, d_tab); }, 5);
        float b = timeit([&] { hipLaunchKernelGGL((lds_probe<4096>), dim3(cus * 4), dim3(256), 0, 0, d_k, d_v, n, d_tab); }, 5);
        float c = timeit([&] { hipLaunchKernelGGL((lds_probe<8192>), dim3(cus * 2), dim3(256), 0, 0, d_k, d_v, n, d_tab); }, 5);
        printf("{\"probe\":\"lds_table\",\"rows\":%lld,\"R1024_Grows_s\":%.3f,\"R4096_Grows_s\":%.3f,\"R8192_Grows_s\":%.3f}\n", n, n / a / 1e6, n / b / 1e6, n / c / 1e6);
    }
    return 0;
}
