// tools/write_probe.hip -- what does a scatter of SEGMENTS of S contiguous bytes cost on MI355X, as a function of S, of the
// per-lane store width W and of the pattern?  (Design input for the two-plane partition records: 8-byte value plane + 2/4-byte meta
// plane -- can a plane leave in 32- or 64-byte pieces, or must every store instruction cover whole 128-byte lines?)
// Not part of the product library.  Build: hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o tools/write_probe
//
// Model of the partition scatter: 256 workgroups x 256 streams (one per (workgroup, partition)); per "drain" a workgroup appends one
// segment of S bytes to each of its streams (pattern "append": the next segment of a stream is adjacent to the previous one, written one
// drain later) or puts it at a pseudo-random S-aligned place of the stream's region (pattern "random": lines never get completed soon).
// Optionally every workgroup also streams READ bytes (16 B per lane, non-temporal) from a big input between the drains, RD bytes read per
// byte written (1.6 = 16 GB read for 10 GB of records).
// One JSON line per configuration: {"probe":"write","S":..,"W":..,"pattern":..,"nt":..,"rd":..,"ms":..,"write_TBps":..,"total_TBps":..}
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u64 mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef u64 v2u64 __attribute__((ext_vector_type(2)));

template <int W> struct Word;
template <> struct Word<2> { typedef unsigned short T; };
template <> struct Word<4> { typedef u32 T; };
template <> struct Word<8> { typedef u64 T; };
template <> struct Word<16> { typedef v2u64 T; };

template <int W, bool NT>
__device__ __forceinline__ void put(char *p, u64 x) {
    typedef typename Word<W>::T T;
    T v;
    if constexpr (W == 16) { v.x = x; v.y = ~x; } else v = (T)x;
    if constexpr (NT) __builtin_nontemporal_store(v, (T *)p);
    else *(T *)p = v;
}

// S: segment bytes, W: bytes per lane.  region: bytes per stream.  ndrain drains.
template <int W, bool NT>
__global__ __launch_bounds__(1024) void k_write(char *out, u64 region, int S, int ndrain, int random, const v2u64 *in, u64 in_elems, float rdw, u64 *sink) {
    const int tid = threadIdx.x;
    const int lanes_per_seg = S / W;
    const int segs_per_sweep = 1024 / lanes_per_seg; // streams covered by one sweep of the workgroup
    const u64 wg_base = (u64)blockIdx.x * 256ULL * region;
    const u64 nseg_region = region / (u64)S;
    u64 acc = 0;
    u64 rpos = ((u64)blockIdx.x * 1024ULL + tid);
    const u64 rstride = (u64)gridDim.x * 1024ULL;
    float credit = 0.f;
    v2u64 p0 = {0, 0}, p1 = {0, 0}; // loads of the previous sweep: consumed one sweep later (software pipeline, as the real kernel prefetches)
    for (int d = 0; d < ndrain; d++) {
        for (int s0 = 0; s0 < 256; s0 += segs_per_sweep) {
            credit += rdw;
            const int n = (int)credit;
            credit -= (float)n;
            v2u64 q0 = {0, 0}, q1 = {0, 0};
            if (n >= 1) { q0 = __builtin_nontemporal_load(in + (rpos & (in_elems - 1))); rpos += rstride; }
            if (n >= 2) { q1 = __builtin_nontemporal_load(in + (rpos & (in_elems - 1))); rpos += rstride; }
            const int st = s0 + tid / lanes_per_seg;
            if (st < 256) {
                u64 segno = (u64)d;
                if (random) segno = mix((u64)d * 0x9E3779B97F4A7C15ULL + (u64)st + ((u64)blockIdx.x << 20)) % nseg_region;
                char *p = out + wg_base + (u64)st * region + segno * (u64)S + (u64)(tid % lanes_per_seg) * W;
                put<W, NT>(p, (u64)tid + (u64)d);
            }
            acc += p0.x ^ p0.y ^ p1.x ^ p1.y;
            p0 = q0;
            p1 = q1;
        }
    }
    acc += p0.x ^ p1.y;
    if (acc == 0x1234567ULL) sink[0] = acc;
}

template <int W, bool NT>
static void run(char *d_out, u64 out_bytes, const v2u64 *d_in, u64 in_elems, u64 *d_sink, int S, int random, double rd, double total_gb) {
    if (S < W || 1024 % (S / W) != 0 || (S / W) > 1024) return;
    const int nwg = 256;
    const u64 region = (out_bytes / (256ULL * nwg)) & ~4095ULL;
    int ndrain = (int)((total_gb * 1e9) / ((double)nwg * 256.0 * S));
    if ((u64)ndrain * (u64)S > region) ndrain = (int)(region / S);
    const double written = (double)nwg * 256.0 * S * ndrain;
    // reads: rd bytes per written byte; a sweep = 1024 lanes: each active lane writes W bytes and issues rd * W / 16 16-byte loads (<= 2)
    const int lanes_per_seg = S / W, segs_per_sweep = 1024 / lanes_per_seg;
    const int sweeps = segs_per_sweep >= 256 ? 1 : 256 / segs_per_sweep;
    const double active = segs_per_sweep >= 256 ? 256.0 * lanes_per_seg / 1024.0 : 1.0; // fraction of the 1024 lanes that store
    float rdw = (float)(rd * W * active / 16.0);
    if (rdw > 2.f) rdw = 2.f;
    const double read = (double)nwg * ndrain * sweeps * (double)rdw * 1024.0 * 16.0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_write<W, NT>), dim3(nwg), dim3(1024), 0, 0, d_out, region, S, ndrain, random, d_in, in_elems, rdw, d_sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("{\"probe\":\"write\",\"S\":%d,\"W\":%d,\"pattern\":\"%s\",\"nt\":%d,\"rd\":%.2f,\"ndrain\":%d,\"written_GB\":%.3f,\"read_GB\":%.3f,\"ms\":%.3f,\"write_TBps\":%.3f,\"total_TBps\":%.3f}\n",
           S, W, random ? "random" : "append", NT ? 1 : 0, rd, ndrain, written / 1e9, read / 1e9, best, written / best / 1e9, (written + read) / best / 1e9);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const double total_gb = argc > 1 ? atof(argv[1]) : 8.0;
    const u64 out_bytes = 16ULL << 30, in_bytes = 16ULL << 30;
    char *d_out;
    v2u64 *d_in;
    u64 *d_sink;
    CK(hipMalloc(&d_out, out_bytes));
    CK(hipMalloc(&d_in, in_bytes));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMemset(d_out, 0, out_bytes));
    CK(hipMemset(d_in, 1, in_bytes));
    const u64 in_elems = in_bytes / 16;
    const int Ss[] = {16, 32, 64, 128, 256, 512};
    for (int random = 0; random < 2; random++) {
        for (double rd : {0.0, 1.6}) {
            for (int S : Ss) {
                run<16, false>(d_out, out_bytes, d_in, in_elems, d_sink, S, random, rd, total_gb);
                run<8, false>(d_out, out_bytes, d_in, in_elems, d_sink, S, random, rd, total_gb);
                run<4, false>(d_out, out_bytes, d_in, in_elems, d_sink, S, random, rd, total_gb);
                if (S <= 128) run<2, false>(d_out, out_bytes, d_in, in_elems, d_sink, S, random, rd, total_gb);
            }
        }
    }
    // non-temporal stores on the interesting sizes
    for (int S : {32, 64, 128, 256}) {
        run<16, true>(d_out, out_bytes, d_in, in_elems, d_sink, S, 0, 1.6, total_gb);
        run<8, true>(d_out, out_bytes, d_in, in_elems, d_sink, S, 0, 1.6, total_gb);
        run<4, true>(d_out, out_bytes, d_in, in_elems, d_sink, S, 0, 1.6, total_gb);
    }
    return 0;
}
