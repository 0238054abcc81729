// tools/fabric_probe.hip -- measurement tool (not product code): what do the L2<->fabric paths sustain?
//   hipcc --offload-arch=gfx950 -O3 -o tools/fabric_probe tools/fabric_probe.hip
// Streams 16 B/lane reads and/or writes over a "big" (HBM-sized) and a "small" (Infinity-Cache-sized)
// region, alone and mixed, and prints GB/s for each mix.  Decides whether a cache-resident FFT
// intermediate is cheaper than an HBM one (DESIGN.md "fabric ceiling").
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// Each block walks `chunks` chunks of 4 KiB x UNROLL; mode bits: 1 = read big, 2 = write big,
// 4 = read small, 8 = write small.  Reads are summed into a register and conditionally stored (never true).
template <int UNROLL>
__global__ __launch_bounds__(256) void k_probe(const v4f *big_r, v4f *big_w, const v4f *small_r, v4f *small_w,
                                               uint64_t big_vecs, uint64_t small_vecs, uint64_t iters, int mode,
                                               float *sink)
{
    v4f acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * 256 * UNROLL;
    uint64_t i = (uint64_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    for (uint64_t it = 0; it < iters; ++it, i += stride) {
        const uint64_t ib = i & (big_vecs - 1), is = i & (small_vecs - 1);  // sizes are powers of two
        v4f a[UNROLL], b[UNROLL];
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) a[u] = big_r[ib + u * 256];
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) b[u] = small_r[is + u * 256];
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += a[u];
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += b[u];
        if (mode & 2)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) big_w[ib + u * 256] = acc + (float)u;
        if (mode & 8)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) small_w[is + u * 256] = acc + (float)u;
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

int main(int argc, char **argv)
{
    const uint64_t big_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 8192ull) << 20;   // MiB
    const uint64_t small_bytes = (argc > 2 ? strtoull(argv[2], 0, 10) : 64ull) << 20;   // MiB
    const int blocks = argc > 3 ? atoi(argv[3]) : 2048;
    v4f *big_a, *big_b, *small_a, *small_b;
    float *sink;
    const uint64_t pad = 1 << 20;  // the unrolled accesses run up to UNROLL*4 KiB past the wrapped index
    CK(hipMalloc(&big_a, big_bytes + pad));
    CK(hipMalloc(&big_b, big_bytes + pad));
    CK(hipMalloc(&small_a, small_bytes + pad));
    CK(hipMalloc(&small_b, small_bytes + pad));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(big_a, 1, big_bytes));
    CK(hipMemset(big_b, 1, big_bytes));
    CK(hipMemset(small_a, 1, small_bytes));
    CK(hipMemset(small_b, 1, small_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    constexpr int UNROLL = 4;
    const uint64_t per_iter_bytes = (uint64_t)blocks * 256 * UNROLL * 16;
    const uint64_t target = 16ull << 30;  // bytes per stream per timed launch
    const uint64_t iters = target / per_iter_bytes;
    struct Mode { int m; const char *name; } modes[] = {
        {1, "read big (HBM)"},      {2, "write big (HBM)"},        {3, "copy big->big (HBM)"},
        {4, "read small (cache)"},  {8, "write small (cache)"},    {12, "copy small->small (cache)"},
        {5, "read big + read small"}, {9, "read big + write small"}, {6, "write big + read small"},
        {15, "copy big + copy small (the two-pass FFT mix)"},
        {10, "write big + write small"},
    };
    printf("big %llu MiB, small %llu MiB, %d blocks, %llu iters\n", (unsigned long long)(big_bytes >> 20),
           (unsigned long long)(small_bytes >> 20), blocks, (unsigned long long)iters);
    for (auto &md : modes) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe<UNROLL>, dim3(blocks), dim3(256), 0, 0, big_a, big_b, small_a, small_b,
                               big_bytes / 16, small_bytes / 16, iters, md.m, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        int streams = __builtin_popcount(md.m);
        double gb = (double)streams * iters * per_iter_bytes / 1e9;
        printf("mode %2d  %-46s %8.3f ms  %8.1f GB/s total (%d streams, %.1f GB/s each)\n", md.m, md.name, best,
               gb / (best * 1e-3), streams, gb / (best * 1e-3) / streams);
        fflush(stdout);
    }
    return 0;
}
