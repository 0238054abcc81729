// tools/fabric_probe2.hip -- measurement tool: does a cache policy (nt / sc1 / sc0) on loads or stores
// lift the mixed HBM + Infinity-Cache traffic ceiling found by fabric_probe.hip?
//   hipcc --offload-arch=gfx950 -O3 -o tools/fabric_probe2 tools/fabric_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode bits: 1 read big, 2 write big, 4 read small, 8 write small.  Regions are <= 2 GiB (32-bit buffer offsets).
template <int LB, int SB, int LS, int SS>
__global__ __launch_bounds__(256) void k_probe(void *big_r, void *big_w, void *small_r, void *small_w,
                                               uint32_t big_bytes, uint32_t small_bytes, uint32_t iters, int mode,
                                               unsigned *sink)
{
    constexpr int U = 4;
    auto rbr = __builtin_amdgcn_make_buffer_rsrc(big_r, 0, big_bytes + (1 << 20), 0x00020000);
    auto rbw = __builtin_amdgcn_make_buffer_rsrc(big_w, 0, big_bytes + (1 << 20), 0x00020000);
    auto rsr = __builtin_amdgcn_make_buffer_rsrc(small_r, 0, small_bytes + (1 << 20), 0x00020000);
    auto rsw = __builtin_amdgcn_make_buffer_rsrc(small_w, 0, small_bytes + (1 << 20), 0x00020000);
    v4u acc = {0, 0, 0, 0};
    const uint32_t stride = gridDim.x * 256 * U * 16;
    uint32_t i = (blockIdx.x * 256 * U + threadIdx.x) * 16;
    for (uint32_t it = 0; it < iters; ++it, i += stride) {
        const uint32_t ib = i & (big_bytes - 1), is = i & (small_bytes - 1);
        v4u a[U], b[U];
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = __builtin_amdgcn_raw_buffer_load_b128(rbr, ib, u * 4096, LB);
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = __builtin_amdgcn_raw_buffer_load_b128(rsr, is, u * 4096, LS);
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < U; ++u) acc += a[u];
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < U; ++u) acc += b[u];
        if (mode & 2)
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)u, rbw, ib, u * 4096, SB);
        if (mode & 8)
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)u, rsw, is, u * 4096, SS);
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y + acc.z + acc.w;
}

typedef void (*kern_t)(void *, void *, void *, void *, uint32_t, uint32_t, uint32_t, int, unsigned *);
struct Variant { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    const uint32_t big_bytes = 1u << 31;                                                  // 2 GiB
    const uint32_t small_bytes = (uint32_t)((argc > 1 ? strtoull(argv[1], 0, 10) : 64ull) << 20);
    const int blocks = argc > 2 ? atoi(argv[2]) : 2048;
    void *big_a, *big_b, *small_a, *small_b; unsigned *sink;
    CK(hipMalloc(&big_a, (size_t)big_bytes + (1 << 20))); CK(hipMalloc(&big_b, (size_t)big_bytes + (1 << 20)));
    CK(hipMalloc(&small_a, (size_t)small_bytes + (1 << 20))); CK(hipMalloc(&small_b, (size_t)small_bytes + (1 << 20)));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(big_a, 1, big_bytes)); CK(hipMemset(big_b, 1, big_bytes));
    CK(hipMemset(small_a, 1, small_bytes)); CK(hipMemset(small_b, 1, small_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t per_iter = (uint64_t)blocks * 256 * 4 * 16;
    const uint32_t iters = (uint32_t)((8ull << 30) / per_iter);
    // aux: 0 default, 1 sc0, 2 nt, 16 sc1, 17 sc0|sc1, 18 sc1|nt, 3 sc0|nt
    Variant vs[] = {
        {"default everywhere            ", k_probe<0, 0, 0, 0>},
        {"nt big loads                  ", k_probe<2, 0, 0, 0>},
        {"nt big stores                 ", k_probe<0, 2, 0, 0>},
        {"nt big loads+stores           ", k_probe<2, 2, 0, 0>},
        {"nt everywhere                 ", k_probe<2, 2, 2, 2>},
        {"sc1 small stores              ", k_probe<0, 0, 0, 16>},
        {"sc0sc1 small stores           ", k_probe<0, 0, 0, 17>},
        {"sc1 small loads+stores        ", k_probe<0, 0, 16, 16>},
        {"nt big, sc1 small             ", k_probe<2, 2, 16, 16>},
        {"sc1 big stores                ", k_probe<0, 16, 0, 0>},
        {"sc0sc1 big stores             ", k_probe<0, 17, 0, 0>},
        {"nt big ld, sc0sc1 big st      ", k_probe<2, 17, 0, 0>},
        {"sc1 nt big stores             ", k_probe<0, 18, 0, 0>},
        {"sc1|nt big ld, nt st, sc1 sm  ", k_probe<18, 2, 16, 16>},
        {"sc0sc1nt big ld, nt st, sc1 sm", k_probe<19, 2, 16, 16>},
        {"sc0sc1 big ld, nt st, sc1 sm  ", k_probe<17, 2, 16, 16>},
        {"sc0|nt big ld, nt st, sc1 sm  ", k_probe<3, 2, 16, 16>},
        {"sc1 big ld, nt st, sc1 sm     ", k_probe<16, 2, 16, 16>},
        {"nt big, sc1|nt small st       ", k_probe<2, 2, 16, 18>},
        {"nt big, sc0sc1nt small st     ", k_probe<2, 2, 16, 19>},
        {"nt big, plain small           ", k_probe<2, 2, 0, 0>},
    };
    int modes[] = {3, 12, 15, 9, 6};
    const char *mn[] = {"copy big", "copy small", "FFT mix(15)", "rd big+wr small", "wr big+rd small"};
    printf("small %u MiB, %d blocks; GB/s total per mode\n%-32s", small_bytes >> 20, blocks, "variant");
    for (auto m : mn) printf("%18s", m);
    printf("\n");
    for (auto &v : vs) {
        printf("%-32s", v.name);
        for (int mi = 0; mi < 5; ++mi) {
            float best = 1e30f;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, big_a, big_b, small_a, small_b, big_bytes, small_bytes,
                                   iters, modes[mi], sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double gb = (double)__builtin_popcount(modes[mi]) * iters * per_iter / 1e9;
            printf("%18.0f", gb / (best * 1e-3));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
