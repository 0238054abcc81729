// tools/fabric_probe4.hip -- measurement tool (round 3): does the LANE WIDTH of the accesses (8 B = one complex sample,
// as every FFT kernel here issues, against 16 B) move the linear-stream ceilings of the two pass mixes?
//   hipcc --offload-arch=gfx950 -O3 -o tools/fabric_probe4 tools/fabric_probe4.hip
// "rd big+wr small" = pass 1 (HBM read nt + ring write), "wr big+rd small" = pass 2 (ring read + HBM write nt).
// Each wave instruction covers 64 lanes x LW bytes of contiguous memory in every variant.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<16> { typedef v4u T; };
template <> struct Vec<8> { typedef v2u T; };
template <int W, int AUX> __device__ __forceinline__ v4u ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s)
{
    if constexpr (W == 16) return __builtin_amdgcn_raw_buffer_load_b128(r, v, s, AUX);
    else { v2u a = __builtin_amdgcn_raw_buffer_load_b64(r, v, s, AUX); return v4u{a.x, a.y, 0, 0}; }
}
template <int W, int AUX> __device__ __forceinline__ void st(v4u d, __amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s)
{
    if constexpr (W == 16) __builtin_amdgcn_raw_buffer_store_b128(d, r, v, s, AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(v2u{d.x, d.y}, r, v, s, AUX);
}

// mode bits: 1 read big, 2 write big, 4 read small, 8 write small.  U accesses per stream per iteration; a block moves
// 256 * U * W bytes per stream per iteration.
template <int LW, int SW, int LB, int SB, int LS, int SS, int U>
__global__ __launch_bounds__(256) void k_probe(void *big_r, void *big_w, void *small_r, void *small_w,
                                               uint32_t big_bytes, uint32_t small_bytes, uint32_t iters, int mode,
                                               unsigned *sink)
{
    static_assert(LW * U == SW * (U * LW / SW), "");
    constexpr int US = U * LW / SW;  // stores per iteration so that bytes in = bytes out
    auto rbr = __builtin_amdgcn_make_buffer_rsrc(big_r, 0, big_bytes + (1 << 20), 0x00020000);
    auto rbw = __builtin_amdgcn_make_buffer_rsrc(big_w, 0, big_bytes + (1 << 20), 0x00020000);
    auto rsr = __builtin_amdgcn_make_buffer_rsrc(small_r, 0, small_bytes + (1 << 20), 0x00020000);
    auto rsw = __builtin_amdgcn_make_buffer_rsrc(small_w, 0, small_bytes + (1 << 20), 0x00020000);
    v4u acc = {0, 0, 0, 0};
    const uint32_t chunk = 256 * U * LW;  // bytes per block per iteration
    const uint32_t stride = gridDim.x * chunk;
    uint32_t base = blockIdx.x * chunk;
    for (uint32_t it = 0; it < iters; ++it, base += stride) {
        const uint32_t bb = base & (big_bytes - 1), bs = base & (small_bytes - 1);
        v4u a[U], b[U];
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = ld<LW, LB>(rbr, bb + threadIdx.x * LW, u * 256 * LW);
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = ld<LW, LS>(rsr, bs + threadIdx.x * LW, u * 256 * LW);
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < U; ++u) acc += a[u];
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < U; ++u) acc += b[u];
        if (mode & 2)
#pragma unroll
            for (int u = 0; u < US; ++u) st<SW, SB>(acc + (unsigned)u, rbw, bb + threadIdx.x * SW, u * 256 * SW);
        if (mode & 8)
#pragma unroll
            for (int u = 0; u < US; ++u) st<SW, SS>(acc + (unsigned)u, rsw, bs + threadIdx.x * SW, u * 256 * SW);
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y + acc.z + acc.w;
}

typedef void (*kern_t)(void *, void *, void *, void *, uint32_t, uint32_t, uint32_t, int, unsigned *);
struct Variant { const char *name; kern_t k; int lw, u; };

int main(int argc, char **argv)
{
    const uint32_t big_bytes = 1u << 31;
    const uint32_t small_bytes = (uint32_t)((argc > 1 ? strtoull(argv[1], 0, 10) : 256ull) << 20);
    const int blocks = argc > 2 ? atoi(argv[2]) : 2048;
    void *big_a, *big_b, *small_a, *small_b; unsigned *sink;
    CK(hipMalloc(&big_a, (size_t)big_bytes + (1 << 20))); CK(hipMalloc(&big_b, (size_t)big_bytes + (1 << 20)));
    CK(hipMalloc(&small_a, (size_t)small_bytes + (1 << 20))); CK(hipMalloc(&small_b, (size_t)small_bytes + (1 << 20)));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(big_a, 1, big_bytes)); CK(hipMemset(big_b, 1, big_bytes));
    CK(hipMemset(small_a, 1, small_bytes)); CK(hipMemset(small_b, 1, small_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // policies: big = nt (2) loads and stores; small loads default (0); small stores sc1 (16) / plain (0) / nt (2)
    Variant vs[] = {
        {"ld16 st16, small st sc1  (U=4)  ", k_probe<16, 16, 2, 2, 0, 16, 4>, 16, 4},
        {"ld16 st16, small st plain       ", k_probe<16, 16, 2, 2, 0, 0, 4>, 16, 4},
        {"ld8  st8 , small st sc1  (U=8)  ", k_probe<8, 8, 2, 2, 0, 16, 8>, 8, 8},
        {"ld8  st8 , small st plain       ", k_probe<8, 8, 2, 2, 0, 0, 8>, 8, 8},
        {"ld8  st8 , small st nt          ", k_probe<8, 8, 2, 2, 0, 2, 8>, 8, 8},
        {"ld8  st16, small st sc1         ", k_probe<8, 16, 2, 2, 0, 16, 8>, 8, 8},
        {"ld16 st8 , small st sc1         ", k_probe<16, 8, 2, 2, 0, 16, 4>, 16, 4},
        {"ld16 st8 , small st plain       ", k_probe<16, 8, 2, 2, 0, 0, 4>, 16, 4},
        {"ld8  st8 , small st sc1  (U=32) ", k_probe<8, 8, 2, 2, 0, 16, 32>, 8, 32},
        {"ld8  st8 , small st plain(U=32) ", k_probe<8, 8, 2, 2, 0, 0, 32>, 8, 32},
        {"ld8  st16, small st sc1  (U=32) ", k_probe<8, 16, 2, 2, 0, 16, 32>, 8, 32},
        {"ld16 st16, small st sc1  (U=16) ", k_probe<16, 16, 2, 2, 0, 16, 16>, 16, 16},
    };
    int modes[] = {9, 6, 15, 3, 1, 8};
    const char *mn[] = {"rd big+wr small", "wr big+rd small", "FFT mix(15)", "copy big", "rd big", "wr small"};
    printf("small (ring) %u MiB, big 2048 MiB, %d blocks; GB/s total per mode\n%-36s", small_bytes >> 20, blocks, "variant");
    for (auto m : mn) printf("%18s", m);
    printf("\n");
    for (auto &v : vs) {
        const uint64_t per_iter = (uint64_t)blocks * 256 * v.u * v.lw;
        const uint32_t iters = (uint32_t)((8ull << 30) / per_iter);
        printf("%-36s", v.name);
        for (int mi = 0; mi < 6; ++mi) {
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
