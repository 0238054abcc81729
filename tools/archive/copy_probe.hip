// tools/copy_probe.hip -- measurement tool: how fast can a plain HBM->HBM copy go on this part, and with which
// shape?  (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; bench.py's calibration copy gets 4.8.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: grid-stride, U consecutive 4-KiB rows per thread-iteration (a[i + u*blockDim]);
// MODE 1: each block owns one contiguous chunk of the buffer and walks it linearly;
// NT: nontemporal loads/stores.
template <int U, int MODE, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f *__restrict__ a, v4f *__restrict__ b, uint64_t n)
{
    uint64_t i, end, step;
    if (MODE == 0) { i = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; end = n; step = (uint64_t)gridDim.x * 256 * U; }
    else { const uint64_t per = n / gridDim.x; i = blockIdx.x * per + threadIdx.x; end = (blockIdx.x + 1) * per; step = 256 * U; }
    for (; i + (U - 1) * 256 < end; i += step) {
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(x[u], b + i + u * 256); else b[i + u * 256] = x[u]; }
    }
}
typedef void (*kern_t)(const v4f *, v4f *, uint64_t);
struct V { const char *name; kern_t k; };
int main(int argc, char **argv)
{
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 8192ull) << 20;
    v4f *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    V vs[] = {
        {"grid-stride U1      ", k_copy<1, 0, false>}, {"grid-stride U2      ", k_copy<2, 0, false>},
        {"grid-stride U4      ", k_copy<4, 0, false>}, {"grid-stride U8      ", k_copy<8, 0, false>},
        {"grid-stride U4 nt   ", k_copy<4, 0, true>},  {"grid-stride U8 nt   ", k_copy<8, 0, true>},
        {"block-chunk U4      ", k_copy<4, 1, false>}, {"block-chunk U8      ", k_copy<8, 1, false>},
        {"block-chunk U4 nt   ", k_copy<4, 1, true>},
    };
    int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384, 65536};
    printf("%-22s", "variant \\ blocks");
    for (int g : grids) printf("%8d", g);
    printf("   (GB/s, read+write, %llu MiB each way)\n", (unsigned long long)(bytes >> 20));
    for (auto &v : vs) {
        printf("%-22s", v.name);
        for (int g : grids) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3(g), dim3(256), 0, 0, a, b, bytes / 16);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%8.0f", 2.0 * bytes / (best * 1e-3) / 1e9);
        }
        printf("\n"); fflush(stdout);
    }
    // hipMemcpy DtoD for reference
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("hipMemcpyAsync DtoD: %.0f GB/s\n", 2.0 * bytes / (best * 1e-3) / 1e9);
    return 0;
}
