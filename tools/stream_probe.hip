// tools/stream_probe.hip -- which launch shape gives the fastest single streaming pass on this part?  (VERDICT round 3,
// item 6: bench.py's calibration copy read 5.16 TB/s while the library's own in-place FFT kernels move 6.1-6.2.)
// One workgroup of 256 threads per contiguous chunk, buffer (SRD) nt loads all issued before the first nt store:
//   lanes of 8 or 16 bytes, U = 16 or 32 accesses per thread, in place (write back where it was read) or out of place,
//   1-4 workgroups per CU by launch bounds.  Footprint: `MiB` read (and as much written).
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int LANE, int U, int OCC>
__global__ __launch_bounds__(256, OCC) void k_stream(const char *a, char *b)
{
    constexpr uint32_t CHUNK = 256u * LANE * U;
    const uint64_t c = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * CHUNK, 0, CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * CHUNK, 0, CHUNK, 0x00020000);
    if constexpr (LANE == 8) {
        v2u x[U];
#pragma unroll
        for (int i = 0; i < U; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, threadIdx.x * 8, i * 2048, 2);
#pragma unroll
        for (int i = 0; i < U; ++i) __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, threadIdx.x * 8, i * 2048, 2);
    } else {
        v4u x[U];
#pragma unroll
        for (int i = 0; i < U; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, threadIdx.x * 16, i * 4096, 2);
#pragma unroll
        for (int i = 0; i < U; ++i) __builtin_amdgcn_raw_buffer_store_b128(x[i], rb, threadIdx.x * 16, i * 4096, 2);
    }
}
typedef void (*kern_t)(const char *, char *);
struct V { const char *name; kern_t k; uint32_t chunk; };
int main(int argc, char **argv)
{
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 16384ull) << 20;
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define VAR(L, U, O) {"lane " #L " B x " #U ", " #O " wg/SIMD-bound", k_stream<L, U, O>, 256u * L * U}
    V vs[] = {VAR(8, 16, 1), VAR(8, 16, 2), VAR(8, 32, 1), VAR(8, 32, 2), VAR(16, 8, 2), VAR(16, 16, 1), VAR(16, 16, 2), VAR(16, 32, 1)};
    printf("%-34s %12s %12s   (GB/s read+write, %llu MiB each way)\n", "variant", "in place", "out of place", (unsigned long long)(bytes >> 20));
    for (auto &v : vs) {
        float best[2] = {1e30f, 1e30f};
        for (int mode = 0; mode < 2; ++mode)
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3((uint32_t)(bytes / v.chunk)), dim3(256), 0, 0, a, mode ? b : a);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best[mode]) best[mode] = ms;
            }
        printf("%-34s %12.0f %12.0f\n", v.name, 2.0 * bytes / (best[0] * 1e-3) / 1e9, 2.0 * bytes / (best[1] * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
