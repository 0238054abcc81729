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
// Sweep patterns: a 256-thread workgroup owns CHUNK = 256 * 8 * U bytes; each WAVE owns a quarter of it and walks through that
// quarter front to back, PIECES pieces of 512 / PIECES bytes per instruction (PIECES = 2: the addressing of k_small32<10>, the
// library's fastest streaming kernel -- lanes 0-31 and 32-63 sweep two halves of the wave's range in 256-byte steps).
// The "linear" variants above make every instruction of the workgroup cover 2 KiB (4 KiB with 16-byte lanes) and hop on.
template <int U, int PIECES, int OCC>
__global__ __launch_bounds__(256, OCC) void k_sweep(const char *a, char *b)
{
    constexpr uint32_t CHUNK = 256u * 8 * U, WAVE = CHUNK / 4, SUB = WAVE / PIECES, STEP = 512 / PIECES;
    const uint64_t c = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * CHUNK, 0, CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * CHUNK, 0, CHUNK, 0x00020000);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t voff = w * WAVE + (lane / (64 / PIECES)) * SUB + (lane % (64 / PIECES)) * 8;
    v2u x[U];
#pragma unroll
    for (int i = 0; i < U; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, i * STEP, 2);
#pragma unroll
    for (int i = 0; i < U; ++i) __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, voff, i * STEP, 2);
}
// 8-byte lanes, any workgroup size: each wave walks U x 512 contiguous bytes
template <int U, int THREADS, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void k_sweep8(const char *a, char *b)
{
    constexpr uint32_t WAVE = 512u * U, CHUNK = WAVE * (THREADS / 64);
    const uint64_t c = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * CHUNK, 0, CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * CHUNK, 0, CHUNK, 0x00020000);
    const uint32_t voff = (threadIdx.x >> 6) * WAVE + (threadIdx.x & 63) * 8;
    v2u x[U];
#pragma unroll
    for (int i = 0; i < U; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, i * 512, 2);
#pragma unroll
    for (int i = 0; i < U; ++i) __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, voff, i * 512, 2);
}
// mixed: which side needs the walk?  LD_SWEEP / ST_SWEEP: that side walks 16 KiB per wave (512-byte steps), the other side hops
// through the 64-KiB chunk (every instruction of the workgroup covers 2 KiB); the values cross over through LDS-free register
// renaming only when both sides agree, so here the copy permutes data inside the chunk (same bytes moved).
template <bool LD_SWEEP, bool ST_SWEEP>
__global__ __launch_bounds__(256, 2) void k_mixed(const char *a, char *b)
{
    const uint64_t c = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * 65536, 0, 65536, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * 65536, 0, 65536, 0x00020000);
    const uint32_t sweep_off = (threadIdx.x >> 6) * 16384 + (threadIdx.x & 63) * 8, hop_off = threadIdx.x * 8;
    v2u x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, LD_SWEEP ? sweep_off : hop_off, i * (LD_SWEEP ? 512 : 2048), 2);
#pragma unroll
    for (int i = 0; i < 32; ++i) __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, ST_SWEEP ? sweep_off : hop_off, i * (ST_SWEEP ? 512 : 2048), 2);
}
// block -> chunk maps for the best shape (64-KiB chunk, each wave walks 16 KiB): MAP 0 identity (consecutive chunks go to
// consecutive XCDs), 1 every XCD owns a contiguous eighth of the buffer, 2 as 1 with the i-th and (i+32)-th workgroup of an XCD
// (the co-residents of a CU when there are two per CU) on adjacent chunks, 3 as 1 with 4 co-residents (i, i+32, i+64, i+96) adjacent
template <int MAP>
__global__ __launch_bounds__(256, 2) void k_mapped(const char *a, char *b)
{
    uint32_t blk = blockIdx.x;
    if (MAP >= 1) {
        const uint32_t per = gridDim.x >> 3, i = blk >> 3;
        uint32_t j = i;
        if (MAP == 2) { const uint32_t g = i & ~63u, r = i & 63u; j = g + (((r & 31u) << 1) | (r >> 5)); }
        if (MAP == 3) { const uint32_t g = i & ~127u, r = i & 127u; j = g + (((r & 31u) << 2) | (r >> 5)); }
        blk = (blk & 7u) * per + j;
    }
    const uint64_t c = blk;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * 65536, 0, 65536, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * 65536, 0, 65536, 0x00020000);
    const uint32_t voff = (threadIdx.x >> 6) * 16384 + (threadIdx.x & 63) * 8;
    v2u x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, i * 512, 2);
#pragma unroll
    for (int i = 0; i < 32; ++i) __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, voff, i * 512, 2);
}
// the same with 16-byte lanes: 1 KiB per wave instruction, U accesses -> each wave walks U KiB
template <int U, int THREADS, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void k_sweep16(const char *a, char *b)
{
    constexpr uint32_t WAVE = 1024u * U, CHUNK = WAVE * (THREADS / 64);
    const uint64_t c = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * CHUNK, 0, CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * CHUNK, 0, CHUNK, 0x00020000);
    const uint32_t voff = (threadIdx.x >> 6) * WAVE + (threadIdx.x & 63) * 16;
    v4u x[U];
#pragma unroll
    for (int i = 0; i < U; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff, i * 1024, 2);
#pragma unroll
    for (int i = 0; i < U; ++i) __builtin_amdgcn_raw_buffer_store_b128(x[i], rb, voff, i * 1024, 2);
}
typedef void (*kern_t)(const char *, char *);
struct V { const char *name; kern_t k; uint32_t chunk; uint32_t threads = 256; };
int main(int argc, char **argv)
{
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 16384ull) << 20;
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define VAR(L, U, O) {"lane " #L " B x " #U ", " #O " wg/SIMD-bound", k_stream<L, U, O>, 256u * L * U}
    V vs[] = {VAR(8, 16, 1), VAR(8, 16, 2), VAR(8, 32, 1), VAR(8, 32, 2), VAR(16, 8, 2), VAR(16, 16, 1), VAR(16, 16, 2), VAR(16, 32, 1),
#define SW(U, P, O) {"sweep per wave, x " #U ", " #P " piece(s), occ " #O, k_sweep<U, P, O>, 256u * 8 * U}
              SW(16, 1, 2), SW(16, 2, 2), SW(16, 4, 2), SW(32, 1, 2), SW(32, 2, 1), SW(32, 2, 2), SW(32, 2, 4), SW(32, 4, 2), SW(32, 8, 2), SW(64, 2, 1), SW(64, 4, 1),
#define S16(U, T, O) {"sweep16 per wave, x " #U ", " #T " threads, occ " #O, k_sweep16<U, T, O>, 1024u * U * (T / 64), T}
              S16(8, 256, 2), S16(16, 256, 2), S16(32, 256, 1), S16(16, 128, 2), S16(16, 512, 1), S16(16, 64, 4), S16(24, 256, 1),
#define S8(U, T, O) {"sweep8 per wave, x " #U ", " #T " threads, occ " #O, k_sweep8<U, T, O>, 512u * U * (T / 64), T}
              S8(16, 512, 1), S8(16, 512, 2), S8(16, 1024, 1), S8(32, 128, 4), S8(32, 512, 1), S8(8, 1024, 1),
              {"loads walk, stores hop", k_mixed<true, false>, 65536u}, {"loads hop, stores walk", k_mixed<false, true>, 65536u},
              {"both hop (64 KiB chunk)", k_mixed<false, false>, 65536u}, {"both walk", k_mixed<true, true>, 65536u},
              {"walk, map 0: identity", k_mapped<0>, 65536u}, {"walk, map 1: XCD-contiguous", k_mapped<1>, 65536u},
              {"walk, map 2: + CU pairs adjacent", k_mapped<2>, 65536u}, {"walk, map 3: + CU quads adjacent", k_mapped<3>, 65536u}};
    printf("%-34s %12s %12s   (GB/s read+write, %llu MiB each way)\n", "variant", "in place", "out of place", (unsigned long long)(bytes >> 20));
    for (auto &v : vs) {
        float best[2] = {1e30f, 1e30f};
        for (int mode = 0; mode < 2; ++mode)
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3((uint32_t)(bytes / v.chunk)), dim3(v.threads), 0, 0, a, mode ? b : a);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best[mode]) best[mode] = ms;
            }
        printf("%-34s %12.0f %12.0f\n", v.name, 2.0 * bytes / (best[0] * 1e-3) / 1e9, 2.0 * bytes / (best[1] * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
