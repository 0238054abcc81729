// kernels_small.hip -- the literal radix-2 stage (cross-check path), elementwise kernels, synthetic fill, copy, spin.
// kernels_*.hip -- hand-written gfx950 kernels of the batched 1-D c2c fp32 FFT.
//
// Reference mapping (all under /root/reference/src):
//   k_r2_stage      <- kernel/fft.wgsl:27-62, ifft.wgsl:25-75 (one butterfly per thread, one launch per stage)
//   k_scale         <- kernel/normalize.wgsl:9-12
// (The LDS radix-2 kernel and the direct-addressing 16-point kernels that used to live here measured slower than
// k_chunk / k_small32 and moved to the laboratory build: kernels_lab_small.hip.)
#include "device_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// radix-2 Stockham stage in global memory (generic fallback, any power of two)
// ---------------------------------------------------------------------------
template <int DIR>
__global__ __launch_bounds__(256) void k_r2_stage(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                  const v2f *__restrict__ tw, uint32_t n, uint32_t lg_half,
                                                  uint32_t stage, uint64_t total_bf, float scale)
{
    const uint32_t half = n >> 1;
    const uint32_t J = 1u << stage;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_bf; idx += stride) {
        const uint64_t t = idx >> lg_half;
        const uint32_t i = (uint32_t)(idx & (half - 1));
        const uint32_t j = i & (J - 1);
        const uint32_t sJ = i - j;  // block_idx * J  (fft.wgsl:37 twiddles[s*J])
        const uint64_t base = t * (uint64_t)n;
        const v2f a = src[base + i];
        const v2f b = src[base + i + half];
        const v2f w = tw[sJ];
        const uint64_t o1 = base + ((uint64_t)sJ << 1) + j;
        dst[o1] = (a + b) * scale;
        dst[o1 + J] = cmul_tw<DIR>(a - b, w) * scale;
    }
}

hipError_t launch_r2_stage(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint32_t stage,
                           uint64_t batch, float scale, hipStream_t st)
{
    const uint64_t total = batch * (uint64_t)(n >> 1);
    if (total == 0) return hipSuccess;
    uint32_t lg_half = 0;
    while ((1u << lg_half) < (n >> 1)) ++lg_half;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > (1u << 20)) blocks = (1u << 20);
    if (dir == FWD)
        hipLaunchKernelGGL(k_r2_stage<FWD>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, tw, n, lg_half, stage,
                           total, scale);
    else
        hipLaunchKernelGGL(k_r2_stage<INV>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, tw, n, lg_half, stage,
                           total, scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// elementwise: normalize (normalize.wgsl:9-12), synthetic fill, calibration copy
// ---------------------------------------------------------------------------
// normalize: the shape of the one-launch FFT kernels (k_chunk): one workgroup per contiguous 32-KiB chunk, linear `nt`
// buffer accesses, all 16 loads of a thread in flight before the first store; the descriptor ends with the data, so
// a ragged last chunk needs no bounds code.  (A grid-stride float4 loop with default policy: 0.55-0.63 of the
// roofline; this shape: the streaming rate of the FFT kernels.)
__global__ __launch_bounds__(256) void k_scale(const v2f *__restrict__ a, v2f *__restrict__ b, uint64_t n_samples,
                                               float scale)
{
    constexpr uint32_t CH = 4096;  // samples per workgroup
    const uint64_t e0 = (uint64_t)blockIdx.x * CH;
    const uint64_t left = n_samples - e0;
    const uint32_t valid = left < CH ? (uint32_t)left * 8u : CH * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(a + e0), 0, valid, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(b + e0, 0, valid, 0x00020000);
    v2f x[16];
    static_for<0, 16>([&](auto u_) { constexpr int u = decltype(u_)::value; x[u] = buf_load<AUX_NT>(rin, threadIdx.x * 8, u * 2048); });
    static_for<0, 16>([&](auto u_) { constexpr int u = decltype(u_)::value; buf_store<AUX_NT>(x[u] * scale, rout, threadIdx.x * 8, u * 2048); });
}

static uint32_t stream_grid(uint64_t work_items)
{
    uint64_t blocks = (work_items + 255) / 256;
    const uint64_t cap = 256 * 8 * 4;  // ~8192 blocks, grid-stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (uint32_t)blocks;
}

hipError_t launch_scale(const v2f *a, v2f *b, uint64_t n_samples, float scale, hipStream_t st)
{
    if (n_samples == 0) return hipSuccess;
    const uint64_t blocks = (n_samples + 4095) / 4096;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scale, dim3((uint32_t)blocks), dim3(256), 0, st, a, b, n_samples, scale);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_fill(v2f *__restrict__ dst, uint64_t seed, uint64_t g0, uint64_t n_samples,
                                              float scale)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += stride)
        dst[i] = gen_sample(seed, g0 + i, scale);
}

hipError_t launch_fill(v2f *dst, uint64_t seed, uint64_t g0, uint64_t n_samples, float scale, hipStream_t st)
{
    if (n_samples == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill, dim3(stream_grid(n_samples)), dim3(256), 0, st, dst, seed, g0, n_samples, scale);
    return hipGetLastError();
}

// A kernel that occupies `blocks` one-wave workgroups for `ticks` x 10 ns and touches no memory: used once per context
// to check that the internal chain streams of the pipelined paths really run kernels side by side (api.cpp: chain_streams).
__global__ __launch_bounds__(64) void k_spin(uint32_t ticks)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
hipError_t launch_spin(uint32_t ticks, uint32_t blocks, hipStream_t st)
{
    if (ticks > 100000u) ticks = 100000u;  // 1 ms at most: the grid always drains
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, st, ticks);
    return hipGetLastError();
}

// Calibration copy, shaped like the one-launch FFT kernels: one workgroup per contiguous 32-KiB chunk, every thread
// issues its 16 non-temporal 8-byte loads (one complex sample each, as every FFT kernel does) before the first store.
// tools/stream_probe.hip (profiles/round4/probe_stream_shapes.txt) compares lanes of 8 / 16 bytes x 8 / 16 / 32 accesses
// per thread, in and out of place: 5.5-5.9 TB/s all of them, this shape on top; a grid-stride float4 loop with default
// policy reaches 4.7-5.0 (profiles/round2/probe_copy_shapes.txt).
__global__ __launch_bounds__(256) void k_copy(const char *__restrict__ a, char *__restrict__ b, uint64_t n_chunks)
{
    constexpr uint32_t CHUNK = 32768, U = 16;
    const uint64_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * CHUNK, 0, CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * CHUNK, 0, CHUNK, 0x00020000);
    v2u x[U];
    static_for<0, U>([&](auto i_) { constexpr int i = decltype(i_)::value; x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, threadIdx.x * 8, i * 2048, AUX_NT); });
    static_for<0, U>([&](auto i_) { constexpr int i = decltype(i_)::value; __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, threadIdx.x * 8, i * 2048, AUX_NT); });
}

__global__ __launch_bounds__(256) void k_copy_tail(const v4f *__restrict__ a, v4f *__restrict__ b, uint64_t first, uint64_t n_vec)
{
    const uint64_t i = first + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_vec) b[i] = a[i];
}

hipError_t launch_copy(const void *src, void *dst, uint64_t bytes, hipStream_t st)
{
    const uint64_t n_chunks = bytes / 32768, n_vec = bytes / 16;
    if (n_vec == 0) return hipSuccess;
    if (n_chunks > 0x7fffffffull) return hipErrorInvalidValue;
    if (n_chunks)
        hipLaunchKernelGGL(k_copy, dim3((uint32_t)n_chunks), dim3(256), 0, st, static_cast<const char *>(src),
                           static_cast<char *>(dst), n_chunks);
    const uint64_t done = n_chunks * 2048;  // 16-byte vectors copied by the chunk kernel
    if (n_vec > done)
        hipLaunchKernelGGL(k_copy_tail, dim3((uint32_t)((n_vec - done + 255) / 256)), dim3(256), 0, st,
                           static_cast<const v4f *>(src), static_cast<v4f *>(dst), done, n_vec);
    return hipGetLastError();
}

}  // namespace fwa
