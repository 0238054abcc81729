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
// normalize: a streaming pass in the launch shape that measured fastest on this part (tools/stream_probe.hip,
// profiles/round4/probe_stream_shapes.txt): one 256-thread workgroup per contiguous, 64-KiB-aligned chunk, every WAVE walks its
// own 16 KiB front to back -- 32 `nt` loads of 512 contiguous bytes per wave, all in flight before the first store.  That is the
// addressing of k_small32 (6.2-6.4 TB/s); chunks of 32 KiB, or instructions that cover 2 KiB of the workgroup's chunk and hop
// on (round 2's shape), stay at 5.5-5.9 TB/s.  The descriptor ends with the data, so a ragged last chunk needs no bounds code.
__global__ __launch_bounds__(256) void k_scale(const v2f *__restrict__ a, v2f *__restrict__ b, uint64_t n_samples,
                                               float scale)
{
    constexpr uint32_t CH = 8192;  // samples per workgroup
    const uint64_t e0 = (uint64_t)one_launch_block() * CH;
    const uint64_t left = n_samples - e0;
    const uint32_t valid = left < CH ? (uint32_t)left * 8u : CH * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(a + e0), 0, valid, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(b + e0, 0, valid, 0x00020000);
    const uint32_t voff = (threadIdx.x >> 6) * 16384 + (threadIdx.x & 63) * 8;
    v2f x[32];
    static_for<0, 32>([&](auto u_) { constexpr int u = decltype(u_)::value; x[u] = buf_load<AUX_NT>(rin, voff, u * 512); });
    static_for<0, 32>([&](auto u_) { constexpr int u = decltype(u_)::value; buf_store<AUX_NT>(x[u] * scale, rout, voff, u * 512); });
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
    const uint64_t blocks = (n_samples + 8191) / 8192;
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
// to check that the internal chain streams of the pipelined paths really run kernels side by side (ctx_streams.cpp: chain_streams).
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

// Calibration copy in the same shape as k_scale above: one workgroup per 64-KiB chunk, each wave walks 16 KiB with 32 `nt`
// 8-byte-lane loads in flight before the first store (6.3-6.4 TB/s; a grid-stride float4 loop with default policy: 4.7-5.0,
// chunk shapes whose instructions hop through the chunk: 5.5-5.9 -- profiles/round4/probe_stream_shapes.txt).
__global__ __launch_bounds__(256) void k_copy(const char *__restrict__ a, char *__restrict__ b, uint64_t n_chunks)
{
    constexpr uint32_t CHUNK = 65536, U = 32;
    const uint64_t c = one_launch_block();
    if (c >= n_chunks) return;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a) + c * CHUNK, 0, CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + c * CHUNK, 0, CHUNK, 0x00020000);
    const uint32_t voff = (threadIdx.x >> 6) * 16384 + (threadIdx.x & 63) * 8;
    v2u x[U];
    static_for<0, U>([&](auto i_) { constexpr int i = decltype(i_)::value; x[i] = __builtin_amdgcn_raw_buffer_load_b64(ra, voff, i * 512, AUX_NT); });
    static_for<0, U>([&](auto i_) { constexpr int i = decltype(i_)::value; __builtin_amdgcn_raw_buffer_store_b64(x[i], rb, voff, i * 512, AUX_NT); });
}

__global__ __launch_bounds__(256) void k_copy_tail(const v4f *__restrict__ a, v4f *__restrict__ b, uint64_t first, uint64_t n_vec)
{
    const uint64_t i = first + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_vec) b[i] = a[i];
}

hipError_t launch_copy(const void *src, void *dst, uint64_t bytes, hipStream_t st)
{
    const uint64_t n_chunks = bytes / 65536, n_vec = bytes / 16;
    if (n_vec == 0) return hipSuccess;
    if (n_chunks > 0x7fffffffull) return hipErrorInvalidValue;
    if (n_chunks)
        hipLaunchKernelGGL(k_copy, dim3((uint32_t)n_chunks), dim3(256), 0, st, static_cast<const char *>(src),
                           static_cast<char *>(dst), n_chunks);
    const uint64_t done = n_chunks * 4096;  // 16-byte vectors copied by the chunk kernel
    if (n_vec > done)
        hipLaunchKernelGGL(k_copy_tail, dim3((uint32_t)((n_vec - done + 255) / 256)), dim3(256), 0, st,
                           static_cast<const v4f *>(src), static_cast<v4f *>(dst), done, n_vec);
    return hipGetLastError();
}

}  // namespace fwa
