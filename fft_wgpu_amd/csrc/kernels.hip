// kernels.hip -- hand-written gfx950 kernels of the batched 1-D c2c fp32 FFT.
//
// Reference mapping (all under /root/reference/src):
//   k_r2_stage      <- kernel/fft.wgsl:27-62, ifft.wgsl:25-75 (one butterfly per thread, one launch per stage)
//   k_lds_small     <- kernel/fft4.wgsl:13-112 (one dispatch, all stages) staged in LDS as kernel/fft2.wgsl:9-10 intended
//   k_p1_1m/k_p2_1m <- kernel/fft4.wgsl at fft_len = 2^20 (config C2/C3), re-designed: two LDS-tiled
//                      passes of 32x32 register FFTs, intermediate in a small cache-resident ring
//   k_normalize     <- kernel/normalize.wgsl:9-12
// Wavefront = 64, 16-waves-per-CU residency (2 x 512-thread workgroups) for the 2^20 passes.
#include "kernels.h"

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
// small transforms (n <= 4096): whole transforms staged in LDS, all stages in one launch
// ---------------------------------------------------------------------------
template <int DIR>
__global__ __launch_bounds__(256) void k_lds_small(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                   const v2f *__restrict__ tw, uint32_t lg_n, uint32_t lg_p,
                                                   uint64_t batch, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t n = 1u << lg_n;
    const uint32_t P = 1u << lg_p;  // points per workgroup (>= n)
    v2f *bufA = reinterpret_cast<v2f *>(smem);
    v2f *bufB = bufA + P;
    const uint32_t tpb = P >> lg_n;  // transforms per block
    const uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t remaining = batch - t0;  // > 0 by grid construction
    const uint32_t valid = (uint32_t)((remaining < tpb ? remaining : tpb) << lg_n);  // valid points in this block
    const v2f *g_in = src + t0 * n;
    v2f *g_out = dst + t0 * n;

    for (uint32_t p = threadIdx.x; p < P; p += 256) bufA[p] = (p < valid) ? g_in[p] : v2f{0.f, 0.f};
    __syncthreads();

    const uint32_t half = n >> 1;
    v2f *a = bufA, *b = bufB;
    for (uint32_t stage = 0; stage < lg_n; ++stage) {
        const uint32_t J = 1u << stage;
        for (uint32_t idx = threadIdx.x; idx < (P >> 1); idx += 256) {
            const uint32_t tl = idx >> (lg_n - 1);
            const uint32_t i = idx & (half - 1);
            const uint32_t j = i & (J - 1);
            const uint32_t sJ = i - j;
            const uint32_t base = tl << lg_n;
            const v2f x = a[base + i], y = a[base + i + half];
            const v2f w = tw[sJ];
            const uint32_t o1 = base + (sJ << 1) + j;
            b[o1] = x + y;
            b[o1 + J] = cmul_tw<DIR>(x - y, w);
        }
        __syncthreads();
        v2f *t = a; a = b; b = t;
    }
    for (uint32_t p = threadIdx.x; p < valid; p += 256) g_out[p] = a[p] * scale;
}

hipError_t launch_lds_small(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                            hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    uint32_t lg_n = 0;
    while ((1u << lg_n) < n) ++lg_n;
    const uint32_t lg_p = lg_n < 11 ? 11 : lg_n;  // 2048 points per block, 4096 for n = 4096
    const uint32_t tpb = 1u << (lg_p - lg_n);
    const uint64_t blocks = (batch + tpb - 1) / tpb;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * sizeof(v2f) << lg_p;
    if (dir == FWD)
        hipLaunchKernelGGL(k_lds_small<FWD>, dim3((uint32_t)blocks), dim3(256), lds, st, src, dst, tw, lg_n, lg_p, batch,
                           scale);
    else
        hipLaunchKernelGGL(k_lds_small<INV>, dim3((uint32_t)blocks), dim3(256), lds, st, src, dst, tw, lg_n, lg_p, batch,
                           scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// n = 2^20 = 1024 x 1024, two passes.
//
// Index algebra (n = 1024*n1 + n2, k = K1 + 1024*K2):
//   X[K1 + 1024 K2] = sum_{n2} W_N^{n2 K1} * ( sum_{n1} x[1024 n1 + n2] W_1024^{n1 K1} ) * W_1024^{n2 K2}
// pass 1: tile = 16 adjacent columns n2; 1024-point FFT over n1 per column; multiply by W_N^{n2 K1};
//         store Y[K1][n2] into the scratch ring (same row-major shape).
// pass 2: tile = 16 adjacent rows K1; 1024-point FFT over n2 per row; store X[K1 + 1024 K2]
//         (16 adjacent K1 = one 128-byte segment per K2).
// Each 1024-point FFT = radix-32 (registers) -> twiddle W_1024^{n' k1} -> LDS exchange -> radix-32.
// 512 threads, 32 points per thread, 64 data VGPRs, one 64-KiB exchange buffer used twice
// (real parts, then imaginary parts) so that two workgroups fit in a CU's 160 KiB.
// ---------------------------------------------------------------------------
constexpr int XCH_BYTES = 65536;
constexpr int TWI_BYTES = 8192;
constexpr int TWO_BYTES = 8192;

template <int DIR>
__device__ __forceinline__ void stage1_fft_twiddle(v2f (&x)[32], const v2f *twi, uint32_t q)
{
    fft_reg<32, DIR>(x);
    // x[brev(k1)] = Z[k1]; multiply by W_1024^{q*k1}; table layout [k1][q]
    static_for<1, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        constexpr int r = brev<32>(k1);
        x[r] = cmul_tw<DIR>(x[r], twi[k1 * 32 + q]);
    });
}

template <int DIR>
__global__ __launch_bounds__(512, 4) void k_p1_1m(const v2f *__restrict__ src, v2f *__restrict__ ring,
                                                  const v2f *__restrict__ tw_inner,
                                                  const v2f *__restrict__ tw_outer, uint32_t ring_slots,
                                                  uint64_t t_first)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + XCH_BYTES + TWI_BYTES);

    const uint32_t tid = threadIdx.x;
    const uint32_t c = tid & 15;   // column inside the tile
    const uint32_t q = tid >> 4;   // n' before the exchange, k1 after it
    const uint32_t tile = blockIdx.x & 63;
    const uint64_t t_local = blockIdx.x >> 6;
    const uint64_t t = t_first + t_local;

    // uniform (scalar) base + 32-bit per-thread offset: keeps addresses out of VGPR pairs
    const v2f *in = src + t * (1ull << 20) + 16 * tile;
    const uint32_t off = q * 1024 + c;
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = (in + j * 32768)[off];
    });

    // tables -> LDS (16 B per thread each)
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    reinterpret_cast<v4f *>(two)[tid] = reinterpret_cast<const v4f *>(tw_outer + (size_t)tile * 1024)[tid];
    __syncthreads();

    stage1_fft_twiddle<DIR>(x, twi, q);

    // exchange: word address c + 16*(k1*32 + (n' ^ (k1&1))) -- conflict-free on both sides
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + 16 * (k1 * 32 + (q ^ (k1 & 1)))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].x = xch[c + 16 * (q * 32 + (np ^ (q & 1)))];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + 16 * (k1 * 32 + (q ^ (k1 & 1)))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].y = xch[c + 16 * (q * 32 + (np ^ (q & 1)))];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = FFT1024 output K1 = q + 32*k2

    // four-step twiddle W_N^{n2*K1} = A[q][c] * B[k2][c]
    const v2f A = two[q * 16 + c];
    v2f *out = ring + (t % ring_slots) * (1ull << 20) + 16 * tile;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        const v2f w = cmul(A, two[512 + k2 * 16 + c]);
        (out + k2 * 32768)[off] = cmul_tw<DIR>(x[brev<32>(k2)], w);
    });
}

template <int DIR>
__global__ __launch_bounds__(512, 4) void k_p2_1m(const v2f *__restrict__ ring, v2f *__restrict__ dst,
                                                  const v2f *__restrict__ tw_inner, uint32_t ring_slots,
                                                  uint64_t t_first, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);

    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x & 63;
    const uint64_t t = t_first + (blockIdx.x >> 6);

    // before the exchange: lane = n' (32 consecutive samples of one row), r = row in the tile
    const uint32_t np = tid & 31;
    const uint32_t r = tid >> 5;
    const v2f *in = ring + (t % ring_slots) * (1ull << 20) + (uint64_t)(16 * tile) * 1024;
    const uint32_t off_in = r * 1024 + np;
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = (in + 32 * j)[off_in];
    });

    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    __syncthreads();

    stage1_fft_twiddle<DIR>(x, twi, np);

    // after the exchange: lane = r' (16 adjacent K1 = one 128-B output segment), k1' = tid >> 4
    const uint32_t r2 = tid & 15;
    const uint32_t k1p = tid >> 4;
    // word address (r*32 + k1)*32 + (n' ^ ((r + 16*(k1&1)) & 31))
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ ((r + 16 * (k1 & 1)) & 31))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    const uint32_t rd_base = (r2 * 32 + k1p) * 32;
    const uint32_t rd_xor = (r2 + 16 * (k1p & 1)) & 31;
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].x = xch[rd_base + (n ^ rd_xor)];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ ((r + 16 * (k1 & 1)) & 31))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].y = xch[rd_base + (n ^ rd_xor)];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = row FFT output K2 = k1p + 32*k2

    v2f *out = dst + t * (1ull << 20) + 16 * tile;
    const uint32_t off_out = k1p * 1024 + r2;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        (out + k2 * 32768)[off_out] = x[brev<32>(k2)] * scale;
    });
}

hipError_t setup_1m_kernels()
{
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p1_1m<FWD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            XCH_BYTES + TWI_BYTES + TWO_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p1_1m<INV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            XCH_BYTES + TWI_BYTES + TWO_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p2_1m<FWD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            XCH_BYTES + TWI_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p2_1m<INV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            XCH_BYTES + TWI_BYTES);
    return e;
}

hipError_t launch_p1_1m(int dir, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t ring_slots, uint64_t t_first, uint32_t n_transforms, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    const dim3 grid(n_transforms * 64), block(512);
    const size_t lds = XCH_BYTES + TWI_BYTES + TWO_BYTES;
    if (dir == FWD)
        hipLaunchKernelGGL(k_p1_1m<FWD>, grid, block, lds, st, src, ring, tw_inner, tw_outer, ring_slots, t_first);
    else
        hipLaunchKernelGGL(k_p1_1m<INV>, grid, block, lds, st, src, ring, tw_inner, tw_outer, ring_slots, t_first);
    return hipGetLastError();
}

hipError_t launch_p2_1m(int dir, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t ring_slots,
                        uint64_t t_first, uint32_t n_transforms, float scale, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    const dim3 grid(n_transforms * 64), block(512);
    const size_t lds = XCH_BYTES + TWI_BYTES;
    if (dir == FWD)
        hipLaunchKernelGGL(k_p2_1m<FWD>, grid, block, lds, st, ring, dst, tw_inner, ring_slots, t_first, scale);
    else
        hipLaunchKernelGGL(k_p2_1m<INV>, grid, block, lds, st, ring, dst, tw_inner, ring_slots, t_first, scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// elementwise: normalize (normalize.wgsl:9-12), synthetic fill, calibration copy
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale(const v4f *a, v4f *b, uint64_t n_vec,
                                               float scale)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) b[i] = a[i] * scale;
}

__global__ __launch_bounds__(256) void k_scale_tail(const v2f *a, v2f *b, uint64_t first,
                                                    uint64_t n, float scale)
{
    const uint64_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] * scale;
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
    const uint64_t n_vec = n_samples / 2;
    if (n_vec)
        hipLaunchKernelGGL(k_scale, dim3(stream_grid(n_vec)), dim3(256), 0, st, reinterpret_cast<const v4f *>(a),
                           reinterpret_cast<v4f *>(b), n_vec, scale);
    if (n_samples & 1)
        hipLaunchKernelGGL(k_scale_tail, dim3(1), dim3(256), 0, st, a, b, n_vec * 2, n_samples, scale);
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

__global__ __launch_bounds__(256) void k_copy(const v4f *__restrict__ a, v4f *__restrict__ b, uint64_t n_vec)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) b[i] = a[i];
}

hipError_t launch_copy(const void *src, void *dst, uint64_t bytes, hipStream_t st)
{
    const uint64_t n_vec = bytes / 16;
    if (n_vec == 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy, dim3(stream_grid(n_vec)), dim3(256), 0, st, reinterpret_cast<const v4f *>(src),
                       reinterpret_cast<v4f *>(dst), n_vec);
    return hipGetLastError();
}

}  // namespace fwa
