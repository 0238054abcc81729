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

// Buffer (SRD) addressing: one 32-bit per-lane byte offset + a scalar offset per access, so the 32 loads
// and 32 stores of a tile need no per-access VALU address math (cdna_hip_programming.md T8/T20).  The
// descriptor covers exactly one 8-MiB transform; out-of-range lanes would read 0 / drop the store.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr uint32_t TRANSFORM_BYTES = 8u << 20;
// cache-policy bits of the aux operand (gfx940+): sc0 = 1, nt = 2, sc1 = 16
constexpr int AUX_DEFAULT = 0, AUX_NT = 2, AUX_SC1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const v2f *transform_base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(transform_base), 0, TRANSFORM_BYTES, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ v2f buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_store(v2f v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, AUX);
}


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
// n = 2, 4, 8: each thread owns 16 consecutive samples (16/n whole transforms), one radix-n butterfly
// network per transform in registers, 16-byte loads and stores.
// ---------------------------------------------------------------------------
template <int N, int DIR>
__global__ __launch_bounds__(256) void k_tiny(const v2f *__restrict__ src, v2f *__restrict__ dst, uint64_t n_samples,
                                              float scale)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 16;
    for (uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; base < n_samples; base += stride) {
        v2f x[16];
        if (base + 16 <= n_samples) {
            static_for<0, 8>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                const v4f v = *reinterpret_cast<const v4f *>(src + base + 2 * i);
                x[2 * i] = v2f{v.x, v.y}; x[2 * i + 1] = v2f{v.z, v.w};
            });
        } else {
            static_for<0, 16>([&](auto i_) { constexpr int i = decltype(i_)::value; x[i] = (base + i < n_samples) ? src[base + i] : v2f{0.f, 0.f}; });
        }
        v2f y[16];
        static_for<0, 16 / N>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            v2f t[N];
            static_for<0, N>([&](auto i_) { constexpr int i = decltype(i_)::value; t[i] = x[g * N + i]; });
            fft_reg<N, DIR>(t);
            static_for<0, N>([&](auto k_) { constexpr int k = decltype(k_)::value; y[g * N + k] = t[brev<N>(k)] * scale; });
        });
        if (base + 16 <= n_samples) {
            static_for<0, 8>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                *reinterpret_cast<v4f *>(dst + base + 2 * i) = v4f{y[2 * i].x, y[2 * i].y, y[2 * i + 1].x, y[2 * i + 1].y};
            });
        } else {
            static_for<0, 16>([&](auto i_) { constexpr int i = decltype(i_)::value; if (base + i < n_samples) dst[base + i] = y[i]; });
        }
    }
}

hipError_t launch_tiny(int dir, const v2f *src, v2f *dst, uint32_t n, uint64_t batch, float scale, hipStream_t st)
{
    const uint64_t n_samples = batch * n;
    if (n_samples == 0) return hipSuccess;
    uint64_t blocks = (n_samples / 16 + 255) / 256 + 1;
    if (blocks > 16384) blocks = 16384;
    const dim3 g((uint32_t)blocks), b(256);
#define FWA_TINY(NN)                                                                                        \
    if (dir == FWD) hipLaunchKernelGGL((k_tiny<NN, FWD>), g, b, 0, st, src, dst, n_samples, scale);         \
    else hipLaunchKernelGGL((k_tiny<NN, INV>), g, b, 0, st, src, dst, n_samples, scale)
    switch (n) {
        case 2: FWA_TINY(2); break;
        case 4: FWA_TINY(4); break;
        case 8: FWA_TINY(8); break;
        default: return hipErrorInvalidValue;
    }
#undef FWA_TINY
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// small transforms, 16 <= n <= 4096: register radix-16 Stockham.  Each thread owns 16 points; a transform
// uses n/16 threads; stages are radix 16, 16, ... and a last stage of radix n / 16^k (2, 4 or 8 -- the
// thread then does 16/R butterflies).  Stage recurrence = the reference's (fft.wgsl:27-62) with the pair
// (a, b) generalised to R inputs:  idx = s*J + j;  inputs idx + m*n/R;  outputs s*R*J + j + q*J, scaled
// by W_n^{s*J*q} (table of processor.rs:43-49).  The first stage reads global memory directly (coalesced
// over idx), the last one writes it directly (coalesced over idx); in between one LDS buffer, padded by
// one element per 16, carries the exchange (conflict-free b64 writes at every stage).
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ v2f tw_lookup(const v2f *__restrict__ tw, uint32_t e)  // W_N^e, 0 <= e < N
{
    const v2f w = tw[e & (N / 2 - 1)];
    return (e & (N / 2)) ? -w : w;
}

template <int R, int N, int DIR, class Get, class Put>
__device__ __forceinline__ void stage_bfly(Get get, Put put, const v2f *__restrict__ tw, uint32_t idx, uint32_t J)
{
    // one radix-R butterfly of the Stockham stage with sub-block size J
    v2f x[R];
    static_for<0, R>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = get(idx + m * (N / R)); });
    fft_reg<R, DIR>(x);
    const uint32_t j = idx & (J - 1);
    const uint32_t sJ = idx - j;  // s*J
    static_for<0, R>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        v2f v = x[brev<R>(q)];
        if constexpr (q != 0) {
            if (J * R < N) v = cmul_tw<DIR>(v, tw_lookup<N>(tw, sJ * q));  // last stage: s = 0, no twiddle
        }
        put(sJ * R + j + q * J, v);
    });
}

// In-wave exchange for n = 32, 64, 128 (one radix-16 stage + one radix-R stage, R = n/16 lanes per
// transform): the element in (lane m, register b*R + r) moves to (lane r, register b*R + m) -- an R x R
// transpose per register group, done as log2(R) butterfly steps of `__shfl_xor` + select.  No LDS memory, no
// barrier.  Measured 3-10 % SLOWER than the padded-LDS exchange at these sizes (ds_bpermute issue cost), so the
// plan uses it only when asked (small_reg = 2); DPP quad-permute moves miscompiled under hipcc 7.2 (one of two
// back-to-back moves of a float2 dropped) and are not used.
template <int R>
__device__ __forceinline__ void wave_transpose(v2f (&x)[16], uint32_t lane_in_group)
{
    static_for<0, ilog2c(R)>([&](auto s_) {
        constexpr int sft = decltype(s_)::value;
        const bool hi = (lane_in_group >> sft) & 1;
        static_for<0, 16>([&](auto q_) {
            constexpr int q0 = decltype(q_)::value;
            if constexpr (((q0 % R) >> sft & 1) == 0) {
                constexpr int q1 = q0 | (1 << sft);
                const v2f send = hi ? x[q0] : x[q1];
                v2f recv;
                recv.x = __shfl_xor(send.x, 1 << sft);
                recv.y = __shfl_xor(send.y, 1 << sft);
                if (hi) x[q0] = recv; else x[q1] = recv;
            }
        });
    });
}

template <int LGN, int DIR, bool SHFL = false>
__global__ __launch_bounds__((LGN <= 12 ? 256 : (1 << (LGN - 4)))) void k_small16(const v2f *__restrict__ src,
                                                                                  v2f *__restrict__ dst,
                                                                                  const v2f *__restrict__ tw,
                                                                                  uint64_t batch, float scale)
{
    constexpr int N = 1 << LGN;
    constexpr int TPX = N / 16;                       // threads per transform
    constexpr int WG = LGN <= 12 ? 256 : TPX;         // 8192 / 16384 points: one transform per 512 / 1024 threads
    constexpr int XPW = WG / TPX;                     // transforms per workgroup
    constexpr int NS16 = LGN / 4;        // radix-16 stages
    constexpr int RL = 1 << (LGN % 4);   // last radix (1 = none)
    constexpr int PADN = N + N / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *lds = reinterpret_cast<v2f *>(smem) + (threadIdx.x / TPX) * PADN;
    const uint32_t t = threadIdx.x % TPX;
    const uint64_t xf = (uint64_t)blockIdx.x * XPW + threadIdx.x / TPX;
    const bool live = xf < batch;
    const v2f *g_in = src + xf * N;
    v2f *g_out = dst + xf * N;
    auto pad = [](uint32_t p) { return p + (p >> 4); };

    if constexpr (SHFL && NS16 == 1 && RL > 1) {
        // n = 32, 64, 128: radix-16 from global, wavefront shuffle exchange, radix-RL to global
        v2f x[16];
        static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = live ? g_in[t + m * TPX] : v2f{0.f, 0.f}; });
        fft_reg<16, DIR>(x);
        v2f y[16];
        static_for<0, 16>([&](auto q_) {  // stage-0 output q of thread t sits at t*16 + q; twiddle W_n^{t*q}
            constexpr int q = decltype(q_)::value;
            y[q] = x[brev<16>(q)];
            if constexpr (q != 0) y[q] = cmul_tw<DIR>(y[q], tw_lookup<N>(tw, t * q));
        });
        wave_transpose<RL>(y, t);  // y[b*RL + m] = input m of butterfly idx = t + b*RL
        static_for<0, 16 / RL>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f z[RL];
            static_for<0, RL>([&](auto m_) { constexpr int m = decltype(m_)::value; z[m] = y[b * RL + m]; });
            fft_reg<RL, DIR>(z);
            static_for<0, RL>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if (live) g_out[t + b * RL + q * 16] = z[brev<RL>(q)] * scale;  // idx + q*J, J = 16
            });
        });
        return;
    }
    // stage 0: global -> (LDS | global)
    {
        constexpr bool only = (NS16 == 1 && RL == 1);
        if (live || !only)
            stage_bfly<16, N, DIR>([&](uint32_t i) { return live ? g_in[i] : v2f{0.f, 0.f}; },
                                   [&](uint32_t o, v2f v) {
                                       if constexpr (only) { if (live) g_out[o] = v * scale; }
                                       else lds[pad(o)] = v;
                                   },
                                   tw, t, 1u);
        if constexpr (only) return;
    }
    uint32_t J = 16;
    // middle radix-16 stages
    static_for<1, NS16>([&](auto s_) {
        constexpr int st = decltype(s_)::value;
        constexpr bool last = (st == NS16 - 1) && RL == 1;
        __syncthreads();
        v2f x[16];
        static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(t + m * (N / 16))]; });
        if constexpr (!last) __syncthreads();
        fft_reg<16, DIR>(x);
        const uint32_t j = t & (J - 1), sJ = t - j;
        static_for<0, 16>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            v2f v = x[brev<16>(q)];
            if constexpr (q != 0 && !last) v = cmul_tw<DIR>(v, tw_lookup<N>(tw, sJ * q));
            const uint32_t o = sJ * 16 + j + q * J;
            if constexpr (last) { if (live) g_out[o] = v * scale; }
            else lds[pad(o)] = v;
        });
        J *= 16;
    });
    // last stage of radix RL < 16: 16/RL butterflies per thread, outputs straight to global
    if constexpr (RL > 1) {
        __syncthreads();
        static_for<0, 16 / RL>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            stage_bfly<RL, N, DIR>([&](uint32_t i) { return lds[pad(i)]; },
                                   [&](uint32_t o, v2f v) { if (live) g_out[o] = v * scale; }, tw, t + b * TPX, J);
        });
    }
}

template <int DIR>
static hipError_t launch_small16_dir(const v2f *src, v2f *dst, const v2f *tw, uint32_t lg_n, uint64_t batch, float scale,
                                     bool shfl, hipStream_t st)
{
    const uint32_t n = 1u << lg_n;
    const uint32_t wg = lg_n <= 12 ? 256 : n / 16;
    const uint32_t xpw = wg / (n / 16);
    const uint64_t blocks = (batch + xpw - 1) / xpw;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (lg_n == 4 || (shfl && lg_n <= 7)) ? 0 : (size_t)xpw * (n + n / 16) * sizeof(v2f);
    const dim3 g((uint32_t)blocks), b(wg);
    switch (lg_n) {
        case 4: hipLaunchKernelGGL((k_small16<4, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 5:
            if (shfl) hipLaunchKernelGGL((k_small16<5, DIR, true>), g, b, lds, st, src, dst, tw, batch, scale);
            else hipLaunchKernelGGL((k_small16<5, DIR>), g, b, lds, st, src, dst, tw, batch, scale);
            break;
        case 6:
            if (shfl) hipLaunchKernelGGL((k_small16<6, DIR, true>), g, b, lds, st, src, dst, tw, batch, scale);
            else hipLaunchKernelGGL((k_small16<6, DIR>), g, b, lds, st, src, dst, tw, batch, scale);
            break;
        case 7:
            if (shfl) hipLaunchKernelGGL((k_small16<7, DIR, true>), g, b, lds, st, src, dst, tw, batch, scale);
            else hipLaunchKernelGGL((k_small16<7, DIR>), g, b, lds, st, src, dst, tw, batch, scale);
            break;
        case 8: hipLaunchKernelGGL((k_small16<8, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 9: hipLaunchKernelGGL((k_small16<9, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 10: hipLaunchKernelGGL((k_small16<10, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 11: hipLaunchKernelGGL((k_small16<11, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 12: hipLaunchKernelGGL((k_small16<12, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 13: hipLaunchKernelGGL((k_small16<13, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        case 14: hipLaunchKernelGGL((k_small16<14, DIR>), g, b, lds, st, src, dst, tw, batch, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_small16(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                          bool wave_shuffle, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    uint32_t lg_n = 0;
    while ((1u << lg_n) < n) ++lg_n;
    return dir == FWD ? launch_small16_dir<FWD>(src, dst, tw, lg_n, batch, scale, wave_shuffle, st)
                      : launch_small16_dir<INV>(src, dst, tw, lg_n, batch, scale, wave_shuffle, st);
}

// ---------------------------------------------------------------------------
// k_tile16: 16 FFTs of length L (64 <= L <= 1024) per workgroup along ONE axis of a multi-dimensional view
// of the transform -- the building block of the 2- and 3-pass paths for n = 2^15..2^19 and 2^21..2^30
// (n = N1*N2[*N3]).  Same register radix-16 Stockham stages as k_small16; what differs is addressing:
//   COLS  (strided axis): element i of FFT c at in + i*pitch + c; 16 adjacent c = one 128-B segment, so
//         loads and stores are coalesced over c.  Output element o is multiplied by the four-step twiddle
//         W_T^{(col0 + c)*o} = hi[e>>10]*lo[e&1023] and stored at out + o*pitch + c (in place allowed).
//   ROWS_T (last axis): FFT c is a contiguous row at in + c*row_pitch; loads are coalesced along the row,
//         the exchange re-maps threads, and output element o of row c goes to out + o*out_stride + c
//         (16 adjacent rows = one 128-B segment): the transposed store that restores natural order.
// ---------------------------------------------------------------------------
template <int LGL, int DIR, int MODE, bool BUF>
__global__ __launch_bounds__((1 << LGL)) void k_tile16(TileArgs a)
{
    constexpr int L = 1 << LGL;
    constexpr int TPX = L / 16;
    constexpr int NS16 = LGL / 4;
    constexpr int RL = 1 << (LGL % 4);
    constexpr int PADN = L + L / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *lds_all = reinterpret_cast<v2f *>(smem);
    auto pad = [](uint32_t p) { return p + (p >> 4); };

    const uint32_t tile = blockIdx.x % a.tile_count;
    const uint32_t rest = blockIdx.x / a.tile_count;
    const uint32_t d1 = rest % a.d1_count;
    const uint64_t b = rest / a.d1_count;
    const v2f *in = a.in + b * a.in_sb + d1 * a.in_s1 + tile * a.in_st;
    v2f *out = a.out + b * a.out_sb + d1 * a.out_s1 + tile * a.out_st;

    // mapping B (FFT index fastest): coalesces every access whose 16 FFTs are adjacent in memory
    const uint32_t cB = threadIdx.x & 15, tB = threadIdx.x >> 4;
    // mapping A (position fastest): coalesces along a contiguous row
    const uint32_t cA = threadIdx.x / TPX, tA = threadIdx.x % TPX;
    const uint32_t c0 = (MODE == TILE_COLS) ? cB : cA, t0 = (MODE == TILE_COLS) ? tB : tA;

    // Addressing.  BUF (every byte offset of the tile < 2^32, checked by the launcher): buffer loads/stores
    // with one 32-bit per-lane offset and a scalar offset per access -- no 64-bit multiply per element
    // (cdna_hip_programming.md T8); otherwise plain 64-bit pointers (only the largest transforms).
    const uint32_t pitch32 = (uint32_t)a.pitch, ostride32 = (uint32_t)a.out_stride;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFFu, 0x00020000);
    const uint32_t vin = (MODE == TILE_COLS) ? (t0 * pitch32 + c0) * 8 : (c0 * pitch32 + t0) * 8;
    const uint32_t sin_step = (MODE == TILE_COLS) ? (uint32_t)(L / 16) * pitch32 * 8 : (uint32_t)(L / 16) * 8;

    // stage 0: global -> LDS (L >= 64, so there is always a later stage); inputs i = t0 + m*L/16
    {
        v2f *lds = lds_all + c0 * PADN;
        v2f x[16];
        static_for<0, 16>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if constexpr (BUF) x[m] = buf_load<AUX_DEFAULT>(rin, vin, m * sin_step);
            else x[m] = (MODE == TILE_COLS) ? in[(uint64_t)(t0 + m * (L / 16)) * a.pitch + c0]
                                            : in[(uint64_t)c0 * a.pitch + t0 + m * (L / 16)];
        });
        fft_reg<16, DIR>(x);
        static_for<0, 16>([&](auto q_) {  // J = 1: s = t0, output position t0*16 + q, twiddle W_L^{t0*q}
            constexpr int q = decltype(q_)::value;
            v2f v = x[brev<16>(q)];
            if constexpr (q != 0) v = cmul_tw<DIR>(v, tw_lookup<L>(a.tw, t0 * q));
            lds[t0 * 17 + q] = v;  // pad(t0*16 + q) = t0*16 + q + t0
        });
    }
    v2f *lds = lds_all + cB * PADN;
    const uint32_t t = tB;
    // Four-step twiddle (COLS).  Every output of this thread has index o = t + m*TPX, m = 0..15, so
    // W_T^{col*o} = [W^{col*t} * (W^{col*TPX})^(m&3)] * W^{col*TPX*4*(m>>2)}: four table look-ups
    // (hi[e>>10]*lo[e&1023] each) and short products instead of one look-up pair per output.
    v2f pa[4], pb[4];
    if constexpr (MODE == TILE_COLS) {
        const uint32_t col = (a.flags & 1) ? 0u : tile * 16 + cB;  // flags&1: timing-only, all twiddles = 1
        auto look = [&](uint32_t e) { return cmul(a.tw_hi[e >> 10], a.tw_lo[e & 1023]); };
        const v2f wt = look(col * t), p1 = look(col * TPX);
        pa[0] = v2f{1.f, 0.f}; pa[1] = look(col * (4 * TPX)); pa[2] = look(col * (8 * TPX)); pa[3] = cmul(pa[2], pa[1]);
        pb[0] = wt; pb[1] = cmul(wt, p1);
        const v2f p2 = cmul(p1, p1);
        pb[2] = cmul(wt, p2); pb[3] = cmul(pb[2], p1);
    }
    // output m of this thread: index o = t + m*TPX (m is a compile-time constant at every call site)
    auto emit = [&](auto m_, v2f v) {
        constexpr uint32_t m = decltype(m_)::value;
        const uint32_t o = t + m * TPX;
        if constexpr (MODE == TILE_COLS) {
            v = cmul_tw<DIR>(v, cmul(pa[m >> 2], pb[m & 3])) * a.scale;
            if constexpr (BUF) buf_store<AUX_DEFAULT>(v, rout, (t * pitch32 + cB) * 8, m * (uint32_t)TPX * pitch32 * 8);
            else out[(uint64_t)o * a.pitch + cB] = v;
        } else {
            v = v * a.scale;
            if constexpr (BUF) buf_store<AUX_DEFAULT>(v, rout, (t * ostride32 + cB) * 8, m * (uint32_t)TPX * ostride32 * 8);
            else out[(uint64_t)o * a.out_stride + cB] = v;
        }
    };
    uint32_t J = 16;
    static_for<1, NS16>([&](auto s_) {
        constexpr int st = decltype(s_)::value;
        constexpr bool last = (st == NS16 - 1) && RL == 1;
        __syncthreads();
        v2f x[16];
        static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(t + m * (L / 16))]; });
        if constexpr (!last) __syncthreads();
        fft_reg<16, DIR>(x);
        const uint32_t j = t & (J - 1), sJ = t - j;
        static_for<0, 16>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            v2f v = x[brev<16>(q)];
            if constexpr (last) {
                emit(q_, v);  // last stage: J = TPX, s = 0, o = t + q*TPX
            } else {
                if constexpr (q != 0) v = cmul_tw<DIR>(v, tw_lookup<L>(a.tw, sJ * q));
                lds[pad(sJ * 16 + j + q * J)] = v;
            }
        });
        J *= 16;
    });
    if constexpr (RL > 1) {
        // last stage of radix RL < 16: butterflies idx = t + b*TPX, inputs idx + m*L/RL, output q at
        // idx + q*L/RL = t + (b + q*16/RL)*TPX; s = 0, so no stage twiddle
        __syncthreads();
        static_for<0, 16 / RL>([&](auto b_) {
            constexpr int bb = decltype(b_)::value;
            v2f x[RL];
            static_for<0, RL>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                x[m] = lds[pad(t + bb * TPX + m * (L / RL))];
            });
            fft_reg<RL, DIR>(x);
            static_for<0, RL>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                emit(std::integral_constant<int, bb + q * (16 / RL)>{}, x[brev<RL>(q)]);
            });
        });
    }
}

template <int DIR, int MODE, bool BUF>
static const void *tile16_kernel_b(uint32_t lg_l)
{
    switch (lg_l) {
        case 6: return reinterpret_cast<const void *>(&k_tile16<6, DIR, MODE, BUF>);
        case 7: return reinterpret_cast<const void *>(&k_tile16<7, DIR, MODE, BUF>);
        case 8: return reinterpret_cast<const void *>(&k_tile16<8, DIR, MODE, BUF>);
        case 9: return reinterpret_cast<const void *>(&k_tile16<9, DIR, MODE, BUF>);
        case 10: return reinterpret_cast<const void *>(&k_tile16<10, DIR, MODE, BUF>);
        default: return nullptr;
    }
}
template <int DIR, int MODE>
static const void *tile16_kernel(uint32_t lg_l, bool buf = true)
{
    return buf ? tile16_kernel_b<DIR, MODE, true>(lg_l) : tile16_kernel_b<DIR, MODE, false>(lg_l);
}
static size_t tile16_lds(uint32_t lg_l) { return (size_t)16 * ((1u << lg_l) + (1u << lg_l) / 16) * sizeof(v2f); }

// called at plan creation: raises the dynamic-LDS limit of the kernels a plan will launch (L >= 512)
hipError_t prepare_tile16(uint32_t lg_l)
{
    const size_t lds = tile16_lds(lg_l);
    if (lds <= 65536) return hipSuccess;
    const void *ks[8] = {tile16_kernel<FWD, TILE_COLS>(lg_l, true),  tile16_kernel<FWD, TILE_ROWS_T>(lg_l, true),
                         tile16_kernel<INV, TILE_COLS>(lg_l, true),  tile16_kernel<INV, TILE_ROWS_T>(lg_l, true),
                         tile16_kernel<FWD, TILE_COLS>(lg_l, false), tile16_kernel<FWD, TILE_ROWS_T>(lg_l, false),
                         tile16_kernel<INV, TILE_COLS>(lg_l, false), tile16_kernel<INV, TILE_ROWS_T>(lg_l, false)};
    for (const void *k : ks) {
        if (!k) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int DIR, int MODE>
static hipError_t launch_tile16_mode(uint32_t lg_l, const TileArgs &a, uint64_t blocks, hipStream_t st)
{
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one tile?  COLS: L rows of `pitch`; ROWS_T: 16 rows of `pitch` in, L outputs of out_stride
    const uint64_t L = 1ull << lg_l;
    const uint64_t span = (MODE == TILE_COLS) ? L * a.pitch * 8 + 128
                                              : ((16 * a.pitch + L) * 8 > (L * a.out_stride + 16) * 8 ? (16 * a.pitch + L) * 8
                                                                                                    : (L * a.out_stride + 16) * 8);
    const void *k = tile16_kernel<DIR, MODE>(lg_l, span < (1ull << 32));
    if (!k) return hipErrorInvalidValue;
    TileArgs copy = a;
    void *args[] = {&copy};
    return hipLaunchKernel(k, dim3((uint32_t)blocks), dim3(1u << lg_l), args, tile16_lds(lg_l), st);
}

hipError_t launch_tile16(int dir, int mode, uint32_t lg_l, const TileArgs &a, uint64_t batch, hipStream_t st)
{
    const uint64_t blocks = batch * a.d1_count * a.tile_count;
    if (dir == FWD)
        return mode == TILE_COLS ? launch_tile16_mode<FWD, TILE_COLS>(lg_l, a, blocks, st)
                                 : launch_tile16_mode<FWD, TILE_ROWS_T>(lg_l, a, blocks, st);
    return mode == TILE_COLS ? launch_tile16_mode<INV, TILE_COLS>(lg_l, a, blocks, st)
                             : launch_tile16_mode<INV, TILE_ROWS_T>(lg_l, a, blocks, st);
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

// One pass-1 tile: column FFTs.  `in`/`out` are the (wave-uniform) bases of a 1024x1024 row-major
// transform; they may be the same transform (in-place: every load of the workgroup completes before the
// first exchange barrier, every store is issued after the last one).
// Requires: twi loaded; `two` free to overwrite (all threads past their previous use).
// OUT_LIN: the output is a scratch slab in tile-contiguous layout -- tile s owns bytes [s*128 KiB, +128 KiB)
// as [K1 (1024)][column (16)]: every pass-1 store instruction of the workgroup covers 4 KiB contiguous,
// and a pass-2 tile finds its 16 rows of a source tile as ONE 2-KiB chunk (measured +5 % over the strided
// matrix layout, tools/tile_probe.hip).
template <int DIR, int AUX_IN, int AUX_OUT, bool OUT_LIN = false>
__device__ __forceinline__ void p1_tile(const v2f *in, v2f *out, uint32_t tile, const v2f *tw_outer_tile,
                                        float *xch, const v2f *twi, v2f *two, uint32_t tid, bool skeleton = false)
{
    const uint32_t c = tid & 15;  // column inside the tile
    const uint32_t q = tid >> 4;  // n' before the exchange, k1 after it
    const uint32_t voff = (q * 1024 + c) * 8;
    const uint32_t soff = tile * 128;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_IN>(rin, voff, soff + j * 262144);
    });
    reinterpret_cast<v4f *>(two)[tid] = reinterpret_cast<const v4f *>(tw_outer_tile)[tid];
    __syncthreads();
    const uint32_t voff_o = OUT_LIN ? (q * 16 + c) * 8 : voff;
    const uint32_t soff_o = OUT_LIN ? tile * 131072 : soff;
    constexpr uint32_t kstep = OUT_LIN ? 4096 : 262144;  // bytes between K1 = q + 32*k2 and q + 32*(k2+1)
    if (skeleton) {  // measurement only: same loads and stores, no arithmetic, no LDS exchange
        static_for<0, 32>([&](auto k_) {
            constexpr int k2 = decltype(k_)::value;
            buf_store<AUX_OUT>(x[k2], rout, voff_o, soff_o + k2 * kstep);
        });
        return;
    }

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
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        const v2f w = cmul(A, two[512 + k2 * 16 + c]);
        buf_store<AUX_OUT>(cmul_tw<DIR>(x[brev<32>(k2)], w), rout, voff_o, soff_o + k2 * kstep);
    });
}

// One pass-2 tile: row FFTs + transposed store.  `in`/`out` are wave-uniform transform bases; the tile
// reads rows [16*tile, 16*tile+16) and writes columns [16*tile, 16*tile+16).  after_load() runs once every load of the calling thread
// has been issued and before the first barrier; before_store() runs right before the first store.
// The in-place fused kernel uses them for the "all 64 tiles loaded" hand-shake.
template <int DIR, int AUX_IN, int AUX_OUT, bool IN_LIN = false, class AfterLoad, class BeforeStore>
__device__ __forceinline__ void p2_tile(const v2f *in, v2f *out, uint32_t tile, float scale, float *xch,
                                        const v2f *twi, uint32_t tid, AfterLoad after_load,
                                        BeforeStore before_store, bool skeleton = false)
{
    // before the exchange: lane = n' (32 consecutive samples of one row), r = row in the tile
    const uint32_t np = tid & 31;
    const uint32_t r = tid >> 5;
    // IN_LIN (tile-contiguous slab, see p1_tile): sample n2 = 32*j + n' lives in source tile 2*j + (n'>>4),
    // whose rows [16*tile, 16*tile+16) form one 2-KiB chunk [row][16 columns].
    const uint32_t voff_in = IN_LIN ? (np >> 4) * 131072 + r * 128 + (np & 15) * 8 : (r * 1024 + np) * 8;
    const uint32_t soff_in = IN_LIN ? tile * 2048 : tile * 131072;
    constexpr uint32_t jstep = IN_LIN ? 262144 : 256;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_IN>(rin, voff_in, soff_in + j * jstep);
    });
    after_load();
    if (skeleton) {  // measurement only
        before_store();
        const uint32_t vo = ((tid >> 4) * 1024 + (tid & 15)) * 8;
        static_for<0, 32>([&](auto k_) {
            constexpr int k2 = decltype(k_)::value;
            buf_store<AUX_OUT>(x[k2], rout, vo, tile * 128 + k2 * 262144);
        });
        return;
    }

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

    before_store();
    const uint32_t voff_out = (k1p * 1024 + r2) * 8;
    const uint32_t soff_out = tile * 128;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        buf_store<AUX_OUT>(x[brev<32>(k2)] * scale, rout, voff_out, soff_out + k2 * 262144);
    });
}

template <int DIR, int AIN = AUX_DEFAULT, int AOUT = AUX_DEFAULT>
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
    const uint32_t tile = blockIdx.x & 63;
    const uint64_t t = t_first + (blockIdx.x >> 6);
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p1_tile<DIR, AIN, AOUT, true>(src + t * (1ull << 20), ring + (t % ring_slots) * (1ull << 20), tile,
                                           tw_outer + (size_t)tile * 1024, xch, twi, two, tid);
}

template <int DIR, int AIN = AUX_DEFAULT, int AOUT = AUX_DEFAULT>
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
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p2_tile<DIR, AIN, AOUT, true>(ring + (t % ring_slots) * (1ull << 20), dst + t * (1ull << 20), tile, scale,
                                           xch, twi, tid, [] { __syncthreads(); }, [] {});
}

// Mixed launch: even workgroups run pass-1 tiles of one group of transforms, odd workgroups run pass-2
// tiles of the PREVIOUS group of the same chain (whose pass 1 finished in the previous launch on this
// stream).  No dependency exists inside a launch, so there is nothing to wait for; every CU hosts
// pass-1 (HBM-read heavy) and pass-2 (HBM-write heavy) workgroups side by side and both HBM directions
// stay busy across the whole launch.
template <int DIR, int A1IN, int A1OUT, int A2IN, int A2OUT>
__global__ __launch_bounds__(512, 4) void k_mix_1m(const v2f *__restrict__ p1_src, v2f *__restrict__ p1_ring,
                                                   uint32_t n1, const v2f *__restrict__ p2_ring,
                                                   v2f *__restrict__ p2_dst, uint32_t n2,
                                                   const v2f *__restrict__ tw_inner,
                                                   const v2f *__restrict__ tw_outer, float scale, uint32_t dbg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + XCH_BYTES + TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    // role assignment: blocks are dealt round-robin over the 8 XCDs, so `blockIdx & 1` would put every pass-1
    // tile on four XCDs and every pass-2 tile on the other four; bit 3 instead gives each XCD both roles
    // (dbg & 128 selects the old split, for A/B timing).
    uint32_t role, idx;
    if (dbg & 128) { role = blockIdx.x & 1; idx = blockIdx.x >> 1; }
    else { role = (blockIdx.x >> 3) & 1; idx = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7); }
    const bool skel = (dbg & 32) != 0;  // timing-only: memory skeleton
    const uint32_t tile = idx & 63;
    const uint64_t t = idx >> 6;
    if (role == 0) {
        if (t >= n1) return;
        reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
        p1_tile<DIR, A1IN, A1OUT, true>(p1_src + t * (1ull << 20), p1_ring + t * (1ull << 20), tile,
                                  tw_outer + (size_t)tile * 1024, xch, twi, two, tid, skel);
    } else {
        if (t >= n2) return;
        reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
        p2_tile<DIR, A2IN, A2OUT, true>(p2_ring + t * (1ull << 20), p2_dst + t * (1ull << 20), tile, scale, xch, twi, tid,
                                  [] { __syncthreads(); }, [] {}, skel);
    }
}

// ---------------------------------------------------------------------------
// Fused, in-place, persistent 2^20 pipeline: ONE launch per exec, no scratch.
//
// Workgroups pull tickets from one counter.  Ticket order interleaves pass-1 tiles of transform t
// with pass-2 tiles of transform t-D, so HBM reads (pass 1), cache-resident intermediate traffic and
// HBM writes (pass 2) overlap continuously.  Pass 1 overwrites its column tile in place with Y; pass 2
// reads 16 rows of Y and writes the 16-column tile of X over the same transform.  Because pass 2
// transposes, a pass-2 tile may only store once ALL 64 pass-2 tiles of that transform hold their rows in
// registers: `loaded[t]`.  `done1[t]` counts finished pass-1 tiles.
//
// Progress: tickets are handed out in order; pass-1 tiles wait for nothing; a pass-2 tile waits only for
// tickets of its own transform or lower.  The dequeued tickets always form a prefix, at most one
// transform is partially dequeued, so at most 63 workgroups can be parked in the loaded[] wait: any
// launch with >= 64 resident workgroups makes progress.  Spins are bounded (ctl->error) regardless.
// Visibility between workgroups follows cdna_hip_programming.md Guideline 16 (agent-scope release by the
// producer after every wave drained its stores; relaxed poll + one agent-scope acquire by the consumer).
// ---------------------------------------------------------------------------
struct FusedCtl {
    uint32_t ticket;
    uint32_t error;
    uint32_t pad[30];
    // followed by done1[batch], loaded[batch]
};

template <bool RMW = false>
__device__ __forceinline__ bool spin_until_64(uint32_t *p, uint32_t *err)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    // RMW: poll with a returning atomic (served where agent-scope atomics execute, never by a cached copy)
    while ((RMW ? __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 64u) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s: never in a healthy run
            __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

template <int DIR, bool FENCES, int P1_IN, int P1_OUT, int P2_IN, int P2_OUT>
__global__ __launch_bounds__(512, 4) void k_fused_1m(v2f *data, const v2f *__restrict__ tw_inner,
                                                     const v2f *__restrict__ tw_outer, uint32_t *ctl_words,
                                                     uint32_t batch, uint32_t depth, float scale, uint32_t dbg)
{
    // dbg: timing-only ablation switches (results are WRONG when any is set; never set by the product path)
    //   1 skip release fence, 2 skip acquire fence, 4 skip loaded[] wait, 8 skip done1[] wait, 16 skip counters
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + XCH_BYTES + TWI_BYTES);
    uint32_t *ticket = ctl_words;
    uint32_t *err = ctl_words + 1;
    uint32_t *done1 = ctl_words + 32;
    uint32_t *loaded = done1 + batch;

    reinterpret_cast<v4f *>(twi)[threadIdx.x] = reinterpret_cast<const v4f *>(tw_inner)[threadIdx.x];

    const uint32_t total = 128u * batch;
    const uint32_t prologue = 64u * depth;             // pass-1 tiles of transforms 0..depth-1
    const uint32_t steady = 128u * (batch - depth);    // interleaved region

    for (;;) {
        // Opaque per-iteration copy of the thread id: without it LICM hoists ~100 lane-constant LDS/global
        // offsets out of the persistent loop and spills them (cdna_hip_programming.md, attention pitfalls).
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid == 0)
            reinterpret_cast<uint32_t *>(xch)[0] =
                __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // readfirstlane: the ticket is wave-uniform, keep everything derived from it in SGPRs
        const uint32_t k = __builtin_amdgcn_readfirstlane(reinterpret_cast<uint32_t *>(xch)[0]);
        __syncthreads();  // xch is reused by the exchange below
        if (k >= total) break;

        uint32_t pass, t, tile;
        if (k < prologue) {
            pass = 1; t = k >> 6; tile = k & 63;
        } else if (k - prologue < steady) {
            const uint32_t kk = k - prologue;
            const uint32_t s = kk >> 7, r = kk & 127;
            tile = r >> 1;
            if ((r & 1) == 0) { pass = 1; t = s + depth; } else { pass = 2; t = s; }
        } else {
            const uint32_t kk = k - prologue - steady;
            pass = 2; t = (batch - depth) + (kk >> 6); tile = kk & 63;
        }
        v2f *base = data + (uint64_t)t * (1ull << 20);

        if (pass == 1) {
            p1_tile<DIR, P1_IN, P1_OUT>(base, base, tile, tw_outer + (size_t)tile * 1024, xch, twi, two, tid,
                                        (dbg & 32) != 0);
            // publish: every wave drains its stores, then one lane releases at agent scope and counts
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (FENCES && !(dbg & 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (!(dbg & 16)) __hip_atomic_fetch_add(&done1[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (tid == 0) {
                if (!(dbg & 8)) { if (dbg & 64) spin_until_64<true>(&done1[t], err); else spin_until_64<false>(&done1[t], err); }
                if (FENCES && !(dbg & 2)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            p2_tile<DIR, P2_IN, P2_OUT, false>(
                base, base, tile, scale, xch, twi, tid,
                [&] {
                    // this tile's rows are in registers: tell the other 63 tiles of the transform
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0 && !(dbg & 16))
                        __hip_atomic_fetch_add(&loaded[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                },
                [&] {
                    if (tid == 0 && !(dbg & 4)) { if (dbg & 64) spin_until_64<true>(&loaded[t], err); else spin_until_64<false>(&loaded[t], err); }
                    __syncthreads();
                },
                (dbg & 32) != 0);
            // the next iteration's first barrier orders these stores' issue after everything above;
            // nothing in this launch reads X, the kernel boundary publishes it.
        }
    }
}

// Cache-policy variants (measured with tools/fabric_probe2: write-through `sc1` stores for the
// cache-resident intermediate and `nt` on the HBM-facing side lift the mixed-traffic ceiling).
//   policy 0: default loads/stores; fused kernel publishes with agent-scope release/acquire fences
//   policy 1: HBM side nt, intermediate stored sc1 (write-through) and loaded sc1: no fences needed
//             (Guideline 16 form "every store of the handed-off bytes sc1, drained, then counter;
//              every load of them an sc1 load after the poll + workgroup barrier")
//   policy 2: as 1 with sc0|sc1 stores      policy 3: as 1 without nt      policy 4: nt only, fences kept
constexpr int N_POLICIES = 5;

template <int DIR>
static const void *fused_kernel(int policy)
{
    switch (policy) {
        case 1: return reinterpret_cast<const void *>(&k_fused_1m<DIR, false, AUX_NT, AUX_SC1, AUX_SC1, AUX_NT>);
        case 2: return reinterpret_cast<const void *>(&k_fused_1m<DIR, false, AUX_NT, AUX_SC1 | 1, AUX_SC1, AUX_NT>);
        case 3: return reinterpret_cast<const void *>(&k_fused_1m<DIR, false, AUX_DEFAULT, AUX_SC1, AUX_SC1, AUX_DEFAULT>);
        case 4: return reinterpret_cast<const void *>(&k_fused_1m<DIR, true, AUX_NT, AUX_DEFAULT, AUX_DEFAULT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_fused_1m<DIR, true, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT>);
    }
}
template <int DIR>
static const void *p1_kernel(int policy)
{
    switch (policy) {
        case 1: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_SC1>);
        case 2: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_SC1 | 1>);
        case 3: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_DEFAULT, AUX_SC1>);
        case 4: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_DEFAULT>);
        default: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_DEFAULT, AUX_DEFAULT>);
    }
}
template <int DIR>
static const void *p2_kernel(int policy)
{
    switch (policy) {
        case 1: case 2: case 4: return reinterpret_cast<const void *>(&k_p2_1m<DIR, AUX_DEFAULT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_p2_1m<DIR, AUX_DEFAULT, AUX_DEFAULT>);
    }
}

template <int DIR>
static const void *mix_kernel(int policy)
{
    switch (policy) {
        case 1: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_NT, AUX_SC1, AUX_DEFAULT, AUX_NT>);
        case 2: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_NT, AUX_SC1 | 1, AUX_DEFAULT, AUX_NT>);
        case 3: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_DEFAULT, AUX_SC1, AUX_DEFAULT, AUX_DEFAULT>);
        case 4: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_NT, AUX_DEFAULT, AUX_DEFAULT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT>);
    }
}

hipError_t launch_mix_1m(int dir, int policy, const v2f *p1_src, v2f *p1_ring, uint32_t n1, const v2f *p2_ring,
                         v2f *p2_dst, uint32_t n2, const v2f *tw_inner, const v2f *tw_outer, float scale,
                         uint32_t dbg, hipStream_t st)
{
    const uint32_t nmax = n1 > n2 ? n1 : n2;
    if (nmax == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    void *args[] = {&p1_src, &p1_ring, &n1, &p2_ring, &p2_dst, &n2, &tw_inner, &tw_outer, &scale, &dbg};
    const void *k = (dir == FWD) ? mix_kernel<FWD>(policy) : mix_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(nmax * 128), dim3(512), args, XCH_BYTES + TWI_BYTES + TWO_BYTES, st);
}

hipError_t setup_small_kernels()
{
    // 8192 / 16384-point transforms need 68 / 136 KiB of dynamic LDS
    const void *ks[4] = {reinterpret_cast<const void *>(&k_small16<13, FWD>), reinterpret_cast<const void *>(&k_small16<13, INV>),
                         reinterpret_cast<const void *>(&k_small16<14, FWD>), reinterpret_cast<const void *>(&k_small16<14, INV>)};
    for (int i = 0; i < 4; ++i) {
        const int n = i < 2 ? 8192 : 16384;
        hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (n + n / 16) * 8);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t setup_1m_kernels()
{
    const int big = XCH_BYTES + TWI_BYTES + TWO_BYTES, small = XCH_BYTES + TWI_BYTES;
    for (int pol = 0; pol < N_POLICIES; ++pol) {
        const void *ks[8] = {fused_kernel<FWD>(pol), fused_kernel<INV>(pol), p1_kernel<FWD>(pol),
                             p1_kernel<INV>(pol),    mix_kernel<FWD>(pol),   mix_kernel<INV>(pol),
                             p2_kernel<FWD>(pol),    p2_kernel<INV>(pol)};
        for (int i = 0; i < 8; ++i) {
            hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 6 ? big : small);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

size_t fused_ctl_bytes(uint64_t batch) { return sizeof(uint32_t) * (32 + 2 * batch); }

hipError_t launch_fused_1m(int dir, int policy, v2f *data, const v2f *tw_inner, const v2f *tw_outer, uint32_t *ctl,
                           uint32_t batch, uint32_t depth, uint32_t n_workgroups, float scale, uint32_t dbg,
                           hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    if (depth < 1) depth = 1;
    if (depth > batch) depth = batch;
    hipError_t e = hipMemsetAsync(ctl, 0, fused_ctl_bytes(batch), st);
    if (e != hipSuccess) return e;
    const uint32_t max_useful = 128u * batch;
    if (n_workgroups > max_useful) n_workgroups = max_useful;
    void *args[] = {&data, &tw_inner, &tw_outer, &ctl, &batch, &depth, &scale, &dbg};
    const void *k = (dir == FWD) ? fused_kernel<FWD>(policy) : fused_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(n_workgroups), dim3(512), args, XCH_BYTES + TWI_BYTES + TWO_BYTES, st);
}

hipError_t launch_p1_1m(int dir, int policy, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t ring_slots, uint64_t t_first, uint32_t n_transforms, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    void *args[] = {&src, &ring, &tw_inner, &tw_outer, &ring_slots, &t_first};
    const void *k = (dir == FWD) ? p1_kernel<FWD>(policy) : p1_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(n_transforms * 64), dim3(512), args, XCH_BYTES + TWI_BYTES + TWO_BYTES, st);
}

hipError_t launch_p2_1m(int dir, int policy, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t ring_slots,
                        uint64_t t_first, uint32_t n_transforms, float scale, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    void *args[] = {&ring, &dst, &tw_inner, &ring_slots, &t_first, &scale};
    const void *k = (dir == FWD) ? p2_kernel<FWD>(policy) : p2_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(n_transforms * 64), dim3(512), args, XCH_BYTES + TWI_BYTES, st);
}

// ---------------------------------------------------------------------------
// Large / mid-size transforms: n = R1 * R2 * M.  Strided register-radix passes split the transform into
// R1*R2 contiguous sub-transforms of length M (M = 2^20 -> the two-pass pipeline above, M = 4096 ->
// k_lds_small), and one digit-reversal permute restores natural order:
//   X[k1 + R*k'] = DFT_S(sub-array k1)[k'],  sub-array k1 [n2] = W_cur^{n2 k1} * sum_{n1} x[n1*S + n2] W_R^{n1 k1}
// (cur = R*S).  This replaces the reference's log2(n) full passes (fft4.wgsl:36-101) by 2-3 passes plus
// the sub-transform.  Twiddle W_cur^e = hi[e >> 10] * lo[e & 1023] (two f64-derived table entries).
// ---------------------------------------------------------------------------
template <int R, int DIR>
__global__ __launch_bounds__(256) void k_radix_pass(const v2f *__restrict__ in, v2f *__restrict__ out,
                                                    const v2f *__restrict__ tw_lo, const v2f *__restrict__ tw_hi,
                                                    uint32_t lg_s, uint64_t total /* n_sub * S */)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const uint32_t S = 1u << lg_s;
    const uint64_t sub = g >> lg_s;
    const uint32_t n2 = (uint32_t)(g & (S - 1));
    const uint64_t base = sub * ((uint64_t)R << lg_s) + n2;
    v2f x[R];
    static_for<0, R>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = in[base + ((uint64_t)j << lg_s)];
    });
    fft_reg<R, DIR>(x);
    static_for<0, R>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        v2f v = x[brev<R>(k1)];
        if constexpr (k1 != 0) {
            const uint32_t e = n2 * (uint32_t)k1;  // < cur <= 2^30
            const v2f w = cmul(tw_hi[e >> 10], tw_lo[e & 1023]);
            v = cmul_tw<DIR>(v, w);
        }
        out[base + ((uint64_t)k1 << lg_s)] = v;
    });
}

template <int DIR>
static hipError_t launch_radix_pass_dir(int R, const v2f *in, v2f *out, const v2f *lo, const v2f *hi, uint32_t lg_s,
                                        uint64_t total, hipStream_t st)
{
    const uint64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 g((uint32_t)blocks), b(256);
    switch (R) {
        case 2: hipLaunchKernelGGL((k_radix_pass<2, DIR>), g, b, 0, st, in, out, lo, hi, lg_s, total); break;
        case 4: hipLaunchKernelGGL((k_radix_pass<4, DIR>), g, b, 0, st, in, out, lo, hi, lg_s, total); break;
        case 8: hipLaunchKernelGGL((k_radix_pass<8, DIR>), g, b, 0, st, in, out, lo, hi, lg_s, total); break;
        case 16: hipLaunchKernelGGL((k_radix_pass<16, DIR>), g, b, 0, st, in, out, lo, hi, lg_s, total); break;
        case 32: hipLaunchKernelGGL((k_radix_pass<32, DIR>), g, b, 0, st, in, out, lo, hi, lg_s, total); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_radix_pass(int dir, int R, const v2f *in, v2f *out, const v2f *tw_lo, const v2f *tw_hi,
                             uint32_t lg_s, uint64_t n_sub, hipStream_t st)
{
    const uint64_t total = n_sub << lg_s;
    if (total == 0) return hipSuccess;
    return dir == FWD ? launch_radix_pass_dir<FWD>(R, in, out, tw_lo, tw_hi, lg_s, total, st)
                      : launch_radix_pass_dir<INV>(R, in, out, tw_lo, tw_hi, lg_s, total, st);
}

// out[t][k1 + R1*(k2 + R2*k3)] = scale * in[t][(k1*R2 + k2)*M + k3].  An Rt x M -> M x Rt transpose per
// transform (Rt = R1*R2), tiled through LDS: a workgroup takes TK = 4096/Rt consecutive k3, reads Rt rows of
// TK contiguous samples (coalesced) and writes one contiguous 32-KiB block (coalesced); rows are padded by one
// element so the transposed LDS read is conflict-free.
__global__ __launch_bounds__(256) void k_permute(const v2f *__restrict__ in, v2f *__restrict__ out, uint32_t lg_r1,
                                                 uint32_t lg_r2, uint32_t lg_m, float scale)
{
    __shared__ v2f tile[4096 + 128];
    const uint32_t lg_rt = lg_r1 + lg_r2, Rt = 1u << lg_rt;
    const uint32_t lg_tk = 12 - lg_rt, TK = 1u << lg_tk;          // k3 per tile
    const uint32_t tiles_per_x = 1u << (lg_m - lg_tk);
    const uint64_t t = blockIdx.x / tiles_per_x;
    const uint32_t k0 = (blockIdx.x % tiles_per_x) << lg_tk;
    const v2f *src = in + (t << (lg_m + lg_rt));
    v2f *dst = out + (t << (lg_m + lg_rt)) + ((uint64_t)k0 << lg_rt);
    const uint32_t R1m = (1u << lg_r1) - 1;
    for (uint32_t e = threadIdx.x; e < 4096; e += 256) {
        const uint32_t q = e >> lg_tk, k = e & (TK - 1);           // q = k1 + R1*k2 (output digit order)
        const uint32_t row = ((q & R1m) << lg_r2) + (q >> lg_r1);  // k1*R2 + k2 (storage order)
        tile[q * (TK + 1) + k] = src[((uint64_t)row << lg_m) + k0 + k];
    }
    __syncthreads();
    for (uint32_t o = threadIdx.x; o < 4096; o += 256) {
        const uint32_t k = o >> lg_rt, q = o & (Rt - 1);
        dst[o] = tile[q * (TK + 1) + k] * scale;
    }
}

hipError_t launch_permute(const v2f *in, v2f *out, uint32_t lg_r1, uint32_t lg_r2, uint32_t lg_m, uint64_t batch,
                          float scale, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    const uint32_t lg_rt = lg_r1 + lg_r2;
    if (lg_rt < 1 || lg_rt > 10 || lg_m + lg_rt < 12) return hipErrorInvalidValue;  // TK = 4096/Rt must divide M
    const uint64_t blocks = batch << (lg_m + lg_rt - 12);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_permute, dim3((uint32_t)blocks), dim3(256), 0, st, in, out, lg_r1, lg_r2, lg_m, scale);
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
