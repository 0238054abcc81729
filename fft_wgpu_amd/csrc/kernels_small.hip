// kernels_small.hip -- (1/3: one-launch kernels for n <= 16384, radix-2 fallback, elementwise)
// kernels_*.hip -- hand-written gfx950 kernels of the batched 1-D c2c fp32 FFT.
//
// Reference mapping (all under /root/reference/src):
//   k_r2_stage      <- kernel/fft.wgsl:27-62, ifft.wgsl:25-75 (one butterfly per thread, one launch per stage)
//   k_lds_small     <- kernel/fft4.wgsl:13-112 (one dispatch, all stages) staged in LDS as kernel/fft2.wgsl:9-10 intended
//   k_p1_1m/k_p2_1m <- kernel/fft4.wgsl at fft_len = 2^20 (config C2/C3), re-designed: two LDS-tiled
//                      passes of 32x32 register FFTs, intermediate in a small cache-resident ring
//   k_normalize     <- kernel/normalize.wgsl:9-12
// Wavefront = 64, 16-waves-per-CU residency (2 x 512-thread workgroups) for the 2^20 passes.
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

template <int LGN, int DIR, bool SHFL = false, bool SPLIT = (LGN >= 13)>
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
    // Exchange between two stages: every thread deposits its 16 stage outputs v[q] at positions opos(q) and
    // collects its 16 next-stage inputs from positions ipos(m).  SPLIT (n = 8192, 16384): real parts first, then
    // imaginary parts, through a float buffer of half the size -- 34 / 68 KiB instead of 68 / 136 KiB, i.e. 4 / 2
    // workgroups per CU instead of 2 / 1 (same padding: one element per 16, conflict-free for b32 as for b64).
    auto exchange = [&](v2f (&v)[16], auto opos, v2f (&x)[16], auto ipos) {
        if constexpr (SPLIT) {
            float *lf = reinterpret_cast<float *>(smem) + (threadIdx.x / TPX) * PADN;
            static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; lf[pad(opos(q_))] = v[q].x; });
            __syncthreads();
            static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m].x = lf[pad(ipos(m_))]; });
            __syncthreads();
            static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; lf[pad(opos(q_))] = v[q].y; });
            __syncthreads();
            static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m].y = lf[pad(ipos(m_))]; });
            __syncthreads();
        } else {
            static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; lds[pad(opos(q_))] = v[q]; });
            __syncthreads();
            static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(ipos(m_))]; });
            __syncthreads();
        }
    };
    v2f v[16], x[16];
    // stage 0 (J = 1, s = t): inputs t + m*N/16 straight from global memory, output q at t*16 + q, twiddle W_n^{t*q}
    static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = live ? g_in[t + m * TPX] : v2f{0.f, 0.f}; });
    fft_reg<16, DIR>(x);
    static_for<0, 16>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        v[q] = x[brev<16>(q)];
        if constexpr (q != 0 && N > 16) v[q] = cmul_tw<DIR>(v[q], tw_lookup<N>(tw, t * q));
    });
    if constexpr (NS16 == 1 && RL == 1) {  // n = 16
        static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; if (live) g_out[t * 16 + q] = v[q] * scale; });
        return;
    }
    uint32_t J = 16, jj = 0, sJ = t;  // positions of v[q]: sJ*16 + jj + q*(J/16)
    // middle radix-16 stages
    static_for<1, NS16>([&](auto s_) {
        constexpr int st = decltype(s_)::value;
        constexpr bool last = (st == NS16 - 1) && RL == 1;
        const uint32_t Jp = J / 16, jo = jj, so = sJ;
        exchange(v, [&](auto q_) { return so * 16 + jo + (uint32_t)decltype(q_)::value * Jp; }, x,
                 [&](auto m_) { return t + (uint32_t)decltype(m_)::value * (N / 16); });
        fft_reg<16, DIR>(x);
        jj = t & (J - 1);
        sJ = t - jj;
        static_for<0, 16>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            v[q] = x[brev<16>(q)];
            if constexpr (q != 0 && !last) v[q] = cmul_tw<DIR>(v[q], tw_lookup<N>(tw, sJ * q));
            if constexpr (last) { if (live) g_out[sJ * 16 + jj + q * J] = v[q] * scale; }
        });
        J *= 16;
    });
    // last stage of radix RL < 16: 16/RL butterflies per thread (idx = t + b*TPX < J, so s = 0: no twiddle),
    // inputs idx + m*N/RL, output q straight to global memory at idx + q*J
    if constexpr (RL > 1) {
        const uint32_t Jp = J / 16, jo = jj, so = sJ;
        exchange(v, [&](auto q_) { return so * 16 + jo + (uint32_t)decltype(q_)::value * Jp; }, x,
                 [&](auto i_) {
                     constexpr uint32_t i = decltype(i_)::value;
                     return t + (i / RL) * TPX + (i % RL) * (N / RL);
                 });
        static_for<0, 16 / RL>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f z[RL];
            static_for<0, RL>([&](auto m_) { constexpr int m = decltype(m_)::value; z[m] = x[b * RL + m]; });
            fft_reg<RL, DIR>(z);
            static_for<0, RL>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if (live) g_out[t + b * TPX + q * J] = z[brev<RL>(q)] * scale;
            });
        });
    }
}

// ---------------------------------------------------------------------------
// n = 512 .. 32768: 32 points per thread, register stages 32 x 16 | 32 x 32 | 32 x 32 x 2 | 32 x 32 x 4 | 32 x 16 x 16 |
// 32 x 32 x 16 | 32 x 32 x 32, i.e. ONE exchange at 512 / 1024 and TWO above (k_small16: two / three), each through a
// float buffer -- real parts, then imaginary parts.  n/32 threads per transform, 256-thread workgroups (512 / 1024 at
// 16384 / 32768) with 33 KiB of LDS (66 / 132 KiB): 256 KiB of loads in flight per CU (k_small16 at 8192 / 16384:
// 128 KiB; measured 0.37 / 0.40 -> 0.63 / 0.66 of the roofline).  Same Stockham recurrence per stage, radix R: idx = s*J + j, inputs idx + m*n/R, output q at
// s*R*J + j + q*J times W_n^{s*J*q}.  Positions are padded by one float per 32 (conflict-free b32 accesses).
// ---------------------------------------------------------------------------
// x[brev<R>(q)] *= W_N^{e*q} for q = 1 .. R-1 with 7 + R/8 - 1 table look-ups instead of R - 1:
// W^{e(8a + b)} = W^{8ea} * W^{eb} (one extra rounding on the twiddles that are products, as in k_tile).
template <int R, int N, int DIR>
__device__ __forceinline__ void twiddle_outputs(v2f (&x)[R], const v2f *__restrict__ tw, uint32_t e)
{
    static_assert(R == 16 || R == 32, "radix");
    v2f pb[8], pa[R / 8];
    static_for<1, 8>([&](auto b_) { constexpr int b = decltype(b_)::value; pb[b] = tw_lookup<N>(tw, e * b); });
    static_for<1, R / 8>([&](auto a_) { constexpr int a = decltype(a_)::value; pa[a] = tw_lookup<N>(tw, e * (8 * a)); });
    static_for<1, R>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        constexpr int a = q / 8, b = q % 8, r = brev<R>(q);
        if constexpr (a == 0) x[r] = cmul_tw<DIR>(x[r], pb[b]);
        else if constexpr (b == 0) x[r] = cmul_tw<DIR>(x[r], pa[a]);
        else x[r] = cmul_tw<DIR>(x[r], cmul(pa[a], pb[b]));
    });
}

template <int LGN, int DIR>
__global__ __launch_bounds__((LGN <= 13 ? 256 : (1 << (LGN - 5))), 4) void k_small32(const v2f *__restrict__ src,
                                                                                    v2f *__restrict__ dst,
                                                                                    const v2f *__restrict__ tw,
                                                                                    uint64_t batch, float scale)
{
    static_assert(LGN >= 9 && LGN <= 15, "k_small32 covers n = 512 .. 32768");
    constexpr int N = 1 << LGN;
    constexpr int T = N / 32;                                   // threads per transform = radix-32 butterflies
    constexpr int WG = LGN <= 13 ? 256 : T;                     // workgroup size; XPW transforms per workgroup
    constexpr int XPW = WG / T;
    constexpr int R1 = (LGN == 9 || LGN == 13) ? 16 : 32;       // second radix
    constexpr bool TWO = (32 * R1 == N);                        // n = 512, 1024: two stages, one exchange
    constexpr int R2 = TWO ? 1 : N / (32 * R1);                 // third radix: 2, 4, 16, 16, 32 for 2^11 .. 2^15
    constexpr int B1 = 32 / R1;                                 // butterflies per thread in stages 1 and 2
    constexpr int J2 = 32 * R1;
    constexpr int PN = N + N / 32;                              // padded floats per transform
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t xf = threadIdx.x / T, t = threadIdx.x % T;
    float *lf = reinterpret_cast<float *>(smem) + xf * PN;
    // buffer (SRD) addressing: one per-lane offset, the per-access part is a scalar (no address VGPR per access); the
    // descriptor ends with the last valid transform of the batch, so surplus lanes of a ragged last workgroup read
    // zeros and their stores are dropped
    const uint64_t first = (uint64_t)blockIdx.x * XPW;
    const uint64_t left = batch - first;
    const uint32_t valid_bytes = (uint32_t)(left < (uint64_t)XPW ? left : (uint64_t)XPW) * (N * 8u);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(src + first * N), 0, valid_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dst + first * N, 0, valid_bytes, 0x00020000);
    const uint32_t voff = (xf * N + t) * 8;

    // In-place exchange: register r deposits its value at wbase + woff(r) and is refilled from rbase + roff(r); real
    // parts first (x[r].y still holds the old imaginary part meanwhile), then imaginary parts.  Every position is a
    // lane-dependent base plus a compile-time offset: for the padding P(p) = p + p/32, P(a + b) = P(a) + P(b) whenever b
    // is a multiple of 32 or a + (b mod 32) < 32 -- so each access is one ds instruction with an immediate offset.
    auto exchange = [&](v2f (&x)[32], uint32_t wbase, auto woff, uint32_t rbase, auto roff) {
        static_for<0, 32>([&](auto r_) { constexpr int r = decltype(r_)::value; lf[wbase + woff(r_)] = x[r].x; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int r = decltype(r_)::value; x[r].x = lf[rbase + roff(r_)]; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int r = decltype(r_)::value; lf[wbase + woff(r_)] = x[r].y; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int r = decltype(r_)::value; x[r].y = lf[rbase + roff(r_)]; });
    };
    constexpr auto P = [](uint32_t p) constexpr { return p + (p >> 5); };
    const uint32_t t_hi = t >> 5, t_lo = t & 31;
    const uint32_t rbase = t + t_hi;  // P(t): every read position is element t plus a constant

    v2f x[32];
    // stage 0: radix 32, J = 1, s = t; output q is left in x[brev(q)] and goes to position t*32 + q (P = 33*t + q)
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_NT>(rin, voff, m * T * 8); });
    fft_reg<32, DIR>(x);
    twiddle_outputs<32, N, DIR>(x, tw, t);
    // -> stage 1 (radix R1, J = 32): butterfly b of this thread is idx = t + b*T, input m at idx + m*N/R1
    exchange(x, 33 * t, [](auto r_) { return (uint32_t)brev<32>(decltype(r_)::value); }, rbase, [&](auto i_) {
        constexpr uint32_t i = decltype(i_)::value;
        return P((i / R1) * T + (i % R1) * (N / R1));
    });
    if constexpr (TWO) {
        // last stage: idx = t + b*T < 32 = J, so s = 0: no twiddle, output q at idx + q*32
        static_for<0, B1>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f(&z)[R1] = *reinterpret_cast<v2f(*)[R1]>(&x[b * R1]);
            fft_reg<R1, DIR>(z);
            static_for<0, R1>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                buf_store<AUX_NT>(z[brev<R1>(q)] * scale, rout, voff, (b * T + q * 32) * 8);
            });
        });
    } else {
        static_for<0, B1>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f(&z)[R1] = *reinterpret_cast<v2f(*)[R1]>(&x[b * R1]);
            fft_reg<R1, DIR>(z);
            const uint32_t idx = t + b * T, sJ = idx & ~31u;
            twiddle_outputs<R1, N, DIR>(z, tw, sJ);  // output q: position sJ*R1 + j + q*32
        });
        __syncthreads();  // every read of the first exchange is done before its buffer is rewritten
        // -> stage 2 (radix R2, J = N/R2, s = 0): butterfly b is idx = t + b*T < N/R2, input m at idx + m*N/R2.
        // Output q of stage-1 butterfly b sits at sJ*R1 + j + q*32 with sJ = (t & ~31) + b*T, j = t & 31 (T is a
        // multiple of 32 here): lane part (t & ~31)*R1 + (t & 31), padded by (t >> 5)*R1; constant part b*T*R1 + q*32
        constexpr int B2 = 32 / R2;
        exchange(x, (t - t_lo) * R1 + t_lo + t_hi * R1, [&](auto i_) {
            constexpr uint32_t i = decltype(i_)::value;
            return P((i / R1) * T * R1 + (uint32_t)brev<R1>(i % R1) * 32);
        }, rbase, [&](auto i_) {
            constexpr uint32_t i = decltype(i_)::value;
            return P((i / R2) * T + (i % R2) * (N / R2));
        });
        static_for<0, B2>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f(&z)[R2] = *reinterpret_cast<v2f(*)[R2]>(&x[b * R2]);
            fft_reg<R2, DIR>(z);
            static_for<0, R2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                buf_store<AUX_NT>(z[brev<R2>(q)] * scale, rout, voff, (b * T + q * J2) * 8);
            });
        });
    }
}

static uint32_t small32_xpw(uint32_t lg_n) { return lg_n <= 13 ? 256u / (1u << (lg_n - 5)) : 1u; }
static size_t small32_lds(uint32_t lg_n)
{
    return (size_t)small32_xpw(lg_n) * ((size_t)(1u << lg_n) + (1u << (lg_n - 5))) * sizeof(float);
}
template <int LGN, int DIR>
static hipError_t launch_small32_n(const v2f *src, v2f *dst, const v2f *tw, uint64_t batch, float scale, hipStream_t st)
{
    const uint32_t xpw = small32_xpw(LGN);
    const uint64_t blocks = (batch + xpw - 1) / xpw;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_small32<LGN, DIR>), dim3((uint32_t)blocks), dim3(LGN <= 13 ? 256 : (1 << (LGN - 5))), small32_lds(LGN), st,
                       src, dst, tw, batch, scale);
    return hipGetLastError();
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
    const size_t lds = (lg_n == 4 || (shfl && lg_n <= 7)) ? 0 : (size_t)xpw * (n + n / 16) * (lg_n >= 13 ? sizeof(float) : sizeof(v2f));
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
        case 13: return launch_small32_n<13, DIR>(src, dst, tw, batch, scale, st);
        case 14: return launch_small32_n<14, DIR>(src, dst, tw, batch, scale, st);
        case 15: return launch_small32_n<15, DIR>(src, dst, tw, batch, scale, st);
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

hipError_t launch_small32(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                          hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    uint32_t lg_n = 0;
    while ((1u << lg_n) < n) ++lg_n;
#define FWA_S32(L)                                                                               \
    case L:                                                                                      \
        return dir == FWD ? launch_small32_n<L, FWD>(src, dst, tw, batch, scale, st)             \
                          : launch_small32_n<L, INV>(src, dst, tw, batch, scale, st)
    switch (lg_n) {
        FWA_S32(9); FWA_S32(10); FWA_S32(11); FWA_S32(12); FWA_S32(13); FWA_S32(14); FWA_S32(15);
        default: return hipErrorInvalidValue;
    }
#undef FWA_S32
}

hipError_t setup_small_kernels()
{
    // 16384 / 32768-point transforms need 66 / 132 KiB of dynamic LDS (8192: 33 KiB, inside the default limit)
    hipError_t e = hipSuccess;
    const void *ks[4] = {reinterpret_cast<const void *>(&k_small32<14, FWD>), reinterpret_cast<const void *>(&k_small32<14, INV>),
                         reinterpret_cast<const void *>(&k_small32<15, FWD>), reinterpret_cast<const void *>(&k_small32<15, INV>)};
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)small32_lds(i < 2 ? 14 : 15));
    return e;
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
