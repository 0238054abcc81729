// kernels_lab_small.hip -- LABORATORY build only (libfft_wgpu_amd_lab.so; `make lab`): the one-launch kernel family that
// measured slower than the shipped k_chunk (n <= 256) / k_small32 (512 .. 32768) and is kept because it holds the
// wavefront-shuffle butterfly exchange BASELINE.json's north_star names (measured 3-10 % slower than the LDS exchange):
//   k_small16 (small_reg = 3: direct addressing, 16 points per thread, 16 <= n <= 4096; = 2 adds the `__shfl_xor` exchange
//   at n = 32 / 64 / 128).
// Round 6 removed k_lds_small (LDS radix 2), k_tiny16 / k_tiny2 (n < 16) from this file: their only loader was a
// bit-identity test (profiles/round6/lab_pruned_families.patch is the code as it was).
#include "device_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// small transforms, 16 <= n <= 4096: register radix-16 Stockham (the default for n <= 256; from 512 on the plan uses
// k_small32 unless small_reg = 3).  Each thread owns 16 points; a transform
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

template <int LGN, int DIR, bool SHFL = false>
__global__ __launch_bounds__(256) void k_small16(const v2f *__restrict__ src,
                                                                                  v2f *__restrict__ dst,
                                                                                  const v2f *__restrict__ tw,
                                                                                  uint64_t batch, float scale)
{
    constexpr int N = 1 << LGN;
    constexpr int TPX = N / 16;                       // threads per transform
    constexpr int WG = 256;
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
    // collects its 16 next-stage inputs from positions ipos(m).
    auto exchange = [&](v2f (&v)[16], auto opos, v2f (&x)[16], auto ipos) {
        static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; lds[pad(opos(q_))] = v[q]; });
        __syncthreads();
        static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(ipos(m_))]; });
        __syncthreads();
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

template <int DIR>
static hipError_t launch_small16_dir(const v2f *src, v2f *dst, const v2f *tw, uint32_t lg_n, uint64_t batch, float scale,
                                     bool shfl, hipStream_t st)
{
    const uint32_t n = 1u << lg_n;
    if (lg_n < 4 || lg_n > 12) return hipErrorInvalidValue;  // n = 8192 .. 32768: k_small32 only
    const uint32_t wg = 256;
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

}  // namespace fwa
