// small32_kernel.h -- k_small32 (one-launch kernels for n = 512 .. 32768) as a template shared by its two translation
// units (kernels_small32.hip: 512 .. 4096 + the launcher; kernels_small32b.hip: 8192 .. 32768).
#pragma once
#include "small32_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// (Below 512 the same structure loses badly -- n/32 = 2 .. 8 threads per transform make every load instruction a
// 16..64-byte-per-transform gather: 0.09 / 0.17 / 0.54 of the roofline at 64 / 128 / 256 against 0.70 for k_small16,
// profiles/round2/sweep_small32_below_512.jsonl -- so k_small16 keeps n <= 256.)
// n = 512 .. 32768: 32 points per thread, register stages 32 x 16 | 32 x 32 | 32 x 32 x 2 | 32 x 32 x 4 | 32 x 32 x 8 |
// 32 x 32 x 16 | 32 x 32 x 32, i.e. ONE exchange at 512 / 1024 and TWO above (k_small16: two / three), each through a
// float buffer -- real parts, then imaginary parts.  n/32 threads per transform, 256-thread workgroups (512 / 1024 at
// 16384 / 32768) with 33 KiB of LDS (66 / 132 KiB): 256 KiB of loads in flight per CU (k_small16 at 8192 / 16384:
// 128 KiB; measured 0.37 / 0.40 -> 0.63 / 0.66 of the roofline).  Same Stockham recurrence per stage, radix R: idx = s*J + j, inputs idx + m*n/R, output q at
// s*R*J + j + q*J times W_n^{s*J*q}.  Positions are padded by one float per 32 (conflict-free b32 accesses).
// ---------------------------------------------------------------------------
// The body for workgroup index `blk` (k_small32: blk = blockIdx.x; tools/archive/small32_persist_probe.hip walks it through a
// persistent loop, measured no faster: profiles/round3/probe_small32_persistent_negative.txt).
// PREFETCH: where the twiddle-table look-ups are issued.  At the point of use (0) each costs its wave an exposed cache
// latency between the last data load landing and the exchange; they depend on the thread index only, so they can go out
// BEFORE the data loads (bit 0: the twiddles of stage 0, bit 1: those of stage 1 of the three-stage sizes) or right BEHIND
// them, before the wait for the data (bits 2, 3).  Per size, at the 32-GiB footprint, interleaved, bit-identical results
// (tools/archive/small32_prefetch_probe.hip).  With the plain block -> chunk map (profiles/round5/probe_small32_twiddle_prefetch.jsonl):
// 2^10 0.775 -> 0.790, 2^11 0.740 -> 0.790, 2^12 0.705 -> 0.738, 2^13 0.717 -> 0.739.  With the pair map of one_launch_block
// (device_common.h), which came later and lifts the point-of-use form by more (probe_small32_twiddle_prefetch_pair_map.jsonl):
// 2^10 0.807 -> 0.814 (behind), 2^11 0.806 -> 0.807 (both stages, before), 2^12 0.774 -> 0.778 (behind) -- kept -- and 2^13
// 0.787 -> 0.758: back at the point of use, as 2^14 / 2^15 (spills from stage 1 on, or a worse schedule: - 1 ... - 9 %).
constexpr int small32_prefetch_default(int lgn) { return lgn == 11 ? 3 : (lgn == 10 || lgn == 12) ? 4 : 0; }

template <int LGN, int DIR, int PREFETCH = small32_prefetch_default(LGN)>
__device__ __forceinline__ void small32_body(const v2f *__restrict__ src, v2f *__restrict__ dst, const v2f *__restrict__ tw,
                                             uint64_t batch, float scale, uint64_t blk, uint32_t tid)
{
    static_assert(LGN >= 9 && LGN <= 15, "k_small32 covers n = 512 .. 32768");
    constexpr int N = 1 << LGN;
    constexpr int T = N / 32;                                   // threads per transform = radix-32 butterflies
    constexpr int WG = LGN <= 13 ? 256 : T;                     // workgroup size; XPW transforms per workgroup
    constexpr int XPW = WG / T;
    constexpr int R1 = (LGN == 9) ? 16 : 32;       // second radix
    constexpr bool TWO = (32 * R1 == N);                        // n <= 1024: two stages, one exchange
    constexpr int R2 = TWO ? 1 : N / (32 * R1);                 // third radix: 2, 4, 8, 16, 32 for 2^11 .. 2^15
    constexpr int B1 = 32 / R1;                                 // butterflies per thread in stages 1 and 2
    constexpr int J2 = 32 * R1;
    constexpr int PN = N + N / 32;                              // padded floats per transform
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t xf = tid / T, t = tid % T;
    float *lf = reinterpret_cast<float *>(smem) + xf * PN;
    // buffer (SRD) addressing: one per-lane offset, the per-access part is a scalar (no address VGPR per access); the
    // descriptor ends with the last valid transform of the batch, so surplus lanes of a ragged last workgroup read
    // zeros and their stores are dropped
    const uint64_t first = blk * XPW;
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

    Twiddles<32, N> w0;
    Twiddles<R1, N> w1[B1];
    if constexpr (PREFETCH & 1) twiddle_fetch<32, N>(w0, tw, t);
    if constexpr (!TWO && (PREFETCH & 2))
        static_for<0, B1>([&](auto b_) { constexpr int b = decltype(b_)::value; twiddle_fetch<R1, N>(w1[b], tw, (t + b * T) & ~31u); });
    v2f x[32];
    // stage 0: radix 32, J = 1, s = t; output q is left in x[brev(q)] and goes to position t*32 + q (P = 33*t + q)
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_NT>(rin, voff, m * T * 8); });
    if constexpr (PREFETCH & 4) twiddle_fetch<32, N>(w0, tw, t);   // behind the data loads, before the wait for them
    if constexpr (!TWO && (PREFETCH & 8))
        static_for<0, B1>([&](auto b_) { constexpr int b = decltype(b_)::value; twiddle_fetch<R1, N>(w1[b], tw, (t + b * T) & ~31u); });
    fft_reg<32, DIR>(x);
    if constexpr (PREFETCH & 5) twiddle_apply<32, N, DIR>(x, w0);
    else twiddle_outputs<32, N, DIR>(x, tw, t);
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
            if constexpr (PREFETCH & 10) twiddle_apply<R1, N, DIR>(z, w1[b]);
            else twiddle_outputs<R1, N, DIR>(z, tw, sJ);  // output q: position sJ*R1 + j + q*32
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

template <int LGN, int DIR, int PREFETCH = small32_prefetch_default(LGN)>
__global__ __launch_bounds__((LGN <= 13 ? 256 : (1 << (LGN - 5))), 4) void k_small32(const v2f *__restrict__ src,
                                                                                    v2f *__restrict__ dst,
                                                                                    const v2f *__restrict__ tw,
                                                                                    uint64_t batch, float scale)
{
    small32_body<LGN, DIR, PREFETCH>(src, dst, tw, batch, scale, one_launch_block(), threadIdx.x);
}

static uint32_t small32_xpw(uint32_t lg_n) { return lg_n <= 13 ? 256u / (1u << (lg_n - 5)) : 1u; }  // lg_n >= 6
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

// n = 8192 .. 32768 (kernels_small32b.hip)
hipError_t launch_small32_big(int dir, uint32_t lg_n, const v2f *src, v2f *dst, const v2f *tw, uint64_t batch, float scale,
                              hipStream_t st);

}  // namespace fwa
