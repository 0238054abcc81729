// tools/wave_kernel_variants.h -- the alternatives of the n = 512 wave-private kernel that were measured and NOT shipped
// (tools/wave_probe.hip; results: profiles/round5/probe_wave512_variants.jsonl): k_wave512 here = the four transforms of a
// wave side by side (0.707 of the roofline), k_wave512s<PERSIST = false> = what the library shipped for a while as fwa::k_wave512
// (tools/wave_kernel.h, 0.784), k_wave512s<PERSIST = true> = the persistent form (0.64-0.67).
//
// One 256-thread workgroup per 64-KiB-aligned chunk, every WAVE walks its own 16 KiB = four whole transforms front to back
// with 32 loads of 512 contiguous bytes (the streaming shape of DESIGN.md 2.1) -- and, unlike k_small32<9> (16 threads per
// transform: 128-byte pieces on both sides), never addresses global memory any other way.  Load j of lane l is point
// l + 64 (j mod 8) of transform j / 8: every thread already holds the eight stride-n/8 points of a radix-8 butterfly of
// four transforms, so there is no parking pass.  n = 8 x 8 x 8 (l = 8a + b, k = k1 + 8 (c + 8 m')):
//   A  radix 8 over m in registers, twiddle W_512^{l k1}
//   E1 (a, b | k1) -> (b, k1 | a) through the wave's own LDS region
//   B  radix 8 over a, twiddle W_64^{b c}
//   E2 (b, k1 | c) -> (k1, c | b)
//   C  radix 8 over b: lane k1 + 8c holds X[lane + 64 m'], m' = 0 .. 7 -> 32 stores of 512 contiguous bytes.
// The exchanges are wave-private: LDS instructions of one wave execute in order, so there is no workgroup barrier in the
// kernel and the four waves of a workgroup drift apart freely.  Real parts, then imaginary parts (as in k_small32), in
// place in the registers; planes padded so that every ds access is a lane base + an immediate and hits 32 distinct banks
// per 32-lane group: E1 element (a, b, k1) at 68 k1 + 8a + b, E2 element (b, k1, c) at 72 c + 8b + k1.
// Same recurrence and twiddle table as every other kernel (fft.wgsl:27-62 generalised to radix 8, processor.rs:43-49).
#pragma once
#include "device_common.h"

namespace fwa_probe {
using namespace fwa;

// KNOCK (tools/wave_probe.hip only; 0 in the library): 1 = no twiddles, 2 = no exchanges, 4 = no arithmetic -- timing
// experiments that compute nothing meaningful.
template <int DIR, int KNOCK = 0>
__global__ __launch_bounds__(256, 4) void k_wave512(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                    const v2f *__restrict__ tw, uint64_t n_samples, float scale)
{
    constexpr int N = 512;
    constexpr uint32_t CH = 8192;   // samples per workgroup (64 KiB)
    constexpr int PL = 568;         // floats per transform plane: 72 * 7 + 64
    __shared__ float lds_all[4 * 4 * PL];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const uint64_t e0 = (uint64_t)blockIdx.x * CH;
    const uint64_t left = n_samples - e0;
    const uint32_t valid = left < CH ? (uint32_t)left * 8u : CH * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(src + e0), 0, valid, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dst + e0, 0, valid, 0x00020000);
    const uint32_t voff = (wv * 2048 + lane) * 8;
    float *lw = lds_all + wv * (4 * PL);
    const uint32_t lo = lane & 7u, hi = lane >> 3;

    v2f y[32];
    static_for<0, 32>([&](auto i_) { constexpr int i = decltype(i_)::value; y[i] = buf_load<AUX_NT>(rin, voff, i * 512); });

    // Wave-private exchange, in place: register (f, r) deposits at wbase + f*PL + wstep*r' and is refilled from
    // rbase + f*PL + 8*r; real parts first (y[].y still holds the old imaginary part meanwhile).  `src_of(r')` = the
    // register that holds element r' of the stage's output (bit-reversed by fft_reg).
    auto exchange = [&](uint32_t wbase, auto wstep_, uint32_t rbase) {
        constexpr int wstep = decltype(wstep_)::value;
        if constexpr (KNOCK & 2) return;
        static_for<0, 32>([&](auto i_) {
            constexpr int f = decltype(i_)::value / 8, q = decltype(i_)::value % 8;
            lw[wbase + f * PL + wstep * q] = y[8 * f + brev<8>(q)].x;
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        static_for<0, 32>([&](auto i_) {
            constexpr int f = decltype(i_)::value / 8, r = decltype(i_)::value % 8;
            y[8 * f + r].x = lw[rbase + f * PL + 8 * r];
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        static_for<0, 32>([&](auto i_) {
            constexpr int f = decltype(i_)::value / 8, q = decltype(i_)::value % 8;
            lw[wbase + f * PL + wstep * q] = y[8 * f + brev<8>(q)].y;
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        static_for<0, 32>([&](auto i_) {
            constexpr int f = decltype(i_)::value / 8, r = decltype(i_)::value % 8;
            y[8 * f + r].y = lw[rbase + f * PL + 8 * r];
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // radix 8 on the four transforms, output q of transform f multiplied by W_512^{e q}; left in y[8f + brev<8>(q)]
    auto stage = [&](uint32_t e, bool twiddle) {
        v2f w[8];
        if constexpr (KNOCK & 4) return;
        if constexpr (KNOCK & 1) twiddle = false;
        if (twiddle) static_for<1, 8>([&](auto q_) { constexpr int q = decltype(q_)::value; w[q] = tw_lookup<N>(tw, e * q); });
        static_for<0, 4>([&](auto f_) {
            constexpr int f = decltype(f_)::value;
            v2f(&z)[8] = *reinterpret_cast<v2f(*)[8]>(&y[8 * f]);
            fft_reg<8, DIR>(z);
            if (twiddle)
                static_for<1, 8>([&](auto q_) { constexpr int q = decltype(q_)::value; z[brev<8>(q)] = cmul_tw<DIR>(z[brev<8>(q)], w[q]); });
        });
    };

    stage(lane, true);                                          // A: W_512^{l k1}
    exchange(lane, std::integral_constant<int, 68>{}, 68 * lo + hi);   // E1: write 68 k1 + (8a + b); read 68 k1' + b' + 8a
    stage(8 * hi, true);                                        // B: W_64^{b c} = W_512^{8 b c}, b = lane >> 3
    exchange(lane, std::integral_constant<int, 72>{}, lo + 72 * hi);   // E2: write 72 c + (8b + k1); read 72 c' + k1' + 8b
    stage(0, false);                                            // C
    static_for<0, 32>([&](auto i_) {
        constexpr int f = decltype(i_)::value / 8, m = decltype(i_)::value % 8;
        buf_store<AUX_NT>(y[8 * f + brev<8>(m)] * scale, rout, voff, (8 * f + m) * 512);
    });
}


// ---- the same transform, one TRANSFORM at a time, optionally persistent ------------------------------------------------
// k_wave512 above computes its four transforms side by side: all 32 loads land, then three stages and two exchanges over 32
// registers, then 32 stores -- for the duration of the arithmetic (8 % of the kernel: profiles/round5/probe_wave512_knockouts.txt)
// the wave has nothing in flight.  Here a wave still issues its 32 loads at once but then takes the transforms one by one
// (A, E1, B, E2, C on 8 registers; one 568-float LDS plane per wave), stores each as soon as it is done and -- PERSIST --
// immediately refills the freed registers with the same transform slot of its next chunk (chunk index + gridDim.x), so that
// three of its four slots are in flight while one computes.  The twiddles W_512^{l k1} and W_64^{b c} depend on the lane only
// and are looked up once per wave.
template <int DIR, bool PERSIST, int KNOCK = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wave512s(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                     const v2f *__restrict__ tw, uint64_t n_samples, float scale)
{
    constexpr int N = 512;
    constexpr uint32_t CH = 8192;
    constexpr int PL = 568;
    __shared__ float lds_all[4 * PL];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const uint32_t voff = (wv * 2048 + lane) * 8;
    float *lw = lds_all + wv * PL;
    const uint32_t lo = lane & 7u, hi = lane >> 3;
    const uint64_t n_chunks = (n_samples + CH - 1) / CH;
    auto rsrc = [&](const v2f *base, uint64_t chunk) {
        const uint64_t e0 = chunk * CH, left = n_samples - e0;
        const uint32_t valid = left < CH ? (uint32_t)left * 8u : CH * 8u;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(base + e0), 0, valid, 0x00020000);
    };
    v2f wa[8], wb[8];
    static_for<1, 8>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        wa[q] = tw_lookup<N>(tw, lane * q);
        wb[q] = tw_lookup<N>(tw, 8 * hi * q);
    });
    auto fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto exchange = [&](v2f (&z)[8], uint32_t wbase, auto wstep_, uint32_t rbase) {
        constexpr int wstep = decltype(wstep_)::value;
        if constexpr (KNOCK & 2) return;
        static_for<0, 8>([&](auto q_) { constexpr int q = decltype(q_)::value; lw[wbase + wstep * q] = z[brev<8>(q)].x; });
        fence();
        static_for<0, 8>([&](auto r_) { constexpr int r = decltype(r_)::value; z[r].x = lw[rbase + 8 * r]; });
        fence();
        static_for<0, 8>([&](auto q_) { constexpr int q = decltype(q_)::value; lw[wbase + wstep * q] = z[brev<8>(q)].y; });
        fence();
        static_for<0, 8>([&](auto r_) { constexpr int r = decltype(r_)::value; z[r].y = lw[rbase + 8 * r]; });
        fence();
    };
    auto stage = [&](v2f (&z)[8], const v2f (&w)[8], bool twiddle) {
        if constexpr (KNOCK & 4) return;
        fft_reg<8, DIR>(z);
        if constexpr (!(KNOCK & 1))
            if (twiddle)
                static_for<1, 8>([&](auto q_) { constexpr int q = decltype(q_)::value; z[brev<8>(q)] = cmul_tw<DIR>(z[brev<8>(q)], w[q]); });
    };

    uint64_t chunk = blockIdx.x;
    __amdgpu_buffer_rsrc_t rin = rsrc(src, chunk), rout = rsrc(dst, chunk);
    v2f y[4][8];
    static_for<0, 32>([&](auto i_) { constexpr int i = decltype(i_)::value; y[i / 8][i % 8] = buf_load<AUX_NT>(rin, voff, i * 512); });
    for (;;) {
        const uint64_t next = chunk + gridDim.x;
        const bool more = PERSIST && next < n_chunks;
        if (more) rin = rsrc(src, next);
        static_for<0, 4>([&](auto f_) {
            constexpr int f = decltype(f_)::value;
            v2f(&z)[8] = y[f];
            stage(z, wa, true);
            exchange(z, lane, std::integral_constant<int, 68>{}, 68 * lo + hi);
            stage(z, wb, true);
            exchange(z, lane, std::integral_constant<int, 72>{}, lo + 72 * hi);
            stage(z, wb, false);
            static_for<0, 8>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                buf_store<AUX_NT>(z[brev<8>(m)] * scale, rout, voff, (8 * f + m) * 512);
            });
            if (more)
                static_for<0, 8>([&](auto m_) { constexpr int m = decltype(m_)::value; z[m] = buf_load<AUX_NT>(rin, voff, (8 * f + m) * 512); });
        });
        if (!more) break;
        chunk = next;
        rout = rsrc(dst, chunk);
    }
}

}  // namespace fwa_probe
