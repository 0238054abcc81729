// tools/wave_kernel.h -- k_wave512, the wave-private one-launch kernel for n = 512 (round 5; tools/wave_probe.hip times it
// beside its variants).  NOT in the library any more: it was the product's n = 512 kernel (plan key "wave") while it led
// k_small32<9> by 9 % (0.78 against 0.71, profiles/round5/ab_wave512.jsonl); the block map it led to (device_common.h:
// one_launch_block) then lifted k_small32<9> to 0.785-0.825 against 0.766-0.789 for this kernel over five boxes
// (ab_wave512_after_pair_map.jsonl), and a kernel that loses is recorded and removed.  The product-side wiring that was
// measured and tested (launcher, plan key, parity tests) is profiles/round5/wave512_product_kernel.patch.
//
// One 256-thread workgroup per 64-KiB-aligned chunk, every WAVE walks its own 16 KiB = four whole transforms front to back
// with 32 loads of 512 contiguous bytes (the streaming shape of DESIGN.md 2.1) and never addresses global memory any other
// way (k_small32<9>: 16 threads per transform, 128-byte pieces on both sides).  Load j of lane l is point l + 64 (j mod 8)
// of transform j / 8: every thread already holds the eight stride-n/8 points of a radix-8 butterfly of four transforms, so
// nothing is parked.  n = 8 x 8 x 8 (l = 8a + b, k = k1 + 8 (c + 8 m')), per transform:
//   A  radix 8 over m in registers, twiddle W_512^{l k1}
//   E1 (a, b | k1) -> (b, k1 | a) through the wave's own LDS plane
//   B  radix 8 over a, twiddle W_64^{b c}
//   E2 (b, k1 | c) -> (k1, c | b)
//   C  radix 8 over b: lane k1 + 8c holds X[lane + 64 m'], m' = 0 .. 7 -> 8 stores of 512 contiguous bytes.
// The exchanges are wave-private: LDS instructions of one wave execute in order, so the kernel has no workgroup barrier and
// the four waves of a workgroup drift apart freely.  Real parts, then imaginary parts (as in k_small32), in place in the
// registers; the plane is padded so that every ds access is a lane base + an immediate and hits 32 distinct banks per
// 32-lane group: E1 element (a, b, k1) at 68 k1 + 8a + b, E2 element (b, k1, c) at 72 c + 8b + k1.
//
// What makes it fast (profiles/round5/probe_wave512_variants.jsonl, 32 GiB out of place): the four transforms of a wave
// are taken ONE AT A TIME -- all 32 loads are issued at once, transform f is computed when its eight loads have landed
// (the other 24 still in flight) and stored at once (the later transforms still computing).  The same arithmetic on all four
// transforms side by side (32 registers per stage, then 32 stores) leaves the wave with nothing in flight for the duration of
// its arithmetic: 0.707 against 0.784 of the 8 TB/s roofline -- the rate of the bare loads and stores of this launch shape
// (0.777).  A persistent form that refills a slot for the next chunk as soon as it is stored needs > 128 VGPRs: 0.64-0.67.
// The twiddles W_512^{l k1} and W_64^{b c} depend on the lane only: looked up once per wave, used by its four transforms.
// Same recurrence and twiddle table as every other kernel (fft.wgsl:27-62 generalised to radix 8, processor.rs:43-49).
#pragma once
#include "device_common.h"

namespace fwa {

// KNOCK (tools/wave_probe.hip only; 0 in the library): 1 = no twiddles, 2 = no exchanges, 4 = no arithmetic -- timing
// experiments that compute nothing meaningful.
template <int DIR, int KNOCK = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wave512(const v2f *__restrict__ src,
                                                                                            v2f *__restrict__ dst,
                                                                                            const v2f *__restrict__ tw,
                                                                                            uint64_t n_samples, float scale)
{
    constexpr int N = 512;
    constexpr uint32_t CH = 8192;   // samples per workgroup (64 KiB)
    constexpr int PL = 568;         // floats of a wave's exchange plane: 72 * 7 + 64
    __shared__ float lds_all[4 * PL];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const uint32_t voff = (wv * 2048 + lane) * 8;
    float *lw = lds_all + wv * PL;
    const uint32_t lo = lane & 7u, hi = lane >> 3;
    // the twiddle look-ups go out BEFORE the data loads: vmcnt counts in issue order, a look-up issued behind the 32 loads
    // would make the first transform wait for all of them (measured: - 2 %)
    v2f wa[8], wb[8];
    static_for<1, 8>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        wa[q] = tw_lookup<N>(tw, lane * q);       // W_512^{l k1}
        wb[q] = tw_lookup<N>(tw, 8 * hi * q);     // W_64^{b c}, b = lane >> 3
    });

    // LDS accesses of one wave execute in program order; the fences only keep the compiler from reordering them
    auto fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // in place: register brev(q) deposits stage output q at wbase + wstep*q, register r is refilled from rbase + 8r; real
    // parts first (z[].y still holds the old imaginary part meanwhile)
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
    // radix 8, output q multiplied by w[q]; output q is left in z[brev<8>(q)]
    auto stage = [&](v2f (&z)[8], const v2f (&w)[8], bool twiddle) {
        if constexpr (KNOCK & 4) return;
        fft_reg<8, DIR>(z);
        if constexpr (!(KNOCK & 1))
            if (twiddle)
                static_for<1, 8>([&](auto q_) { constexpr int q = decltype(q_)::value; z[brev<8>(q)] = cmul_tw<DIR>(z[brev<8>(q)], w[q]); });
    };

    // the descriptor ends with the data: lanes of a ragged last chunk read zeros, their stores are dropped
    const uint64_t e0 = (uint64_t)blockIdx.x * CH;   // plain map: the pair map of the other one-launch kernels loses 1-2 % here
    const uint64_t left = n_samples - e0;
    const uint32_t valid = left < CH ? (uint32_t)left * 8u : CH * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(src + e0), 0, valid, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dst + e0, 0, valid, 0x00020000);
    v2f y[4][8];
    static_for<0, 32>([&](auto i_) { constexpr int i = decltype(i_)::value; y[i / 8][i % 8] = buf_load<AUX_NT>(rin, voff, i * 512); });
    static_for<0, 4>([&](auto f_) {
        constexpr int f = decltype(f_)::value;
        v2f(&z)[8] = y[f];
        stage(z, wa, true);                                                      // A
        exchange(z, lane, std::integral_constant<int, 68>{}, 68 * lo + hi);      // E1: write 68 k1 + (8a + b); read 68 k1' + b' + 8a
        stage(z, wb, true);                                                      // B
        exchange(z, lane, std::integral_constant<int, 72>{}, lo + 72 * hi);      // E2: write 72 c + (8b + k1); read 72 c' + k1' + 8b
        stage(z, wb, false);                                                     // C
        static_for<0, 8>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            buf_store<AUX_NT>(z[brev<8>(m)] * scale, rout, voff, (8 * f + m) * 512);
        });
    });
}

}  // namespace fwa
