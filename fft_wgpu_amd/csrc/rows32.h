// rows32.h -- k_rows32 (last pass of the two-pass plans) as a template shared by its two translation units
// (kernels_rows32.hip: 512 / 1024-point rows + the launcher; kernels_rows32b.hip: 2048 / 4096-point rows) and the
// Rows32 geometry that the column kernels of kernels_cols32.hip reuse.
#pragma once
#include "small32_common.h"

// Where the stage-0 twiddle look-ups of k_rows32 / k_colsw are issued (as small32_kernel.h's PREFETCH: bit 0 = before the
// data loads, bit 2 = right behind them, 0 = at the point of use).  These passes are bound by their access pattern, not by
// an exposed look-up: one library per value, interleaved in one process (tools/build_variant.sh, tools/ab_libs.py,
// profiles/round5/ab_twiddle_prefetch_pass_kernels.jsonl), the 2^16 .. 2^19 plans gain 0.3-1.4 % with the look-ups first,
// 2^21 / 2^22 (2048 / 4096-point rows) nothing; k_cols32 spills with them (2^23: - 6 %) and keeps them at the point of use.
#ifndef FWA_PF_ROWS32
#define FWA_PF_ROWS32 1
#endif
#ifndef FWA_PF_COLSW
#define FWA_PF_COLSW 1
#endif

namespace fwa {

// ---------------------------------------------------------------------------
// k_rows32: the LAST pass of a two-pass plan n = N1 x L with L = 512, 1024, 2048 or 4096 -- the k_small32 network on 16
// adjacent rows (K1 = 16*tile .. +15, each a contiguous L-point transform in the slab, so the 16 rows of a tile are one
// contiguous 16*L*8-byte chunk) with the transposed store X[K1 + N1*K2] of the four-step algorithm.  The transposition
// costs nothing extra: the last register stage is free to pick its operands from ANY row's exchange buffer, so the
// thread that was (row xf = tid / T, butterfly t = tid % T) while loading becomes (row r = tid % 16, butterfly
// kk = tid / 16) for the last stage -- its outputs K2 then sit beside those of the 15 other rows of the same K2 and a
// store instruction writes 128-byte segments.  Row buffers are skewed (Rows32::SKEW floats mod 32) so that the rows read
// by one instruction fall on different banks.  RW*T threads and RW*PNS*4 bytes of LDS: 256 / 34 KiB at 512-point rows,
// 512 / 69 KiB at 1024, 512 / 68 KiB at 2048 and 1024 / 136 KiB at 4096 (RW = 8 there, see rows32_rows).
// (k_tile covers these lengths with 16 points per thread and two full-complex exchanges: 512-point rows were the slow
// pass of the 2^19 plan, and 2048-point rows did not exist: 2^21 needed three passes.)
// ---------------------------------------------------------------------------
template <int LGN, int RW = 16>
struct Rows32 {
    static constexpr int N = 1 << LGN, T = N / 32, WG = RW * T;
    static constexpr int PN = N + N / 32;
    // Padded floats per row.  In the transposed role a 32-lane group of a ds_read_b32 / ds_write_b32 holds RW adjacent
    // rows x 32/RW adjacent positions: with a row skew of 32/RW floats (mod 32) the RW x 32/RW addresses fall on 32
    // different banks (RW = 8: 4, RW = 16: 2); from 32 rows on, any odd skew does.  (Round 2 used 17 for every RW: one
    // 2-way conflict per access at 16 rows, 20 % conflict cycles at 8 rows: profiles/round3/pmc_counters_sizes_2p21_2p24.txt.)
    static constexpr int SKEW = RW >= 32 ? 17 : 32 / RW;
    static constexpr int PNS = PN + ((SKEW - PN % 32) + 32) % 32;
    static constexpr int LDS_BYTES = RW * PNS * 4;
};

// rows per workgroup: 16 (128-byte store segments); 8 at 2048-point rows -- 64-byte segments (the size of an L2 -> fabric
// write request anyway) but two 512-thread workgroups per CU instead of one of 1024: 2^21 1.715 -> 1.633 ms, 2^22 1.892 ->
// 1.845 ms; at 1024-point rows (two workgroups per CU either way) 8 rows are 3 % slower (profiles/round2/probe_rows32_8_rows.txt)
constexpr int rows32_rows(int lgn) { return lgn >= 11 ? 8 : 16; }

// IN_CW = 0: the slab is the n1 x N matrix (row k1 = N contiguous samples).  IN_CW = 32 / 64: the slab is the
// tile-contiguous ring k_colsw writes, [n2 / IN_CW][k1][n2 % IN_CW] -- the RW rows of a tile are RW*IN_CW*8 contiguous
// bytes per column tile, and a load instruction (T = N/32 >= IN_CW consecutive n2 per row) moves IN_CW*8-byte pieces.
template <int LGN, int DIR, int RW = 16, int IN_CW = 0>
__global__ __launch_bounds__((RW << (LGN - 5)), 4) void k_rows32(const v2f *__restrict__ in, v2f *__restrict__ out,
                                                                  const v2f *__restrict__ tw, uint32_t n1, uint64_t in_sb,
                                                                  uint64_t out_sb, float scale, uint32_t xcd_swizzle)
{
    static_assert(LGN >= 9 && LGN <= 12, "k_rows32 covers row lengths 512 .. 4096");
    using G = Rows32<LGN, RW>;
    constexpr int N = G::N, T = G::T, PNS = G::PNS;
    constexpr int LGRW = RW == 8 ? 3 : 4;
    constexpr int R1 = (LGN == 9) ? 16 : 32;
    constexpr bool TWO = (32 * R1 == N);
    constexpr int R2 = TWO ? 1 : N / (32 * R1);
    constexpr int B1 = 32 / R1;
    constexpr int J2 = 32 * R1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_map(xcd_swizzle);
    const uint32_t tiles = n1 >> LGRW;
    const uint32_t tile = bid % tiles;
    const uint64_t bt = bid / tiles;
    const uint32_t xf = tid / T, t = tid % T;  // loading role: row, butterfly
    const uint32_t r = tid & (RW - 1), kk = tid >> LGRW;  // storing role
    float *lfw = lds + xf * PNS;
    const float *lfr_same = lfw;
    const float *lfr_t = lds + r * PNS;
    static_assert(IN_CW == 0 || (T % IN_CW == 0), "a load instruction covers whole column tiles");
    const __amdgpu_buffer_rsrc_t rin =
        IN_CW ? __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in + bt * in_sb), 0, n1 * (N * 8u), 0x00020000)
              : __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in + bt * in_sb + (uint64_t)tile * RW * N), 0, (uint32_t)RW * N * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + bt * out_sb, 0, n1 * (N * 8u), 0x00020000);
    // sample n2 = t + T*m of row k1 = RW*tile + xf
    const uint32_t voff = IN_CW ? (((t / (IN_CW ? IN_CW : 1)) * n1 + tile * RW + xf) * (uint32_t)IN_CW + (t % (IN_CW ? IN_CW : 1))) * 8 : (xf * N + t) * 8;
    const uint32_t mstep = IN_CW ? n1 * (T * 8u) : T * 8u;  // bytes between samples n2 and n2 + T of a row

    auto exchange = [&](v2f (&x)[32], float *wp, uint32_t wbase, auto woff, const float *rp, uint32_t rbase, auto roff) {
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; wp[wbase + woff(r_)] = x[i].x; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; x[i].x = rp[rbase + roff(r_)]; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; wp[wbase + woff(r_)] = x[i].y; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; x[i].y = rp[rbase + roff(r_)]; });
    };
    constexpr auto P = [](uint32_t p) constexpr { return p + (p >> 5); };
    const uint32_t t_hi = t >> 5, t_lo = t & 31;

    constexpr int PF = LGN <= 10 ? FWA_PF_ROWS32 : 0;   // 512 / 1024-point rows (the 2^16 .. 2^19 plans)
    Twiddles<32, N> w0;
    Twiddles<R1, N> w1[B1];
    if constexpr (PF & 1) twiddle_fetch<32, N>(w0, tw, t);
    if constexpr (!TWO && (PF & 2))
        static_for<0, B1>([&](auto b_) { constexpr int b = decltype(b_)::value; twiddle_fetch<R1, N>(w1[b], tw, (t + b * T) & ~31u); });
    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP_B(0);
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_DEFAULT>(rin, voff, m * mstep); });
    FWA_STAMP_B(1);
    if constexpr (PF & 4) twiddle_fetch<32, N>(w0, tw, t);
    if constexpr (!TWO && (PF & 8))
        static_for<0, B1>([&](auto b_) { constexpr int b = decltype(b_)::value; twiddle_fetch<R1, N>(w1[b], tw, (t + b * T) & ~31u); });
    fft_reg<32, DIR>(x);
    if constexpr (PF & 5) twiddle_apply<32, N, DIR>(x, w0);
    else twiddle_outputs<32, N, DIR>(x, tw, t);
    // transposed store role: output K2 of row r goes to element (16*tile + r) + n1*K2
    const uint32_t voff_o = (kk * n1 + r) * 8;
    const uint32_t soff_o = tile * (RW * 8);
    const uint32_t kstep = n1 * 8;  // bytes per unit of K2
    if constexpr (TWO) {
        // -> last stage (radix R1, J = 32, s = 0) in the storing role: butterfly idx = kk + b*T of row r
        exchange(x, lfw, 33 * t, [](auto r_) { return (uint32_t)brev<32>(decltype(r_)::value); }, lfr_t, kk + (kk >> 5), [&](auto i_) {
            constexpr uint32_t i = decltype(i_)::value;
            return P((i / R1) * T + (i % R1) * (N / R1));
        });
        static_for<0, B1>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f(&z)[R1] = *reinterpret_cast<v2f(*)[R1]>(&x[b * R1]);
            fft_reg<R1, DIR>(z);
            static_for<0, R1>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                buf_store<AUX_NT>(z[brev<R1>(q)] * scale, rout, voff_o, soff_o + (b * T + q * 32) * kstep);
            });
        });
    } else {
        exchange(x, lfw, 33 * t, [](auto r_) { return (uint32_t)brev<32>(decltype(r_)::value); }, lfr_same, t + t_hi, [&](auto i_) {
            constexpr uint32_t i = decltype(i_)::value;
            return P((i / R1) * T + (i % R1) * (N / R1));
        });
        static_for<0, B1>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f(&z)[R1] = *reinterpret_cast<v2f(*)[R1]>(&x[b * R1]);
            fft_reg<R1, DIR>(z);
            const uint32_t idx = t + b * T, sJ = idx & ~31u;
            if constexpr (PF & 10) twiddle_apply<R1, N, DIR>(z, w1[b]);
            else twiddle_outputs<R1, N, DIR>(z, tw, sJ);
        });
        __syncthreads();
        // -> last stage (radix R2, J = N/R2, s = 0) in the storing role: butterfly idx = kk + b*T of row r
        constexpr int B2 = 32 / R2;
        exchange(x, lfw, (t - t_lo) * R1 + t_lo + t_hi * R1, [&](auto i_) {
            constexpr uint32_t i = decltype(i_)::value;
            return P((i / R1) * T * R1 + (uint32_t)brev<R1>(i % R1) * 32);
        }, lfr_t, kk + (kk >> 5), [&](auto i_) {
            constexpr uint32_t i = decltype(i_)::value;
            return P((i / R2) * T + (i % R2) * (N / R2));
        });
        static_for<0, B2>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            v2f(&z)[R2] = *reinterpret_cast<v2f(*)[R2]>(&x[b * R2]);
            fft_reg<R2, DIR>(z);
            static_for<0, R2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                buf_store<AUX_NT>(z[brev<R2>(q)] * scale, rout, voff_o, soff_o + (b * T + q * J2) * kstep);
            });
        });
    }
    FWA_STAMP_B(3);
}

// kernel entry points by row length; lg_l = 11, 12 live in kernels_rows32b.hip
const void *rows32_kernel_small(uint32_t lg_l, int dir, uint32_t in_cw);
const void *rows32_kernel_big(uint32_t lg_l, int dir, uint32_t in_cw);
const void *rows32_kernel_4096(int dir, uint32_t in_cw);  // kernels_rows32c.hip

template <int LGN, int DIR, int IN_CW>
static const void *rows32_kernel() { return reinterpret_cast<const void *>(&k_rows32<LGN, DIR, rows32_rows(LGN), IN_CW>); }
template <int LGN>
static const void *rows32_kernel_of(int dir, uint32_t in_cw)
{
    if constexpr (LGN >= 10) {
        if (in_cw == 32) return dir == FWD ? rows32_kernel<LGN, FWD, 32>() : rows32_kernel<LGN, INV, 32>();
        if constexpr (LGN >= 11)
            if (in_cw == 64) return dir == FWD ? rows32_kernel<LGN, FWD, 64>() : rows32_kernel<LGN, INV, 64>();
    }
    return dir == FWD ? rows32_kernel<LGN, FWD, 0>() : rows32_kernel<LGN, INV, 0>();
}

}  // namespace fwa
