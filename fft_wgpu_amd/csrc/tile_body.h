// tile_body.h -- the tile of the multi-pass paths (kernels_tiled.hip: k_tile).
#pragma once
#include "device_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// k_tile: CW FFTs of length L (64 <= L <= 1024) per workgroup along ONE axis of a multi-dimensional view
// of the transform -- the building block of the 2- and 3-pass paths (n = N1*N2[*N3]).  Same register radix-16
// Stockham stages as k_small16; what differs is addressing:
//   COLS  (strided axis): element i of FFT c at in + i*pitch + c; CW adjacent c = one CW*8-byte segment, so
//         loads and stores are coalesced over c.  Output element o is multiplied by the four-step twiddle
//         W_T^{(col0 + c)*o} = hi[e>>10]*lo[e&1023] and stored at out + o*pitch + c (in place allowed).
//   ROWS_T (last axis): FFT c is a contiguous row at in + c*row_pitch; loads are coalesced along the row,
//         the exchange re-maps threads, and output element o of row c goes to out + o*out_stride + c
//         (CW adjacent rows = one segment): the transposed store that restores natural order.
// LDS: one padded array per FFT (pad(p) = p + p/16, conflict-free over the position index as in k_small16);
// the arrays are PSTR elements apart with PSTR = 17 (mod 32): lanes that differ in the FFT index c (the
// fastest lane index of every stage after the first) then hit distinct banks for b64 writes (16-lane groups)
// and b64 reads (32-lane groups).  With PSTR = L + L/16 (a multiple of 16 for L >= 256) those accesses were
// 8- to 16-way bank conflicts.
// ---------------------------------------------------------------------------
// ROLE: cache policy of the global accesses (measured on the 2^20 pipeline: `nt` on user-buffer accesses and
// write-through `sc1` ring stores): ROLE_FIRST (user buffer -> ring: loads nt, stores sc1), ROLE_MIDDLE
// (ring -> ring: stores sc1), ROLE_LAST (ring -> user buffer: stores nt); BUF = false has no policy bits.
constexpr uint32_t tile_pstr(uint32_t L)
{
    uint32_t p = L + L / 16;
    while (p % 32 != 17) ++p;
    return p;
}

// The tile itself: CW FFTs of length L read at `in`, written at `out` (both already offset to the tile), first FFT of
// the tile = column / row `col0` of its matrix.  AIN / AOUT: cache-policy bits of the global loads / stores.
template <int LGL, int CW, int DIR, int MODE, bool BUF, int AIN, int AOUT>
__device__ __forceinline__ void tile_body(const v2f *in, v2f *out, uint32_t col0, const v2f *__restrict__ tw_l,
                                          const v2f *__restrict__ tw_lo, const v2f *__restrict__ tw_hi, uint64_t pitch,
                                          uint64_t out_stride, float scale, v2f *lds_all, uint32_t tid)
{
    constexpr int L = 1 << LGL;
    constexpr int TPX = L / 16;
    constexpr int NS16 = LGL / 4;
    constexpr int RL = 1 << (LGL % 4);
    constexpr int PSTR = tile_pstr(L);
    auto pad = [](uint32_t p) { return p + (p >> 4); };

    // mapping B (FFT index fastest): coalesces every access whose CW FFTs are adjacent in memory
    const uint32_t cB = tid & (CW - 1), tB = tid / CW;
    // mapping A (position fastest): coalesces along a contiguous row
    const uint32_t cA = tid / TPX, tA = tid % TPX;
    const uint32_t c0 = (MODE == TILE_COLS) ? cB : cA, t0 = (MODE == TILE_COLS) ? tB : tA;
    // Addressing.  BUF (every byte offset of the tile < 2^32, checked by the launcher): buffer loads/stores
    // with one 32-bit per-lane offset and a scalar offset per access -- no 64-bit multiply per element
    // (cdna_hip_programming.md T8); otherwise plain 64-bit pointers (only the largest transforms).
    const uint32_t pitch32 = (uint32_t)pitch, ostride32 = (uint32_t)out_stride;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFFu, 0x00020000);
    const uint32_t vin = (MODE == TILE_COLS) ? (t0 * pitch32 + c0) * 8 : (c0 * pitch32 + t0) * 8;
    const uint32_t sin_step = (MODE == TILE_COLS) ? (uint32_t)(L / 16) * pitch32 * 8 : (uint32_t)(L / 16) * 8;

    // stage 0: global -> LDS (L >= 64, so there is always a later stage); inputs i = t0 + m*L/16
    {
        v2f *lds = lds_all + c0 * PSTR;
        v2f x[16];
        static_for<0, 16>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if constexpr (BUF) x[m] = buf_load<AIN>(rin, vin, m * sin_step);
            else x[m] = (MODE == TILE_COLS) ? in[(uint64_t)(t0 + m * (L / 16)) * pitch + c0]
                                            : in[(uint64_t)c0 * pitch + t0 + m * (L / 16)];
        });
        fft_reg<16, DIR>(x);
        static_for<0, 16>([&](auto q_) {  // J = 1: s = t0, output position t0*16 + q, twiddle W_L^{t0*q}
            constexpr int q = decltype(q_)::value;
            v2f v = x[brev<16>(q)];
            if constexpr (q != 0) v = cmul_tw<DIR>(v, tw_lookup<L>(tw_l, t0 * q));
            lds[t0 * 17 + q] = v;  // pad(t0*16 + q) = t0*16 + q + t0
        });
    }
    v2f *lds = lds_all + cB * PSTR;
    const uint32_t t = tB;
    // Four-step twiddle (COLS).  Every output of this thread has index o = t + m*TPX, m = 0..15, so
    // W_T^{col*o} = [W^{col*t} * (W^{col*TPX})^(m&3)] * W^{col*TPX*4*(m>>2)}: four table look-ups
    // (hi[e>>10]*lo[e&1023] each) and short products instead of one look-up pair per output.
    v2f pa[4], pb[4];
    if constexpr (MODE == TILE_COLS) {
        const uint32_t col = col0 + cB;
        auto look = [&](uint32_t e) { return cmul(tw_hi[e >> 10], tw_lo[e & 1023]); };
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
            v = cmul_tw<DIR>(v, cmul(pa[m >> 2], pb[m & 3])) * scale;
            if constexpr (BUF) buf_store<AOUT>(v, rout, (t * pitch32 + cB) * 8, m * (uint32_t)TPX * pitch32 * 8);
            else out[(uint64_t)o * pitch + cB] = v;
        } else {
            v = v * scale;
            if constexpr (BUF) buf_store<AOUT>(v, rout, (t * ostride32 + cB) * 8, m * (uint32_t)TPX * ostride32 * 8);
            else out[(uint64_t)o * out_stride + cB] = v;
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
                if constexpr (q != 0) v = cmul_tw<DIR>(v, tw_lookup<L>(tw_l, sJ * q));
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


static inline size_t tile_lds(uint32_t lg_l, uint32_t cw) { return (size_t)cw * tile_pstr(1u << lg_l) * sizeof(v2f); }

}  // namespace fwa
