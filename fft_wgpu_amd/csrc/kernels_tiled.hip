// kernels_tiled.hip -- (2/3) the building block of the 2- and 3-pass paths: k_tile.
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

template <int LGL, int CW, int DIR, int MODE, bool BUF, int ROLE>
__global__ __launch_bounds__(((1 << LGL) / 16) * CW) void k_tile(TileArgs a)
{
    constexpr int L = 1 << LGL;
    constexpr int TPX = L / 16;
    constexpr int NS16 = LGL / 4;
    constexpr int RL = 1 << (LGL % 4);
    constexpr int PSTR = tile_pstr(L);
    constexpr int AOUT = (ROLE == ROLE_FIRST || ROLE == ROLE_MIDDLE) ? AUX_SC1 : (ROLE == ROLE_LAST ? AUX_NT : AUX_DEFAULT);
    constexpr int AIN = (ROLE == ROLE_FIRST) ? AUX_NT : AUX_DEFAULT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *lds_all = reinterpret_cast<v2f *>(smem);
    auto pad = [](uint32_t p) { return p + (p >> 4); };

    // XCD-aware block -> tile mapping: each XCD gets a contiguous run of tiles (see kernels_1m.hip xcd_block)
    const uint32_t bid = a.xcd_swizzle ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t tile = bid % a.tile_count;
    const uint32_t rest = bid / a.tile_count;
    const uint32_t d1 = rest % a.d1_count;
    const uint64_t b = rest / a.d1_count;
    const v2f *in = a.in + b * a.in_sb + d1 * a.in_s1 + tile * a.in_st;
    v2f *out = a.out + b * a.out_sb + d1 * a.out_s1 + tile * a.out_st;

    // mapping B (FFT index fastest): coalesces every access whose CW FFTs are adjacent in memory
    const uint32_t cB = threadIdx.x & (CW - 1), tB = threadIdx.x / CW;
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
        v2f *lds = lds_all + c0 * PSTR;
        v2f x[16];
        static_for<0, 16>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if constexpr (BUF) x[m] = buf_load<AIN>(rin, vin, m * sin_step);
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
    v2f *lds = lds_all + cB * PSTR;
    const uint32_t t = tB;
    // Four-step twiddle (COLS).  Every output of this thread has index o = t + m*TPX, m = 0..15, so
    // W_T^{col*o} = [W^{col*t} * (W^{col*TPX})^(m&3)] * W^{col*TPX*4*(m>>2)}: four table look-ups
    // (hi[e>>10]*lo[e&1023] each) and short products instead of one look-up pair per output.
    v2f pa[4], pb[4];
    if constexpr (MODE == TILE_COLS) {
        const uint32_t col = tile * CW + cB;
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
            if constexpr (BUF) buf_store<AOUT>(v, rout, (t * pitch32 + cB) * 8, m * (uint32_t)TPX * pitch32 * 8);
            else out[(uint64_t)o * a.pitch + cB] = v;
        } else {
            v = v * a.scale;
            if constexpr (BUF) buf_store<AOUT>(v, rout, (t * ostride32 + cB) * 8, m * (uint32_t)TPX * ostride32 * 8);
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

bool tile_supported(uint32_t lg_l, uint32_t cw)
{
    return (cw == 16 && lg_l >= 6 && lg_l <= 10) || (cw == 32 && lg_l >= 6 && lg_l <= 9);
}

template <int CW, int DIR, int MODE, bool BUF, int ROLE>
static const void *tile_kernel_p(uint32_t lg_l)
{
    switch (lg_l) {
        case 6: return reinterpret_cast<const void *>(&k_tile<6, CW, DIR, MODE, BUF, ROLE>);
        case 7: return reinterpret_cast<const void *>(&k_tile<7, CW, DIR, MODE, BUF, ROLE>);
        case 8: return reinterpret_cast<const void *>(&k_tile<8, CW, DIR, MODE, BUF, ROLE>);
        case 9: return reinterpret_cast<const void *>(&k_tile<9, CW, DIR, MODE, BUF, ROLE>);
        case 10:
            if constexpr (CW == 16) return reinterpret_cast<const void *>(&k_tile<10, CW, DIR, MODE, BUF, ROLE>);
            else return nullptr;
        default: return nullptr;
    }
}
// COLS passes come as first or middle pass, ROWS_T is always the last; the 64-bit-pointer form (BUF = false,
// only above 4-GiB tiles) has no policy bits.
template <int CW, int DIR, int MODE>
static const void *tile_kernel_m(uint32_t lg_l, bool buf, int role)
{
    if (!buf) return tile_kernel_p<CW, DIR, MODE, false, 0>(lg_l);
    if constexpr (MODE == TILE_COLS) {
        if (role == ROLE_MIDDLE) return tile_kernel_p<CW, DIR, MODE, true, ROLE_MIDDLE>(lg_l);
        return tile_kernel_p<CW, DIR, MODE, true, ROLE_FIRST>(lg_l);
    } else {
        return tile_kernel_p<CW, DIR, MODE, true, ROLE_LAST>(lg_l);
    }
}
static const void *tile_kernel(int dir, int mode, uint32_t cw, uint32_t lg_l, bool buf, int role)
{
    if (!tile_supported(lg_l, cw)) return nullptr;
#define FWA_TK(CWV)                                                                                          \
    (dir == FWD ? (mode == TILE_COLS ? tile_kernel_m<CWV, FWD, TILE_COLS>(lg_l, buf, role)                   \
                                     : tile_kernel_m<CWV, FWD, TILE_ROWS_T>(lg_l, buf, role))                \
                : (mode == TILE_COLS ? tile_kernel_m<CWV, INV, TILE_COLS>(lg_l, buf, role)                   \
                                     : tile_kernel_m<CWV, INV, TILE_ROWS_T>(lg_l, buf, role)))
    return cw == 16 ? FWA_TK(16) : FWA_TK(32);
#undef FWA_TK
}
static size_t tile_lds(uint32_t lg_l, uint32_t cw) { return (size_t)cw * tile_pstr(1u << lg_l) * sizeof(v2f); }

// called at plan creation: raises the dynamic-LDS limit of the kernels a plan will launch
hipError_t prepare_tile(uint32_t lg_l, uint32_t cw)
{
    if (!tile_supported(lg_l, cw)) return hipErrorInvalidValue;
    const size_t lds = tile_lds(lg_l, cw);
    if (lds <= 65536) return hipSuccess;
    for (int dir : {FWD, INV})
        for (int mode : {TILE_COLS, TILE_ROWS_T})
            for (int buf = 0; buf < 2; ++buf)
                for (int role : {ROLE_FIRST, ROLE_MIDDLE}) {
                    const void *k = tile_kernel(dir, mode, cw, lg_l, buf != 0, role);
                    if (!k) return hipErrorInvalidValue;
                    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                }
    return hipSuccess;
}

hipError_t launch_tile(int dir, int mode, uint32_t lg_l, const TileArgs &a, uint64_t batch, hipStream_t st)
{
    const uint64_t blocks = batch * a.d1_count * a.tile_count;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one tile?  COLS: L rows of `pitch`; ROWS_T: cw rows of `pitch` in, L outputs of out_stride
    const uint64_t L = 1ull << lg_l, cw = a.cw;
    const uint64_t span_in = (cw * a.pitch + L) * 8, span_out = (L * a.out_stride + cw) * 8;
    const uint64_t span = (mode == TILE_COLS) ? L * a.pitch * 8 + cw * 8 : (span_in > span_out ? span_in : span_out);
    const void *k = tile_kernel(dir, mode, a.cw, lg_l, span < (1ull << 32), (int)a.role);
    if (!k) return hipErrorInvalidValue;
    TileArgs copy = a;
    if (blocks % 8) copy.xcd_swizzle = 0;
    void *args[] = {&copy};
    return hipLaunchKernel(k, dim3((uint32_t)blocks), dim3((uint32_t)((L / 16) * cw)), args, tile_lds(lg_l, a.cw), st);
}

}  // namespace fwa
