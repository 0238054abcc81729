// kernels_tiled.hip -- (2/3) multi-pass building blocks: k_tile16, strided radix passes, permute.
#include "device_common.h"

namespace fwa {

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
// POL: cache policy of the global accesses (measured on the 2^20 pipeline: `nt` on user-buffer accesses and
// write-through `sc1` ring stores): 0 = default everywhere, 1 = first pass (user buffer -> ring: loads nt,
// stores sc1), 2 = ring -> ring (stores sc1), 3 = last pass (ring -> user buffer: stores nt)
template <int LGL, int DIR, int MODE, bool BUF, int POL>
__global__ __launch_bounds__((1 << LGL)) void k_tile16(TileArgs a)
{
    constexpr int L = 1 << LGL;
    constexpr int TPX = L / 16;
    constexpr int NS16 = LGL / 4;
    constexpr int RL = 1 << (LGL % 4);
    constexpr int PADN = L + L / 16;
    constexpr int AOUT = (POL == 1 || POL == 2) ? AUX_SC1 : (POL == 3 ? AUX_NT : AUX_DEFAULT);
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
            if constexpr (BUF) x[m] = buf_load<(POL == 1 ? AUX_NT : AUX_DEFAULT)>(rin, vin, m * sin_step);
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

template <int DIR, int MODE, bool BUF, int POL>
static const void *tile16_kernel_p(uint32_t lg_l)
{
    switch (lg_l) {
        case 6: return reinterpret_cast<const void *>(&k_tile16<6, DIR, MODE, BUF, POL>);
        case 7: return reinterpret_cast<const void *>(&k_tile16<7, DIR, MODE, BUF, POL>);
        case 8: return reinterpret_cast<const void *>(&k_tile16<8, DIR, MODE, BUF, POL>);
        case 9: return reinterpret_cast<const void *>(&k_tile16<9, DIR, MODE, BUF, POL>);
        case 10: return reinterpret_cast<const void *>(&k_tile16<10, DIR, MODE, BUF, POL>);
        default: return nullptr;
    }
}
// COLS passes come as first (policy 1) or middle (2) pass, ROWS_T is always the last (3); the 64-bit-pointer
// fallback (BUF = false, only above 4-GiB tiles) has no policy bits.  pol 0 = default policies (A/B timing).
template <int DIR, int MODE>
static const void *tile16_kernel(uint32_t lg_l, bool buf, int pol)
{
    if (!buf) return tile16_kernel_p<DIR, MODE, false, 0>(lg_l);
    if constexpr (MODE == TILE_COLS) {
        if (pol == 1) return tile16_kernel_p<DIR, MODE, true, 1>(lg_l);
        if (pol == 2) return tile16_kernel_p<DIR, MODE, true, 2>(lg_l);
        return tile16_kernel_p<DIR, MODE, true, 0>(lg_l);
    } else {
        if (pol == 3) return tile16_kernel_p<DIR, MODE, true, 3>(lg_l);
        return tile16_kernel_p<DIR, MODE, true, 0>(lg_l);
    }
}
static size_t tile16_lds(uint32_t lg_l) { return (size_t)16 * ((1u << lg_l) + (1u << lg_l) / 16) * sizeof(v2f); }

// called at plan creation: raises the dynamic-LDS limit of the kernels a plan will launch (L >= 512)
hipError_t prepare_tile16(uint32_t lg_l)
{
    const size_t lds = tile16_lds(lg_l);
    if (lds <= 65536) return hipSuccess;
    const void *ks[14] = {
        tile16_kernel<FWD, TILE_COLS>(lg_l, true, 0),  tile16_kernel<FWD, TILE_COLS>(lg_l, true, 1),
        tile16_kernel<FWD, TILE_COLS>(lg_l, true, 2),  tile16_kernel<FWD, TILE_ROWS_T>(lg_l, true, 0),
        tile16_kernel<FWD, TILE_ROWS_T>(lg_l, true, 3), tile16_kernel<FWD, TILE_COLS>(lg_l, false, 0),
        tile16_kernel<FWD, TILE_ROWS_T>(lg_l, false, 0),
        tile16_kernel<INV, TILE_COLS>(lg_l, true, 0),  tile16_kernel<INV, TILE_COLS>(lg_l, true, 1),
        tile16_kernel<INV, TILE_COLS>(lg_l, true, 2),  tile16_kernel<INV, TILE_ROWS_T>(lg_l, true, 0),
        tile16_kernel<INV, TILE_ROWS_T>(lg_l, true, 3), tile16_kernel<INV, TILE_COLS>(lg_l, false, 0),
        tile16_kernel<INV, TILE_ROWS_T>(lg_l, false, 0)};
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
    const void *k = tile16_kernel<DIR, MODE>(lg_l, span < (1ull << 32), (int)(a.flags >> 8) & 3);
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

}  // namespace fwa
