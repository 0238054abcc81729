// kernels_cols32.hip -- first passes on the 32-point-per-thread network: k_cols32 (2048-point columns, 16 per workgroup) and
// k_colsw (512 x 32 / 256 x 64 column tiles).
#include "rows32.h"

namespace fwa {

// ---------------------------------------------------------------------------
// k_cols32: pass A of a plan whose first factor is 2048 or 4096: the k_small32 network (32 x 32 x 2 | 32 x 32 x 4) on CW
// adjacent COLUMNS of a 2^LGN-row matrix at run-time pitch -- the "column c = tid % CW, butterfly kk = tid / CW" role
// for every stage, so each load / store instruction moves CW*8-byte row segments.  Output in the matrix layout,
// multiplied by the four-step factor W_n^{col*k1} = A[kk][c] * B[j][c] (k1 = kk + off_j), both from the two-level table
// of domain n as in k_p1_gen.  1024 threads, one workgroup per CU:
//   2048 rows x 16 columns (128-byte segments), 137 + 4 KiB of LDS -- the only instantiation that ships;
//   4096 rows x  8 columns ( 64-byte segments: half a cache line per row) was built and measured: correct, and 40-50 %
//   slower than three passes (C5 0.204 ms against 0.137; profiles/round2/sweep_cols4096_negative.jsonl) -- half-line
//   READS cost what half-line writes (k_rows32 at 2048 / 4096-point rows) do not.
// ---------------------------------------------------------------------------
template <int LGN, int CW, int DIR, int AUX_OUT>
__global__ __launch_bounds__(1024, 4) void k_cols32(const v2f *__restrict__ in, v2f *__restrict__ out,
                                                    const v2f *__restrict__ tw, const v2f *__restrict__ tw_lo,
                                                    const v2f *__restrict__ tw_hi, uint32_t pitch, uint64_t in_sb,
                                                    uint64_t out_sb, uint32_t xcd_swizzle)
{
    using G = Rows32<LGN, CW>;
    static_assert(G::WG == 1024 && (CW == 8 || CW == 16), "1024 threads");
    constexpr int N = G::N, T = G::T, PNS = G::PNS, J2 = 1024, R2 = N / 1024, B2 = 32 / R2;
    constexpr int LGCW = CW == 8 ? 3 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    v2f *two = reinterpret_cast<v2f *>(smem + G::LDS_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_map(xcd_swizzle);
    const uint32_t tiles = pitch >> LGCW;
    const uint32_t tile = bid % tiles;
    const uint64_t bt = bid / tiles;
    const uint32_t c = tid & (CW - 1), kk = tid >> LGCW;  // column of the tile, butterfly (0 .. T-1)
    float *lf = lds + c * PNS;
    const uint32_t tbytes = pitch * (N * 8u);  // n * 8 <= 2^31 (launcher)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in + bt * in_sb), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + bt * out_sb, 0, tbytes, 0x00020000);
    const uint32_t voff = (kk * pitch + c) * 8;
    const uint32_t soff = tile * (CW * 8);
    const uint32_t rstep = pitch * 8;  // bytes per matrix row

    v2f x[32];
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_NT>(rin, voff, soff + (m * T) * rstep); });
    const uint32_t col = tile * CW + c;
    auto look = [&](uint32_t e) { return cmul(tw_hi[e >> 10], tw_lo[e & 1023]); };  // e < n
    if (kk < 32) {  // B[j][c] = W_n^{col * off_j}, off_j = (j % B2) * T + (j / B2) * 1024
        const uint32_t off = (kk % B2) * T + (kk / B2) * J2;
        two[kk * CW + c] = look(col * off);
    }
    const v2f A = look(col * kk);

    auto exchange = [&](v2f (&v)[32], uint32_t wbase, auto woff, uint32_t rbase, auto roff) {
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; lf[wbase + woff(r_)] = v[i].x; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; v[i].x = lf[rbase + roff(r_)]; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; lf[wbase + woff(r_)] = v[i].y; });
        __syncthreads();
        static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; v[i].y = lf[rbase + roff(r_)]; });
    };
    constexpr auto P = [](uint32_t p) constexpr { return p + (p >> 5); };
    const uint32_t k_hi = kk >> 5, k_lo = kk & 31;
    const uint32_t rbase = kk + k_hi;

    fft_reg<32, DIR>(x);
    twiddle_outputs<32, N, DIR>(x, tw, kk);   // look-ups at the point of use: prefetched they spill here (- 6 % at 2^23)
    exchange(x, 33 * kk, [](auto r_) { return (uint32_t)brev<32>(decltype(r_)::value); }, rbase, [&](auto i_) {
        constexpr uint32_t i = decltype(i_)::value;
        return P(i * (N / 32));
    });
    fft_reg<32, DIR>(x);
    twiddle_outputs<32, N, DIR>(x, tw, kk & ~31u);
    __syncthreads();
    exchange(x, (kk - k_lo) * 32 + k_lo + k_hi * 32, [&](auto i_) {
        constexpr uint32_t i = decltype(i_)::value;
        return P((uint32_t)brev<32>(i) * 32);
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
            const v2f w = cmul(A, two[(q * B2 + b) * CW + c]);
            buf_store<AUX_OUT>(cmul_tw<DIR>(z[brev<R2>(q)], w), rout, voff, soff + (b * T + q * J2) * rstep);
        });
    });
}

// ---------------------------------------------------------------------------
// k_colsw: pass A with SHORT columns and WIDE tiles: 2^LGN-row columns (LGN = 8, 9), CW = 2^14 /
// 2^LGN adjacent columns per workgroup -- the same 16 Ki points, 512 threads x 32 points, ~72-76 KiB of LDS and two
// workgroups per CU as the 1024 x 16 tile of k_p1_1m / k_p1_gen, but every load instruction moves CW*8 = 256- or 512-byte
// row segments instead of 128-byte ones.  Network: 32 x (N/32), one exchange (k_small32's two-stage form) in the
// "column c = tid % CW, butterfly kk = tid / CW" role.  Output k1 = kk + b*T + 32*q times the four-step factor
// W_n^{col*k1} = A[kk] * Bb[b] * Bq[q] (per column; A in a register, Bb / Bq in LDS: B1 + R1 entries per column instead of
// 32, which is what keeps the 256 x 64 tile under 80 KiB).  The output goes to out + k1*out_sk + tile*out_st + c*8 (bytes):
// matrix layout (out_sk = pitch*8, out_st = CW*8) or tile-contiguous ring (out_sk = CW*8, out_st = N*CW*8: every store
// instruction writes one contiguous 512-byte piece and the RW rows of a last-pass tile are RW*CW*8 contiguous bytes).
// ---------------------------------------------------------------------------
template <int LGN, int CW, int DIR, int AUX_OUT>
__global__ __launch_bounds__(512, 4) void k_colsw(const v2f *__restrict__ in, v2f *__restrict__ out,
                                                  const v2f *__restrict__ tw, const v2f *__restrict__ tw_lo,
                                                  const v2f *__restrict__ tw_hi, uint32_t pitch, uint64_t in_sb,
                                                  uint64_t out_sb, uint32_t out_sk, uint32_t out_st, uint32_t xcd_swizzle)
{
    using G = Rows32<LGN, CW>;
    static_assert(G::WG == 512 && (LGN == 8 || LGN == 9), "512 threads: 512 x 32 or 256 x 64 columns");
    constexpr int N = G::N, T = G::T, PNS = G::PNS, R1 = N / 32, B1 = 32 / R1;
    constexpr int LGCW = 14 - LGN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    v2f *tq = reinterpret_cast<v2f *>(smem + G::LDS_BYTES);  // Bq[R1][CW], then Bb[B1][CW]
    v2f *tb = tq + R1 * CW;
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_map(xcd_swizzle);
    const uint32_t tiles = pitch >> LGCW;
    const uint32_t tile = bid % tiles;
    const uint64_t bt = bid / tiles;
    const uint32_t c = tid & (CW - 1), kk = tid >> LGCW;  // column of the tile, butterfly (0 .. T-1)
    float *lf = lds + c * PNS;
    const uint32_t tbytes = pitch * (N * 8u);  // n * 8 <= 2^31 (launcher)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in + bt * in_sb), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + bt * out_sb, 0, tbytes, 0x00020000);
    const uint32_t voff = (kk * pitch + c) * 8;
    const uint32_t soff = tile * (CW * 8);
    const uint32_t rstep = pitch * 8;  // bytes per matrix row

    constexpr int PF = FWA_PF_COLSW;
    Twiddles<32, N> w0;
    if constexpr (PF & 1) twiddle_fetch<32, N>(w0, tw, kk);
    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP(0);
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_NT>(rin, voff, soff + (m * T) * rstep); });
    FWA_STAMP(1);
    if constexpr (PF & 4) twiddle_fetch<32, N>(w0, tw, kk);
    const uint32_t col = tile * CW + c;
    auto look = [&](uint32_t e) { return cmul(tw_hi[e >> 10], tw_lo[e & 1023]); };  // W_n^e, e < n
    tq[kk * CW + c] = look(col * (32 * kk));           // R1 == T rows: one per thread
    if (kk < B1) tb[kk * CW + c] = look(col * (kk * T));
    const v2f A = look(col * kk);

    fft_reg<32, DIR>(x);
    if constexpr (PF & 5) twiddle_apply<32, N, DIR>(x, w0);
    else twiddle_outputs<32, N, DIR>(x, tw, kk);
    constexpr auto P = [](uint32_t p) constexpr { return p + (p >> 5); };
    auto wpos = [](auto r_) { return (uint32_t)brev<32>(decltype(r_)::value); };
    auto rpos = [&](auto i_) { constexpr uint32_t i = decltype(i_)::value; return P((i / R1) * T + (i % R1) * 32); };
    static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; lf[33 * kk + wpos(r_)] = x[i].x; });
    __syncthreads();
    static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; x[i].x = lf[kk + rpos(r_)]; });
    __syncthreads();
    static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; lf[33 * kk + wpos(r_)] = x[i].y; });
    __syncthreads();
    static_for<0, 32>([&](auto r_) { constexpr int i = decltype(r_)::value; x[i].y = lf[kk + rpos(r_)]; });

    const uint32_t voff_o = kk * out_sk + c * 8;
    const uint32_t soff_o = tile * out_st;
    FWA_STAMP(2);
    static_for<0, B1>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        v2f(&z)[R1] = *reinterpret_cast<v2f(*)[R1]>(&x[b * R1]);
        fft_reg<R1, DIR>(z);
        v2f ab = A;
        if constexpr (b != 0) ab = cmul(A, tb[b * CW + c]);
        static_for<0, R1>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            v2f w = ab;
            if constexpr (q != 0) w = cmul(ab, tq[q * CW + c]);
            buf_store<AUX_OUT>(cmul_tw<DIR>(z[brev<R1>(q)], w), rout, voff_o, soff_o + (b * T + q * 32) * out_sk);
        });
    });
    FWA_STAMP(3);
}

template <int LGN>
static int colsw_lds() { return Rows32<LGN, (1 << (14 - LGN))>::LDS_BYTES + ((1 << (LGN - 5)) + (1024 >> LGN)) * (1 << (14 - LGN)) * 8; }
template <int LGN>
static const void *colsw_kernel(int dir, bool ring)
{
    constexpr int CW = 1 << (14 - LGN);
    return dir == FWD ? (ring ? reinterpret_cast<const void *>(&k_colsw<LGN, CW, FWD, AUX_SC1>) : reinterpret_cast<const void *>(&k_colsw<LGN, CW, FWD, AUX_NT>))
                      : (ring ? reinterpret_cast<const void *>(&k_colsw<LGN, CW, INV, AUX_SC1>) : reinterpret_cast<const void *>(&k_colsw<LGN, CW, INV, AUX_NT>));
}

bool colsw_supported(uint32_t lg_l) { return lg_l == 8 || lg_l == 9; }
uint32_t colsw_width(uint32_t lg_l) { return 1u << (14 - lg_l); }

hipError_t prepare_colsw(uint32_t lg_l)
{
    if (!colsw_supported(lg_l)) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    for (int dir : {FWD, INV})
        for (bool ring : {true, false})
            if (e == hipSuccess)
                e = lg_l == 9 ? hipFuncSetAttribute(colsw_kernel<9>(dir, ring), hipFuncAttributeMaxDynamicSharedMemorySize, colsw_lds<9>())
                              : hipFuncSetAttribute(colsw_kernel<8>(dir, ring), hipFuncAttributeMaxDynamicSharedMemorySize, colsw_lds<8>());
    return e;
}

// n = 2^lg_l * pitch <= 2^28 per transform; tw = half table of W_{2^lg_l}, (tw_lo, tw_hi) = two-level table of W_n.
// tile_ring: tile-contiguous output [tile][k1][CW] (read back by k_rows32 with in_cw = CW), else the matrix layout.
hipError_t launch_colsw(int dir, uint32_t lg_l, bool out_is_ring, bool tile_ring, const v2f *in, v2f *out, const v2f *tw,
                        const v2f *tw_lo, const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb,
                        uint32_t n_transforms, uint32_t xcd_swizzle, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (!colsw_supported(lg_l)) return hipErrorInvalidValue;
    const uint32_t cw = colsw_width(lg_l);
    if (pitch < cw || ((uint64_t)pitch << lg_l) > (1ull << 28) || (pitch & (pitch - 1))) return hipErrorInvalidValue;
    const uint64_t blocks = (uint64_t)n_transforms * (pitch / cw);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (blocks % 8) xcd_swizzle = 0;
    uint32_t out_sk = tile_ring ? cw * 8u : pitch * 8u;
    uint32_t out_st = tile_ring ? (cw * 8u) << lg_l : cw * 8u;
    void *args[] = {&in, &out, &tw, &tw_lo, &tw_hi, &pitch, &in_sb, &out_sb, &out_sk, &out_st, &xcd_swizzle};
    return lg_l == 9 ? hipLaunchKernel(colsw_kernel<9>(dir, out_is_ring), dim3((uint32_t)blocks), dim3(512), args, colsw_lds<9>(), st)
                     : hipLaunchKernel(colsw_kernel<8>(dir, out_is_ring), dim3((uint32_t)blocks), dim3(512), args, colsw_lds<8>(), st);
}

template <int LGN, int CW>
static int cols32_lds() { return Rows32<LGN, CW>::LDS_BYTES + 32 * CW * 8; }
template <int LGN, int CW>
static const void *cols32_kernel(int dir, bool ring)
{
    return dir == FWD ? (ring ? reinterpret_cast<const void *>(&k_cols32<LGN, CW, FWD, AUX_SC1>) : reinterpret_cast<const void *>(&k_cols32<LGN, CW, FWD, AUX_NT>))
                      : (ring ? reinterpret_cast<const void *>(&k_cols32<LGN, CW, INV, AUX_SC1>) : reinterpret_cast<const void *>(&k_cols32<LGN, CW, INV, AUX_NT>));
}

bool cols32_supported(uint32_t lg_l) { return lg_l == 11; }

hipError_t prepare_cols32(uint32_t lg_l)
{
    hipError_t e = hipSuccess;
    for (int dir : {FWD, INV})
        for (bool ring : {true, false})
            if (e == hipSuccess && lg_l == 11)
                e = hipFuncSetAttribute(cols32_kernel<11, 16>(dir, ring), hipFuncAttributeMaxDynamicSharedMemorySize, cols32_lds<11, 16>());
    return cols32_supported(lg_l) ? e : hipErrorInvalidValue;
}

// n = 2^lg_l * pitch <= 2^28 per transform; tw = half table of W_{2^lg_l}, (tw_lo, tw_hi) = two-level table of W_n
hipError_t launch_cols32(int dir, uint32_t lg_l, bool out_is_ring, const v2f *in, v2f *out, const v2f *tw, const v2f *tw_lo,
                         const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb, uint32_t n_transforms,
                         uint32_t xcd_swizzle, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (!cols32_supported(lg_l)) return hipErrorInvalidValue;
    const uint32_t cw = 16;
    if (pitch < 16 || ((uint64_t)pitch << lg_l) > (1ull << 28) || (pitch & (pitch - 1))) return hipErrorInvalidValue;
    const uint64_t blocks = (uint64_t)n_transforms * (pitch / cw);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (blocks % 8) xcd_swizzle = 0;
    void *args[] = {&in, &out, &tw, &tw_lo, &tw_hi, &pitch, &in_sb, &out_sb, &xcd_swizzle};
    return hipLaunchKernel(cols32_kernel<11, 16>(dir, out_is_ring), dim3((uint32_t)blocks), dim3(1024), args, cols32_lds<11, 16>(), st);
}

}  // namespace fwa
