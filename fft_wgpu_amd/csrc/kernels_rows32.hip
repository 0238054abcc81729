// kernels_rows32.hip -- the 32-point-per-thread network of k_small32 as passes of the multi-pass plans: k_rows32 (last
// pass: 512 .. 4096-point rows with the transposed store) and k_cols32 (first pass: 2048 / 4096-point columns).
#include "small32_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// k_rows32: the LAST pass of a two-pass plan n = N1 x L with L = 512, 1024, 2048 or 4096 -- the k_small32 network on 16
// adjacent rows (K1 = 16*tile .. +15, each a contiguous L-point transform in the slab, so the 16 rows of a tile are one
// contiguous 16*L*8-byte chunk) with the transposed store X[K1 + N1*K2] of the four-step algorithm.  The transposition
// costs nothing extra: the last register stage is free to pick its operands from ANY row's exchange buffer, so the
// thread that was (row xf = tid / T, butterfly t = tid % T) while loading becomes (row r = tid % 16, butterfly
// kk = tid / 16) for the last stage -- its outputs K2 then sit beside those of the 15 other rows of the same K2 and a
// store instruction writes 128-byte segments.  Row buffers are skewed to 17 mod 32 floats so that the 16 rows read
// by one instruction fall on different banks.  RW*T threads and RW*PNS*4 bytes of LDS: 256 / 34 KiB at 512-point rows,
// 512 / 69 KiB at 1024, 512 / 68 KiB at 2048 and 1024 / 136 KiB at 4096 (RW = 8 there, see rows32_rows).
// (k_tile covers these lengths with 16 points per thread and two full-complex exchanges: 512-point rows were the slow
// pass of the 2^19 plan, and 2048-point rows did not exist: 2^21 needed three passes.)
// ---------------------------------------------------------------------------
template <int LGN, int RW = 16>
struct Rows32 {
    static constexpr int N = 1 << LGN, T = N / 32, WG = RW * T;
    static constexpr int PN = N + N / 32;
    static constexpr int PNS = PN + ((17 - PN % 32) + 32) % 32;  // padded floats per row, = 17 mod 32
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
    const uint32_t b0 = blockIdx.x;
    const uint32_t bid = (xcd_swizzle & 1u) ? (b0 & 7u) * (gridDim.x >> 3) + (b0 >> 3) : b0;
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

    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP(0);
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_DEFAULT>(rin, voff, m * mstep); });
    FWA_STAMP(1);
    fft_reg<32, DIR>(x);
    twiddle_outputs<32, N, DIR>(x, tw, t);
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
            twiddle_outputs<R1, N, DIR>(z, tw, sJ);
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
    FWA_STAMP(3);
}

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
    const uint32_t b0 = blockIdx.x;
    const uint32_t bid = (xcd_swizzle & 1u) ? (b0 & 7u) * (gridDim.x >> 3) + (b0 >> 3) : b0;
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
    twiddle_outputs<32, N, DIR>(x, tw, kk);
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
    const uint32_t b0 = blockIdx.x;
    const uint32_t bid = (xcd_swizzle & 1u) ? (b0 & 7u) * (gridDim.x >> 3) + (b0 >> 3) : b0;
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
    FWA_ENTRY_HOOK();
    FWA_STAMP(0);
    static_for<0, 32>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = buf_load<AUX_NT>(rin, voff, soff + (m * T) * rstep); });
    FWA_STAMP(1);
    const uint32_t col = tile * CW + c;
    auto look = [&](uint32_t e) { return cmul(tw_hi[e >> 10], tw_lo[e & 1023]); };  // W_n^e, e < n
    tq[kk * CW + c] = look(col * (32 * kk));           // R1 == T rows: one per thread
    if (kk < B1) tb[kk * CW + c] = look(col * (kk * T));
    const v2f A = look(col * kk);

    fft_reg<32, DIR>(x);
    twiddle_outputs<32, N, DIR>(x, tw, kk);
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

bool rows32_supported(uint32_t lg_l) { return lg_l >= 9 && lg_l <= 12; }
// tile-contiguous ring input of width in_cw (written by k_colsw): rows of >= 32*in_cw points
bool rows32_ring_supported(uint32_t lg_l, uint32_t in_cw) { return (in_cw == 32 || in_cw == 64) && lg_l >= 10 && lg_l <= 12 && (1u << (lg_l - 5)) >= in_cw; }

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
static const void *rows32_pick(uint32_t lg_l, int dir, uint32_t in_cw)
{
    switch (lg_l) {
        case 9: return rows32_kernel_of<9>(dir, 0);
        case 10: return rows32_kernel_of<10>(dir, in_cw);
        case 11: return rows32_kernel_of<11>(dir, in_cw);
        case 12: return rows32_kernel_of<12>(dir, in_cw);
        default: return nullptr;
    }
}
static void rows32_geometry(uint32_t lg_l, uint32_t *rw, uint32_t *threads, int *lds)
{
    *rw = (uint32_t)rows32_rows((int)lg_l);
    *threads = *rw << (lg_l - 5);
    const int n = 1 << lg_l, pn = n + n / 32, pns = pn + ((17 - pn % 32) + 32) % 32;
    *lds = (int)*rw * pns * 4;
}

// > 64 KiB of dynamic LDS needs the attribute once per device (plan setup)
hipError_t prepare_rows32(uint32_t lg_l)
{
    if (!rows32_supported(lg_l)) return hipErrorInvalidValue;
    uint32_t rw, th;
    int lds;
    rows32_geometry(lg_l, &rw, &th, &lds);
    hipError_t e = hipSuccess;
    for (int dir : {FWD, INV})
        for (uint32_t cw : {0u, 32u, 64u})
            if (e == hipSuccess && (cw == 0 || rows32_ring_supported(lg_l, cw)))
                e = hipFuncSetAttribute(rows32_pick(lg_l, dir, cw), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}

// n = n1 * 2^lg_l per transform, n * 8 < 2^32; `in`: n1 rows of 2^lg_l contiguous samples (in_cw = 0) or the
// tile-contiguous ring of k_colsw (in_cw = its tile width); out[k1 + n1*k2]
hipError_t launch_rows32(int dir, uint32_t lg_l, const v2f *in, v2f *out, const v2f *tw, uint32_t n1, uint64_t in_sb,
                         uint64_t out_sb, uint32_t n_transforms, float scale, uint32_t xcd_swizzle, uint32_t in_cw,
                         hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (!rows32_supported(lg_l) || n1 < 16 || (n1 & (n1 - 1)) || ((uint64_t)n1 << lg_l) > (1ull << 28))
        return hipErrorInvalidValue;
    if (in_cw && !rows32_ring_supported(lg_l, in_cw)) return hipErrorInvalidValue;
    uint32_t rw, th;
    int lds;
    rows32_geometry(lg_l, &rw, &th, &lds);
    const uint64_t blocks = (uint64_t)n_transforms * (n1 / rw);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (blocks % 8) xcd_swizzle = 0;
    void *args[] = {&in, &out, &tw, &n1, &in_sb, &out_sb, &scale, &xcd_swizzle};
    return hipLaunchKernel(rows32_pick(lg_l, dir, in_cw), dim3((uint32_t)blocks), dim3(th), args, (size_t)lds, st);
}

}  // namespace fwa
