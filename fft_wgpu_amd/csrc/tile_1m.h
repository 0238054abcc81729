// tile_1m.h -- the 1024 x 1024 tile bodies of the 2^20 two-pass pipeline (kernels_1m.hip; the laboratory build
// instantiates them a second time in kernels_lab_1m.hip: 32-column tiles and the persistent ring kernel).
#pragma once
#include "device_common.h"

// Cache policy of the ring accesses (measured: stores sc1 = write-through, loads default); overridable at compile time
// for the policy A/B of tools/archive/run_r3_policies.sh only.
#ifndef FWA_RING_ST_AUX
#define FWA_RING_ST_AUX AUX_SC1
#endif
#ifndef FWA_RING_LD_AUX
#define FWA_RING_LD_AUX AUX_DEFAULT
#endif
#ifndef FWA_USER_LD_AUX
#define FWA_USER_LD_AUX AUX_NT
#endif
#ifndef FWA_USER_ST_AUX
#define FWA_USER_ST_AUX AUX_NT
#endif

namespace fwa {

// ---------------------------------------------------------------------------
// n = 2^20 = 1024 x 1024, two passes.
//
// Index algebra (n = 1024*n1 + n2, k = K1 + 1024*K2):
//   X[K1 + 1024 K2] = sum_{n2} W_N^{n2 K1} * ( sum_{n1} x[1024 n1 + n2] W_1024^{n1 K1} ) * W_1024^{n2 K2}
// pass 1: tile = W adjacent columns n2 (one W*8-byte segment per matrix row); 1024-point FFT over n1 per
//         column; multiply by W_N^{n2 K1}; store Y[K1][n2] into the scratch ring, tile-contiguous.
// pass 2: tile = W adjacent rows K1; 1024-point FFT over n2 per row; store X[K1 + 1024 K2]
//         (W adjacent K1 = one W*8-byte segment per K2).
// Each 1024-point FFT = radix-32 (registers) -> twiddle W_1024^{n' k1} -> LDS exchange -> radix-32.
// A workgroup has 32*W threads with 32 points each (64 data VGPRs); the exchange buffer holds the real
// parts, then the imaginary parts (W*4 KiB).
//   W = 16: 512 threads, 80 KiB LDS, two workgroups per CU, 128-B HBM segments.
//   W = 32: 1024 threads, 152 KiB LDS, one workgroup per CU, 256-B HBM segments (the column-tile stream
//           sustains more with 256-B segments: profiles/round1/probe_tile_pitch_width.txt).
// Cache policy (measured, profiles/round1/probe_fabric_cache_policies.txt): user-buffer accesses `nt`,
// ring stores `sc1` (write-through), ring loads default.
// ---------------------------------------------------------------------------
template <int W>
struct Geom {
    static_assert(W == 16 || W == 32, "tile width");
    static constexpr int LGW = (W == 16) ? 4 : 5;
    static constexpr int THREADS = 32 * W;
    static constexpr int TILES = 1024 / W;
    static constexpr int XCH_BYTES = W * 4096;        // one float per point of the tile
    static constexpr int TWI_BYTES = 8192;            // [k1][n'] = W_1024^{n' k1}
    static constexpr int TWO_BYTES = 2 * 32 * W * 8;  // per tile A[32][W], B[32][W]
    static constexpr uint32_t TILE_BYTES = W * 8192;  // one tile of the ring slab
    // XOR swizzles that make both sides of the exchange conflict-free (bank = word address mod 32)
    static __device__ __forceinline__ constexpr uint32_t sw1(uint32_t k1) { return W == 16 ? (k1 & 1) : 0; }
    static __device__ __forceinline__ uint32_t sw2(uint32_t r, uint32_t k1)
    {
        return W == 16 ? ((r + 16 * (k1 & 1)) & 31) : r;
    }
};

template <int DIR>
__device__ __forceinline__ void stage1_fft_twiddle(v2f (&x)[32], const v2f *twi, uint32_t q)
{
    fft_reg<32, DIR>(x);
    // x[brev(k1)] = Z[k1]; multiply by W_1024^{q*k1}; table layout [k1][q]
    static_for<1, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        constexpr int r = brev<32>(k1);
        x[r] = cmul_tw<DIR>(x[r], twi[k1 * 32 + q]);
    });
}

// One pass-1 tile: column FFTs.  `in` is the (wave-uniform) base of a 1024x1024 row-major transform, `out`
// the base of its ring slab: tile s owns bytes [s*W*8 KiB, +W*8 KiB) as [K1 (1024)][column (W)], so every
// store instruction of a wave covers 512 contiguous bytes and a pass-2 tile finds its W rows of a source
// tile as ONE contiguous W*W*8-byte chunk.
template <int DIR, int W>
__device__ __forceinline__ void p1_tile(const v2f *in, v2f *out, uint32_t tile, const v2f *tw_outer_tile,
                                        float *xch, const v2f *twi, v2f *two, uint32_t tid)
{
    using G = Geom<W>;
    const uint32_t c = tid & (W - 1);  // column inside the tile
    const uint32_t q = tid >> G::LGW;  // n' before the exchange, k1 after it
    const uint32_t voff = (q * 1024 + c) * 8;
    const uint32_t soff = tile * (W * 8);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP(0);
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<FWA_USER_LD_AUX>(rin, voff, soff + j * 262144);
    });
    FWA_STAMP(1);
    reinterpret_cast<v4f *>(two)[tid] = reinterpret_cast<const v4f *>(tw_outer_tile)[tid];
    __syncthreads();

    stage1_fft_twiddle<DIR>(x, twi, q);

    // exchange: word address c + W*(k1*32 + (n' ^ sw1(k1)))
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + W * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].x = xch[c + W * (q * 32 + (np ^ G::sw1(q)))];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + W * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].y = xch[c + W * (q * 32 + (np ^ G::sw1(q)))];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = FFT1024 output K1 = q + 32*k2

    FWA_STAMP(2);
    // four-step twiddle W_N^{n2*K1} = A[q][c] * B[k2][c]
    const v2f A = two[q * W + c];
    const uint32_t voff_o = (q * W + c) * 8;
    const uint32_t soff_o = tile * G::TILE_BYTES;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        const v2f w = cmul(A, two[32 * W + k2 * W + c]);
        buf_store<FWA_RING_ST_AUX>(cmul_tw<DIR>(x[brev<32>(k2)], w), rout, voff_o, soff_o + k2 * (32 * W * 8));
    });
    FWA_STAMP(3);
}

// One pass-2 tile: row FFTs + transposed store.  `in` = ring slab of the transform, `out` = its 1024x1024
// result matrix; the tile reads rows [W*tile, W*tile+W) and writes columns [W*tile, W*tile+W).
// AUX_IN: cache policy of the ring loads; after_load() runs once every load of the calling thread has been issued
// and must contain a workgroup barrier (it also makes the twiddle table visible).
template <int DIR, int W, int AUX_IN, class AfterLoad>
__device__ __forceinline__ void p2_tile(const v2f *in, v2f *out, uint32_t tile, float scale, float *xch,
                                        const v2f *twi, uint32_t tid, AfterLoad after_load)
{
    using G = Geom<W>;
    // before the exchange: lane = n' (32 consecutive samples of one row), r = row in the tile.
    // Sample n2 = 32*j + n' lives in source tile n2 / W, whose rows [W*tile, +W) are one chunk [row][W columns].
    const uint32_t np = tid & 31;
    const uint32_t r = tid >> 5;
    const uint32_t voff_in = (np >> G::LGW) * G::TILE_BYTES + r * (W * 8) + (np & (W - 1)) * 8;
    const uint32_t soff_in = tile * (W * W * 8);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP_B(0);
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_IN>(rin, voff_in, soff_in + j * 262144);
    });
    FWA_STAMP_B(1);
    after_load();

    stage1_fft_twiddle<DIR>(x, twi, np);

    // after the exchange: lane = r' (W adjacent K1 = one output segment), k1' = tid / W
    const uint32_t r2 = tid & (W - 1);
    const uint32_t k1p = tid >> G::LGW;
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ G::sw2(r, k1))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    const uint32_t rd_base = (r2 * 32 + k1p) * 32;
    const uint32_t rd_xor = G::sw2(r2, k1p);
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].x = xch[rd_base + (n ^ rd_xor)];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ G::sw2(r, k1))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].y = xch[rd_base + (n ^ rd_xor)];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = row FFT output K2 = k1p + 32*k2

    FWA_STAMP_B(2);
    const uint32_t voff_out = (k1p * 1024 + r2) * 8;
    const uint32_t soff_out = tile * (W * 8);
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        buf_store<FWA_USER_ST_AUX>(x[brev<32>(k2)] * scale, rout, voff_out, soff_out + k2 * 262144);
    });
    FWA_STAMP_B(3);
}

// XCD-aware block -> tile mapping: xcd_map (device_common.h).  Bit 0 hands each XCD a contiguous run of (transform, tile)
// indices: its 64 resident workgroups are 64 consecutive tiles and cover whole 8-KiB rows (+ 2 %); bit 2 puts the two residents
// of a CU on adjacent tiles (+ 4-8 % for a launch that runs alone, + 0.6 % with two chains: sweep_pair_map_two_chains.jsonl).
__device__ __forceinline__ uint32_t xcd_block(uint32_t swizzle, bool pair = false)
{
    return xcd_map((swizzle & 1u) | (pair ? 4u : 0u));
}

template <int DIR, int W>
__global__ __launch_bounds__(32 * W) void k_p1_1m(const v2f *__restrict__ src, v2f *__restrict__ ring,
                                                  const v2f *__restrict__ tw_inner,
                                                  const v2f *__restrict__ tw_outer, uint32_t xcd_swizzle)
{
    using G = Geom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + G::XCH_BYTES + G::TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(xcd_swizzle & 1u, (xcd_swizzle & 4u) != 0);
    const uint32_t tile = bid % G::TILES;
    const uint64_t t = bid / G::TILES;  // transform inside the group = ring slot
    if (tid < 512) reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p1_tile<DIR, W>(src + t * (1ull << 20), ring + t * (1ull << 20), tile, tw_outer + (size_t)tile * (64 * W), xch,
                    twi, two, tid);
}

template <int DIR, int W>
__global__ __launch_bounds__(32 * W) void k_p2_1m(const v2f *__restrict__ ring, v2f *__restrict__ dst,
                                                  const v2f *__restrict__ tw_inner, float scale, uint32_t xcd_swizzle)
{
    using G = Geom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(xcd_swizzle & 1u, (xcd_swizzle & 4u) != 0);
    const uint32_t tile = bid % G::TILES;
    // bit 1: newest ring slots first (the transforms pass 1 wrote last are the likeliest to still sit in the
    // Infinity Cache when this launch starts)
    const uint64_t t = (xcd_swizzle & 2u) ? (gridDim.x / G::TILES - 1) - bid / G::TILES : bid / G::TILES;
    if (tid < 512) reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p2_tile<DIR, W, FWA_RING_LD_AUX>(ring + t * (1ull << 20), dst + t * (1ull << 20), tile, scale, xch, twi, tid,
                                 [] { __syncthreads(); });
}

template <int W>
static hipError_t setup_w()
{
    using G = Geom<W>;
    const int p1 = G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES, p2 = G::XCH_BYTES + G::TWI_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p1_1m<FWD, W>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, p1);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p1_1m<INV, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, p1);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p2_1m<FWD, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, p2);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p2_1m<INV, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, p2);
    return e;
}

template <int DIR, int W>
static hipError_t launch_p1_w(const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer, uint32_t n_transforms,
                              uint32_t swz, hipStream_t st)
{
    using G = Geom<W>;
    void *args[] = {&src, &ring, &tw_inner, &tw_outer, &swz};
    return hipLaunchKernel(reinterpret_cast<const void *>(&k_p1_1m<DIR, W>), dim3(n_transforms * G::TILES),
                           dim3(G::THREADS), args, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES, st);
}
template <int DIR, int W>
static hipError_t launch_p2_w(const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t n_transforms, float scale,
                              uint32_t swz, hipStream_t st)
{
    using G = Geom<W>;
    void *args[] = {&ring, &dst, &tw_inner, &scale, &swz};
    return hipLaunchKernel(reinterpret_cast<const void *>(&k_p2_1m<DIR, W>), dim3(n_transforms * G::TILES),
                           dim3(G::THREADS), args, G::XCH_BYTES + G::TWI_BYTES, st);
}

}  // namespace fwa
