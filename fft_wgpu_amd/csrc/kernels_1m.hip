// kernels_1m.hip -- (3/3) the n = 2^20 two-pass pipeline (headline config C3).
//
// Replaces the 20 global-memory radix-2 passes of reference src/kernel/fft4.wgsl:36-101 by two passes of
// register-resident 32 x 32 FFTs: one HBM round trip plus one round trip through a cache-sized ring.
#include "tile_1m.h"

namespace fwa {

// ---------------------------------------------------------------------------
// k_p1_gen: the same 1024-point column pass for any n = 1024 * P (P = 2^4 .. 2^20 columns): pass A of the tiled
// plans whose first factor is 1024 (2^18, 2^19 and re-factorised larger sizes).  Differences to p1_tile: the row
// pitch P is a run-time value (scalar offsets), the output keeps the matrix layout (the next pass is k_tile, which
// reads rows at pitch P), and the per-tile four-step factors A[q][c] = W_n^{col*q}, B[k2][c] = W_n^{32*col*k2} are
// computed by the workgroup itself from the two-level table of domain n (one look-up pair per thread each) instead
// of a precomputed per-tile table.  Same 80 KiB of LDS, two workgroups per CU (k_tile at L = 1024 needs 138 KiB and
// two full-complex exchanges).
// ---------------------------------------------------------------------------
template <int DIR, int AUX_OUT>
__global__ __launch_bounds__(512) void k_p1_gen(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                const v2f *__restrict__ tw_inner, const v2f *__restrict__ tw_lo,
                                                const v2f *__restrict__ tw_hi, uint32_t pitch, uint64_t in_sb,
                                                uint64_t out_sb, uint32_t xcd_swizzle)
{
    using G = Geom<16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + G::XCH_BYTES + G::TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_map(xcd_swizzle);
    const uint32_t tiles = pitch >> 4;
    const uint32_t tile = bid % tiles;
    const uint64_t t = bid / tiles;
    const uint32_t c = tid & 15, q = tid >> 4;
    // rows q + 32*j of the matrix: one buffer descriptor per eight j (a quarter of the transform, <= 2 GiB at
    // n = 2^30), so that every byte offset stays below 2^32
    const uint64_t quarter = (uint64_t)pitch * 256;  // elements
    const uint32_t qbytes = pitch * 2048u;
    const v2f *sbase = src + t * in_sb;
    const uint32_t voff = (q * pitch + c) * 8;
    const uint32_t soff = tile * 128;
    const uint32_t jstep = pitch * 256;  // bytes between rows q + 32*j and q + 32*(j+1)
    v2f x[32];
    static_for<0, 4>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(sbase + g * quarter), 0, qbytes, 0x00020000);
        static_for<0, 8>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            x[8 * g + j] = buf_load<AUX_NT>(rin, voff, soff + j * jstep);
        });
    });
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    {   // A[q][c] = W_n^{col*q} (thread (q, c)), B[k2][c] = W_n^{32*col*k2} (thread (k2, c)): exponents < n <= 2^30
        const uint32_t col = tile * 16 + c;
        const uint32_t ea = col * q, eb = ea << 5;
        two[q * 16 + c] = cmul(tw_hi[ea >> 10], tw_lo[ea & 1023]);
        two[512 + q * 16 + c] = cmul(tw_hi[eb >> 10], tw_lo[eb & 1023]);
    }
    __syncthreads();

    stage1_fft_twiddle<DIR>(x, twi, q);
    static_for<0, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; xch[c + 16 * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].x; });
    __syncthreads();
    static_for<0, 32>([&](auto n_) { constexpr int np = decltype(n_)::value; x[np].x = xch[c + 16 * (q * 32 + (np ^ G::sw1(q)))]; });
    __syncthreads();
    static_for<0, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; xch[c + 16 * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].y; });
    __syncthreads();
    static_for<0, 32>([&](auto n_) { constexpr int np = decltype(n_)::value; x[np].y = xch[c + 16 * (q * 32 + (np ^ G::sw1(q)))]; });
    fft_reg<32, DIR>(x);  // x[brev(k2)] = output K1 = q + 32*k2

    const v2f A = two[q * 16 + c];
    v2f *dbase = dst + t * out_sb;
    static_for<0, 4>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dbase + g * quarter, 0, qbytes, 0x00020000);
        static_for<0, 8>([&](auto j_) {
            constexpr int k2 = 8 * g + decltype(j_)::value;
            const v2f w = cmul(A, two[512 + k2 * 16 + c]);
            buf_store<AUX_OUT>(cmul_tw<DIR>(x[brev<32>(k2)], w), rout, voff, soff + (k2 & 7) * jstep);
        });
    });
}

hipError_t launch_p1_gen(int dir, bool out_is_ring, const v2f *src, v2f *dst, const v2f *tw_inner, const v2f *tw_lo,
                         const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb, uint32_t n_transforms,
                         uint32_t xcd_swizzle, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (pitch < 16 || pitch > (1u << 20) || (pitch & (pitch - 1))) return hipErrorInvalidValue;
    const uint64_t blocks = (uint64_t)n_transforms * (pitch / 16);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    using G = Geom<16>;
    if (blocks % 8) xcd_swizzle = 0;  // the XCD mapping needs a grid that is a multiple of 8
    void *args[] = {&src, &dst, &tw_inner, &tw_lo, &tw_hi, &pitch, &in_sb, &out_sb, &xcd_swizzle};
    const void *k = dir == FWD ? (out_is_ring ? reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_SC1>) : reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_NT>))
                               : (out_is_ring ? reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_SC1>) : reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_NT>));
    return hipLaunchKernel(k, dim3((uint32_t)blocks), dim3(512), args, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES, st);
}

hipError_t setup_1m_kernels()
{
    hipError_t e = setup_w<16>();
    using G = Geom<16>;
    for (const void *kg : {reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_SC1>), reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_NT>),
                           reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_SC1>), reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_NT>)})
        if (e == hipSuccess)
            e = hipFuncSetAttribute(kg, hipFuncAttributeMaxDynamicSharedMemorySize, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES);
    return e;
}

// 1024 x 16-column tiles, 512 threads, two workgroups per CU (the 32-column tile of one 1024-thread workgroup per CU measured
// 23.5 ms against 21.3 at C3 and left the tree in round 6: profiles/round6/lab_pruned_families.patch)
hipError_t launch_p1_1m(int dir, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t n_transforms, uint32_t swz, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    return dir == FWD ? launch_p1_w<FWD, 16>(src, ring, tw_inner, tw_outer, n_transforms, swz, st)
                      : launch_p1_w<INV, 16>(src, ring, tw_inner, tw_outer, n_transforms, swz, st);
}

hipError_t launch_p2_1m(int dir, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t n_transforms,
                        float scale, uint32_t swz, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    return dir == FWD ? launch_p2_w<FWD, 16>(ring, dst, tw_inner, n_transforms, scale, swz, st)
                      : launch_p2_w<INV, 16>(ring, dst, tw_inner, n_transforms, scale, swz, st);
}

}  // namespace fwa
