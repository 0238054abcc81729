// tools/whatif_p1.hip -- what-if timing of the 2^20 pass-1 tile (round 3): the tile body of tile_1m.h copied here with
// compile-time knock-outs (results are WRONG by design) to price the pieces of its compute phase:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/whatif_p1 tools/whatif_p1.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fft_wgpu_amd/csrc/tile_1m.h"

namespace fwa {
// KO bits: 1 = no stage-1 twiddle (LDS table reads + 31 cmul), 2 = no four-step twiddle (33 LDS reads + 64 cmul),
// 4 = no LDS exchange (no ds ops, no barriers), 8 = no first radix-32, 16 = no second radix-32, 32 = twiddle table in
// [q][k1] layout read as b128, 64 = stores only 1 of 4
template <int KO>
__global__ __launch_bounds__(512) void k_p1_whatif(const v2f *__restrict__ src, v2f *__restrict__ ring, const v2f *__restrict__ tw_inner,
                                                   const v2f *__restrict__ tw_outer)
{
    constexpr int W = 16;
    using G = Geom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + G::XCH_BYTES + G::TWI_BYTES + 1024);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(1);
    const uint32_t tile = bid % G::TILES;
    const uint64_t t = bid / G::TILES;
    const v2f *in = src + t * (1ull << 20);
    v2f *out = ring + t * (1ull << 20);
    const v2f *tw_outer_tile = tw_outer + (size_t)tile * (64 * W);
    if constexpr (KO & 32) {
        // [q][k1] rows of 34 v2f (272 B: conflict-free b128 reads for 32 different q as well)
        for (uint32_t i = tid; i < 1024; i += 512) twi[(i & 31) * 34 + (i >> 5)] = tw_inner[i];
    } else {
        reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    }
    const uint32_t c = tid & (W - 1), q = tid >> G::LGW;
    const uint32_t voff = (q * 1024 + c) * 8, soff = tile * (W * 8);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    static_for<0, 32>([&](auto j_) { constexpr int j = decltype(j_)::value; x[j] = buf_load<AUX_NT>(rin, voff, soff + j * 262144); });
    reinterpret_cast<v4f *>(two)[tid] = reinterpret_cast<const v4f *>(tw_outer_tile)[tid];
    __syncthreads();
    if constexpr (!(KO & 8)) fft_reg<32, FWD>(x);
    if constexpr (!(KO & 1)) {
        if constexpr (KO & 32) {
            const v4f *row = reinterpret_cast<const v4f *>(twi + q * 34);
            static_for<0, 16>([&](auto p_) {
                constexpr int p = decltype(p_)::value;
                const v4f w2 = row[p];
                if constexpr (p != 0) x[brev<32>(2 * p)] = cmul_tw<FWD>(x[brev<32>(2 * p)], v2f{w2.x, w2.y});
                x[brev<32>(2 * p + 1)] = cmul_tw<FWD>(x[brev<32>(2 * p + 1)], v2f{w2.z, w2.w});
            });
        } else {
            static_for<1, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; constexpr int r = brev<32>(k1); x[r] = cmul_tw<FWD>(x[r], twi[k1 * 32 + q]); });
        }
    }
    if constexpr (!(KO & 4)) {
        static_for<0, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; xch[c + W * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].x; });
        __syncthreads();
        static_for<0, 32>([&](auto n_) { constexpr int np = decltype(n_)::value; x[np].x = xch[c + W * (q * 32 + (np ^ G::sw1(q)))]; });
        __syncthreads();
        static_for<0, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; xch[c + W * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].y; });
        __syncthreads();
        static_for<0, 32>([&](auto n_) { constexpr int np = decltype(n_)::value; x[np].y = xch[c + W * (q * 32 + (np ^ G::sw1(q)))]; });
    }
    if constexpr (!(KO & 16)) fft_reg<32, FWD>(x);
    const v2f A = two[q * W + c];
    const uint32_t voff_o = (q * W + c) * 8, soff_o = tile * G::TILE_BYTES;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        if constexpr ((KO & 64) && (k2 & 3)) return;
        v2f v = x[brev<32>(k2)];
        if constexpr (!(KO & 2)) v = cmul_tw<FWD>(v, cmul(A, two[32 * W + k2 * W + c]));
        buf_store<AUX_SC1>(v, rout, voff_o, soff_o + k2 * (32 * W * 8));
    });
}
}  // namespace fwa

namespace fwa {
// memory skeleton only: 32 loads per thread, then 32 stores per thread (the values pass through untouched)
// RD: 0 = column tile (128-B segments at 8-KiB pitch), 1 = the workgroup's 128 KiB as one contiguous chunk
// WR: 0 = tile-contiguous ring (as shipped), 1 = same addresses as the reads (in place into `ring` at the read offsets)
// AUXW: store policy
template <int RD, int WR, int AUXW, int AUXR>
__global__ __launch_bounds__(512) void k_skel(const v2f *__restrict__ src, v2f *__restrict__ ring)
{
    constexpr int W = 16;
    using G = Geom<W>;
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(1);
    const uint32_t tile = bid % G::TILES;
    const uint64_t t = bid / G::TILES;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(src + t * (1ull << 20)), rout = make_rsrc(ring + t * (1ull << 20));
    const uint32_t c = tid & (W - 1), q = tid >> G::LGW;
    const uint32_t voff_t = (q * 1024 + c) * 8, soff_t = tile * (W * 8);       // tile pattern: + j * 262144
    const uint32_t voff_l = tid * 8, soff_l = tile * G::TILE_BYTES;             // linear: + j * 4096
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = RD == 0 ? buf_load<AUXR>(rin, voff_t, soff_t + j * 262144) : buf_load<AUXR>(rin, voff_l, soff_l + j * 4096);
    });
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (WR == 0) buf_store<AUXW>(x[j], rout, voff_l, soff_l + j * 4096);
        else buf_store<AUXW>(x[j], rout, voff_t, soff_t + j * 262144);
    });
}
}  // namespace fwa

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using fwa::v2f;
__global__ void k_fill(v2f *d, uint64_t n) { for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) d[i] = fwa::gen_sample(1, i, 1e-3f); }

template <int KO> static float run(const v2f *src, v2f *ring, const v2f *twi, const v2f *two, v2f *flush, int nt)
{
    const int lds = 65536 + 8192 + 1024 + 2 * 32 * 16 * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwa::k_p1_whatif<KO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 7; ++rep) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, flush, (uint64_t)64 << 20);  // 512 MiB written: cold caches
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fwa::k_p1_whatif<KO>, dim3(nt * 64), dim3(512), lds, 0, src, ring, twi, two);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}

template <int RD, int WR, int AUXW, int AUXR> static float run_skel(const v2f *src, v2f *ring, v2f *flush, int nt)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 7; ++rep) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, flush, (uint64_t)64 << 20);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fwa::k_skel<RD, WR, AUXW, AUXR>), dim3(nt * 64), dim3(512), 0, 0, src, ring);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main(int argc, char **argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 16; const uint64_t N = 1 << 20;
    v2f *src, *ring, *twi, *two, *flush;
    CK(hipMalloc(&src, nt * N * 8)); CK(hipMalloc(&ring, nt * N * 8)); CK(hipMalloc(&twi, 1024 * 8)); CK(hipMalloc(&two, 64 * 1024 * 8));
    CK(hipMalloc(&flush, 512ull << 20));
    std::vector<v2f> ones(64 * 1024, v2f{0.6f, 0.8f});
    CK(hipMemcpy(twi, ones.data(), 1024 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(two, ones.data(), 64 * 1024 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, src, nt * N);
    printf("k_p1 what-if, one launch of %d transforms (%d tiles), best of 6, us:\n", nt, nt * 64);
    printf("  full kernel                                   %7.2f\n", run<0>(src, ring, twi, two, flush, nt));
    printf("  twiddle table [q][k1], b128 reads             %7.2f\n", run<32>(src, ring, twi, two, flush, nt));
    printf("  no stage-1 twiddle (LDS reads + 31 cmul)      %7.2f\n", run<1>(src, ring, twi, two, flush, nt));
    printf("  no four-step twiddle (LDS reads + 64 cmul)    %7.2f\n", run<2>(src, ring, twi, two, flush, nt));
    printf("  no LDS exchange (no ds ops / barriers)        %7.2f\n", run<4>(src, ring, twi, two, flush, nt));
    printf("  no first radix-32                             %7.2f\n", run<8>(src, ring, twi, two, flush, nt));
    printf("  no second radix-32                            %7.2f\n", run<16>(src, ring, twi, two, flush, nt));
    printf("  no radix-32s, no twiddles (exchange only)     %7.2f\n", run<8 | 16 | 1 | 2>(src, ring, twi, two, flush, nt));
    printf("  loads + stores only                           %7.2f\n", run<8 | 16 | 1 | 2 | 4>(src, ring, twi, two, flush, nt));
    printf("  full compute, a quarter of the stores         %7.2f\n", run<64>(src, ring, twi, two, flush, nt));
    printf("memory skeleton (32 loads then 32 stores per thread, nothing else), same launch shape, us (268 MB moved):\n");
    printf("  tile reads nt, tile-contiguous writes sc1 (= shipped)   %7.2f\n", run_skel<0, 0, 16, 2>(src, ring, flush, nt));
    printf("  tile reads nt, tile-contiguous writes plain             %7.2f\n", run_skel<0, 0, 0, 2>(src, ring, flush, nt));
    printf("  tile reads nt, tile-contiguous writes nt                %7.2f\n", run_skel<0, 0, 2, 2>(src, ring, flush, nt));
    printf("  tile reads default, writes sc1                          %7.2f\n", run_skel<0, 0, 16, 0>(src, ring, flush, nt));
    printf("  LINEAR reads nt, tile-contiguous writes sc1             %7.2f\n", run_skel<1, 0, 16, 2>(src, ring, flush, nt));
    printf("  LINEAR reads nt, writes plain                           %7.2f\n", run_skel<1, 0, 0, 2>(src, ring, flush, nt));
    printf("  tile reads nt, TILE (strided) writes sc1                %7.2f\n", run_skel<0, 1, 16, 2>(src, ring, flush, nt));
    printf("  linear reads nt, TILE (strided) writes nt               %7.2f\n", run_skel<1, 1, 2, 2>(src, ring, flush, nt));
    return 0;
}
