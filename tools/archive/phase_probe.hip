// tools/phase_probe.hip -- diagnostic build of the two-pass kernels with in-kernel time stamps (round 3).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/phase_probe tools/phase_probe.hip
// Question: where does a pass-A / pass-C workgroup spend its ~26 us (load wait, arithmetic + exchange, store drain), are
// the two workgroups of a CU and the CUs of the chip in lockstep (all loading, then all computing, then all storing),
// and does a start stagger (FWA_ENTRY_HOOK: the second workgroup of every CU starts D us late) shorten a launch?
// The library's kernels are compiled here unchanged except for the two hook macros of device_common.h; in the library
// both are empty.  Stamps (s_memrealtime, 100 MHz) go to a buffer of their own, one row per wave.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

__device__ uint64_t *g_stamps;        // [block][wave][8]
__device__ uint32_t g_stagger_ticks;  // 100-MHz ticks
__device__ uint32_t g_stagger_lo, g_stagger_hi;  // blocks in [lo, hi) start late
__device__ uint32_t g_wait_at_1;      // 1: stamp 1 is taken after s_waitcnt vmcnt(0) (all loads of the wave landed)

__device__ __forceinline__ void fwa_stamp(int slot)
{
    if ((threadIdx.x & 63) == 0 && g_stamps) {
        uint64_t *row = g_stamps + ((uint64_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8;
        row[slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) {
            row[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
            row[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
        }
    }
}
#define FWA_STAMP(slot)                                                            \
    do {                                                                           \
        if ((slot) == 1 && g_wait_at_1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        fwa_stamp(slot);                                                           \
        if ((slot) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); fwa_stamp(4); } \
    } while (0)
#define FWA_ENTRY_HOOK()                                                           \
    do {                                                                           \
        if (g_stagger_ticks && blockIdx.x >= g_stagger_lo && blockIdx.x < g_stagger_hi) { \
            const uint64_t t0_ = __builtin_amdgcn_s_memrealtime();                 \
            while (__builtin_amdgcn_s_memrealtime() - t0_ < g_stagger_ticks) __builtin_amdgcn_s_sleep(8); \
        }                                                                          \
    } while (0)

#include "../fft_wgpu_amd/csrc/kernels_rows32.hip"
#include "../fft_wgpu_amd/csrc/kernels_1m.hip"

namespace fwa {
__global__ void k_probe_fill(v2f *dst, uint64_t n, float scale)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = gen_sample(1, i, scale);
}
hipError_t launch_fill(v2f *dst, uint64_t, uint64_t, uint64_t n, float scale, hipStream_t st)
{
    hipLaunchKernelGGL(k_probe_fill, dim3(4096), dim3(256), 0, st, dst, n, scale);
    return hipGetLastError();
}
}  // namespace fwa

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using fwa::v2f;

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

int main(int argc, char **argv)
{
    const int nt = 16;  // transforms per launch (one group)
    const uint64_t N = 1ull << 20;
    v2f *src, *ring, *dst, *tw, *tw_lo, *tw_hi, *tw_inner, *tw_outer;
    CK(hipMalloc(&src, nt * N * 8)); CK(hipMalloc(&ring, nt * N * 8)); CK(hipMalloc(&dst, nt * N * 8));
    CK(hipMalloc(&tw, 4096 * 8)); CK(hipMalloc(&tw_lo, 1024 * 8)); CK(hipMalloc(&tw_hi, 1024 * 8 * 1024));
    CK(hipMalloc(&tw_inner, 1024 * 8)); CK(hipMalloc(&tw_outer, 64 * 1024 * 8));
    // twiddle VALUES do not matter for timing; unit-modulus constants keep the data finite
    std::vector<v2f> ones(1024 * 1024, v2f{0.6f, 0.8f});
    CK(hipMemcpy(tw, ones.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw_lo, ones.data(), 1024 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw_hi, ones.data(), 1024 * 1024 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw_inner, ones.data(), 1024 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw_outer, ones.data(), 64 * 1024 * 8, hipMemcpyHostToDevice));
    CK(fwa::launch_fill(src, 1, 0, nt * N, 1e-3f, 0));
    CK(fwa::setup_1m_kernels()); CK(fwa::prepare_colsw(9)); CK(fwa::prepare_colsw(8));
    for (uint32_t l = 9; l <= 12; ++l) CK(fwa::prepare_rows32(l));
    uint64_t *stamps;
    const size_t stamp_bytes = (size_t)2048 * 16 * 8 * 8;
    CK(hipMalloc(&stamps, stamp_bytes));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    struct K { const char *name; int id; uint32_t blocks, waves; };
    K ks[] = {{"k_p1_1m<16>      (1024 x 16 cols, HBM nt -> ring sc1)", 0, 1024, 8},
              {"k_colsw<9,32>    ( 512 x 32 cols)", 1, 1024, 8},
              {"k_colsw<8,64>    ( 256 x 64 cols)", 2, 1024, 8},
              {"k_p2_1m<16>      (ring -> HBM nt, 16 rows of 1024)", 3, 1024, 8},
              {"k_rows32<10,16>  (16 rows of 1024, 512 x 1024)", 4, 512, 8},
              {"k_rows32<11,8>   (8 rows of 2048, 512 x 2048)", 5, 1024, 8}};
    auto launch = [&](int id) {
        switch (id) {
            case 0: return fwa::launch_p1_1m(fwa::FWD, 16, src, ring, tw_inner, tw_outer, nt, 1, 0);
            case 1: return fwa::launch_colsw(fwa::FWD, 9, true, true, src, ring, tw, tw_lo, tw_hi, 2048, N, N, nt, 0, 0);
            case 2: return fwa::launch_colsw(fwa::FWD, 8, true, true, src, ring, tw, tw_lo, tw_hi, 4096, N, N, nt, 0, 0);
            case 3: return fwa::launch_p2_1m(fwa::FWD, 16, ring, dst, tw_inner, nt, 1.0f, 1, 0);
            case 4: return fwa::launch_rows32(fwa::FWD, 10, ring, dst, tw, 512, N / 2, N / 2, 2 * nt, 1.0f, 0, 0, 0);
            default: return fwa::launch_rows32(fwa::FWD, 11, ring, dst, tw, 512, N, N, nt, 1.0f, 0, 0, 0);
        }
    };
    const uint32_t staggers_us[] = {0, 0, 3, 6, 9, 12};
    for (auto &k : ks) {
        printf("== %s\n", k.name);
        for (int si = 0; si < 6; ++si) {
            const uint32_t wait1 = si == 0 ? 0u : 1u;  // first row: stamp 1 without the forced wait (least perturbed total)
            const uint32_t ticks = staggers_us[si] * 100, lo = 256, hi = 512;  // the second workgroup of every CU
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wait_at_1), &wait1, 4));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stagger_ticks), &ticks, 4));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stagger_lo), &lo, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stagger_hi), &hi, 4));
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(fwa::launch_fill(src, 1, 0, nt * N, 1e-3f, 0));  // evicts the caches, as the pipeline's other traffic does
                CK(hipMemsetAsync(stamps, 0, stamp_bytes, 0));
                CK(hipEventRecord(e0));
                CK(launch(k.id));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            std::vector<uint64_t> h(stamp_bytes / 8);
            CK(hipMemcpy(h.data(), stamps, stamp_bytes, hipMemcpyDeviceToHost));
            uint64_t tmin = ~0ull, tmax = 0;
            std::vector<double> d_load, d_comp, d_issue, d_drain, d_life, start;
            for (uint32_t b = 0; b < k.blocks; ++b)
                for (uint32_t w = 0; w < k.waves; ++w) {
                    const uint64_t *r = &h[((size_t)b * 16 + w) * 8];
                    if (!r[0]) continue;
                    tmin = std::min(tmin, r[0]); tmax = std::max(tmax, r[4]);
                }
            for (uint32_t b = 0; b < k.blocks; ++b) {
                const uint64_t *r = &h[((size_t)b * 16) * 8];  // wave 0
                if (!r[0]) continue;
                d_load.push_back((r[1] - r[0]) * 0.01); d_comp.push_back((r[3] - r[1]) * 0.01);
                d_drain.push_back((r[4] - r[3]) * 0.01); d_life.push_back((r[4] - r[0]) * 0.01);
                start.push_back((r[0] - tmin) * 0.01);
            }
            std::vector<double> s2 = start; std::sort(s2.begin(), s2.end());
            printf("stagger %2u us wait@1 %u: event %.2f us, first start -> last end %.2f us | wave 0 medians: load %.2f, fft+stores issued %.2f, drain %.2f, life %.2f us | starts: p25 %.1f p50 %.1f p75 %.1f max %.1f us\n",
                   staggers_us[si], wait1, best * 1e3, (tmax - tmin) * 0.01, median(d_load), median(d_comp), median(d_drain), median(d_life),
                   s2[s2.size() / 4], s2[s2.size() / 2], s2[3 * s2.size() / 4], s2.back());
            if (si == 1) {  // co-residency: how many distinct (xcc, se, cu) and what the first 4 blocks of one CU look like
                std::vector<uint64_t> cu;
                for (uint32_t b = 0; b < k.blocks; ++b) {
                    const uint64_t *r = &h[((size_t)b * 16) * 8];
                    cu.push_back(((r[7] & 15) << 16) | (r[6] & 0xff00));
                }
                std::vector<uint64_t> u = cu; std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
                printf("   distinct (xcc, se/sh/cu) ids: %zu; blocks sharing block 0's CU:", u.size());
                for (uint32_t b = 0; b < k.blocks; ++b) if (cu[b] == cu[0]) printf(" %u(start %.1f)", b, (h[((size_t)b * 16) * 8] - tmin) * 0.01);
                printf("\n");
            }
        }
    }
    return 0;
}
