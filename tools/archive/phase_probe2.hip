// tools/phase_probe2.hip -- the two-chain 2^20 pipeline exactly as fwa_plan_exec issues it (k_p1_1m then k_p2_1m per group of
// 16 transforms, groups alternating over two streams), compiled with in-kernel time stamps: where does a workgroup spend
// its life UNDER LOAD (other chain running beside it), and how many workgroups are resident / loading / computing at a time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/phase_probe2 tools/phase_probe2.hip
// Wave 0 of every workgroup stamps into the row of its (kind, transform, tile): no atomics, one store per stamp.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ uint64_t *g_rows;      // [kind - 1][transform][tile][8]
__device__ uint64_t g_base;       // address of the base of the user buffer: a tile's row follows from its transform pointer, no atomics
__device__ uint32_t g_transforms;

// wave 0 only; `ptr_` = the tile's transform in the USER buffer (p1: `in`, p2: `out`), `tile` = the tile index (both are
// locals of p1_tile / p2_tile)
#define FWA_STAMP_IMPL(kind, slot, ptr_)                                                                  \
    do {                                                                                                  \
        if ((slot) == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 \
        if (threadIdx.x == 0) {                                                                           \
            const uint64_t t_ = ((uint64_t)(ptr_) - g_base) >> 23;             \
            uint64_t *r_ = g_rows + ((((uint64_t)((kind) - 1) * g_transforms + t_) * 64 + tile) * 8);     \
            r_[3 + (slot)] = __builtin_amdgcn_s_memrealtime();                                            \
            if ((slot) == 0) r_[0] = (kind);                                                              \
            if ((slot) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); r_[7] = __builtin_amdgcn_s_memrealtime(); } \
        }                                                                                                 \
    } while (0)
#define FWA_STAMP(slot) FWA_STAMP_IMPL(1, slot, in)
#define FWA_STAMP_B(slot) FWA_STAMP_IMPL(2, slot, out)

#include "../fft_wgpu_amd/csrc/kernels_1m.hip"

namespace fwa {
__global__ void k_probe_fill(v2f *dst, uint64_t n, float scale)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = gen_sample(1, i, scale);
}
}  // namespace fwa

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using fwa::v2f;
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }
static double pct(std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[(size_t)(p * (v.size() - 1))]; }

int main(int argc, char **argv)
{
    const int groups = argc > 1 ? atoi(argv[1]) : 24;   // groups of 16 transforms (12 per chain)
    const int chains = argc > 2 ? atoi(argv[2]) : 2;
    const uint64_t N = 1ull << 20, G = 16;
    v2f *src, *ring, *tw_inner, *tw_outer;
    CK(hipMalloc(&src, groups * G * N * 8));
    CK(hipMalloc(&ring, chains * G * N * 8));
    CK(hipMalloc(&tw_inner, 1024 * 8)); CK(hipMalloc(&tw_outer, 64 * 1024 * 8));
    std::vector<v2f> ones(64 * 1024, v2f{0.6f, 0.8f});
    CK(hipMemcpy(tw_inner, ones.data(), 1024 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw_outer, ones.data(), 64 * 1024 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fwa::k_probe_fill, dim3(4096), dim3(256), 0, 0, src, groups * G * N, 1e-6f);
    CK(fwa::setup_1m_kernels());
    const uint32_t n_tr = (uint32_t)(groups * G), max_rows = 2 * n_tr * 64;
    uint64_t *rows;
    CK(hipMalloc(&rows, (size_t)max_rows * 64));
    CK(hipMemset(rows, 0, (size_t)max_rows * 64));
    const uint64_t base = (uint64_t)src;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows), &rows, sizeof(rows)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_base), &base, sizeof(base)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_transforms), &n_tr, 4));
    std::vector<hipStream_t> st(chains);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventRecord(fork, 0));
    for (auto &s : st) CK(hipStreamWaitEvent(s, fork, 0));
    for (int g = 0; g < groups; ++g) {
        const int c = g % chains;
        v2f *slab = ring + (uint64_t)c * G * N;
        CK(fwa::launch_p1_1m(fwa::FWD, 16, src + (uint64_t)g * G * N, slab, tw_inner, tw_outer, (uint32_t)G, 1, st[c]));
        CK(fwa::launch_p2_1m(fwa::FWD, 16, slab, src + (uint64_t)g * G * N, tw_inner, (uint32_t)G, 1.0f, 1, st[c]));
    }
    for (auto &s : st) { hipEvent_t d; CK(hipEventCreateWithFlags(&d, hipEventDisableTiming)); CK(hipEventRecord(d, s)); CK(hipStreamWaitEvent(0, d, 0)); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d groups of 16 transforms on %d chain(s): %.3f ms = %.2f us per transform (C3 = %.2f ms)\n", groups, chains, ms, ms * 1e3 / (groups * G),
           ms / (groups * G) * 4096);
    const uint32_t used = max_rows;
    std::vector<uint64_t> h((size_t)used * 8);
    CK(hipMemcpy(h.data(), rows, (size_t)used * 64, hipMemcpyDeviceToHost));
    uint64_t tmin = ~0ull, tmax = 0;
    for (uint32_t r = 0; r < used; ++r) { const uint64_t *q = &h[(size_t)r * 8]; if (q[3] && q[3] < tmin) tmin = q[3]; if (q[7] > tmax) tmax = q[7]; }
    // steady state: the middle half of the run
    const uint64_t lo = tmin + (tmax - tmin) / 4, hi = tmin + 3 * (tmax - tmin) / 4;
    for (int kind = 1; kind <= 2; ++kind) {
        std::vector<double> load, comp, drain, life, fft, sti;
        for (uint32_t r = 0; r < used; ++r) {
            const uint64_t *q = &h[(size_t)r * 8];
            if ((int)q[0] != kind || q[3] < lo || q[7] > hi || !q[7]) continue;
            load.push_back((q[4] - q[3]) * 0.01); comp.push_back((q[6] - q[4]) * 0.01); drain.push_back((q[7] - q[6]) * 0.01);
            fft.push_back((q[5] - q[4]) * 0.01); sti.push_back((q[6] - q[5]) * 0.01);
            life.push_back((q[7] - q[3]) * 0.01);
        }
        printf("%s wave 0 of %zu workgroups in the steady half: load wait med %.2f (p10 %.2f p90 %.2f), fft+exchange+fft med %.2f (p10 %.2f p90 %.2f), twiddle+store issue med %.2f (p10 %.2f p90 %.2f), drain med %.2f, life med %.2f (p10 %.2f p90 %.2f) us\n",
               kind == 1 ? "k_p1_1m" : "k_p2_1m", life.size(), med(load), pct(load, .1), pct(load, .9), med(fft), pct(fft, .1), pct(fft, .9), med(sti), pct(sti, .1), pct(sti, .9),
               med(drain), med(life), pct(life, .1), pct(life, .9));
    }
    // occupancy over time in the steady half: resident workgroups, and how many of them are waiting for their loads / computing / draining
    const int bins = 40;
    std::vector<double> res(bins, 0), ld(bins, 0), cp(bins, 0), dr(bins, 0);
    const double bw = (double)(hi - lo) / bins;
    for (uint32_t r = 0; r < used; ++r) {
        const uint64_t *q = &h[(size_t)r * 8];
        if (!q[7]) continue;
        for (int b = 0; b < bins; ++b) {
            const double t = lo + (b + 0.5) * bw;
            if (t >= q[3] && t < q[7]) { res[b] += 1; if (t < q[4]) ld[b] += 1; else if (t < q[6]) cp[b] += 1; else dr[b] += 1; }
        }
    }
    double ar = 0, al = 0, ac = 0, ad = 0;
    for (int b = 0; b < bins; ++b) { ar += res[b]; al += ld[b]; ac += cp[b]; ad += dr[b]; }
    printf("steady half, averages over %d sample times: resident workgroups %.0f of 512 slots; waiting for loads %.0f, computing %.0f, draining stores %.0f\n",
           bins, ar / bins, al / bins, ac / bins, ad / bins);
    return 0;
}
