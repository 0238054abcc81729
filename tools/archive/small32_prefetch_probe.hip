// k_small32 with its twiddle look-ups issued before the data loads (PREFETCH bit 0: stage 0, bit 1: stage 1 of the
// three-stage sizes) or right behind them, before the wait (bits 2, 3) against the look-ups at the point of use, n = 2^lg (10 .. 15), 2^32 samples, interleaved; in place for
// even log2 n and out of place for odd, as the plans run them.  Bit-identical by construction (checked on 2^24 samples).
//   small32_prefetch_probe [rounds = 4]
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ifft_wgpu_amd/csrc tools/small32_prefetch_probe.hip -o tools/small32_prefetch_probe
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "small32_kernel.h"

using fwa::v2f;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_fill(v2f *p, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = fwa::gen_sample(0x5EED, i, 1.0f / 1048576.0f);
}

template <int LGN, int P>
static hipError_t go(const v2f *src, v2f *dst, const v2f *tw, uint64_t batch, hipStream_t st)
{
    constexpr int threads = LGN <= 13 ? 256 : (1 << (LGN - 5));
    const uint32_t xpw = fwa::small32_xpw(LGN);
    const size_t lds = fwa::small32_lds(LGN);
    static bool once = false;
    if (!once) {
        once = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fwa::k_small32<LGN, fwa::FWD, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((fwa::k_small32<LGN, fwa::FWD, P>), dim3((uint32_t)((batch + xpw - 1) / xpw)), dim3(threads), lds, st, src, dst, tw, batch, 1.0f);
    return hipGetLastError();
}

template <int LGN>
static int run(v2f *a, v2f *b, int rounds, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    constexpr uint32_t n = 1u << LGN;
    const uint64_t total = 1ull << 32, batch = total >> LGN;
    std::vector<v2f> h(n / 2);
    for (uint32_t k = 0; k < n / 2; ++k) h[k] = v2f{(float)std::cos(-2.0 * M_PI * k / n), (float)std::sin(-2.0 * M_PI * k / n)};
    v2f *tw = nullptr;
    CK(hipMalloc(&tw, n / 2 * 8));
    CK(hipMemcpy(tw, h.data(), n / 2 * 8, hipMemcpyHostToDevice));
    using Fn = hipError_t (*)(const v2f *, v2f *, const v2f *, uint64_t, hipStream_t);
    const Fn fns[6] = {go<LGN, 0>, go<LGN, 1>, go<LGN, 4>, go<LGN, 3>, go<LGN, 12>, go<LGN, 9>};
    const int pf[6] = {0, 1, 4, 3, 12, 9};
    const int nv = LGN <= 10 ? 3 : 6;   // two-stage sizes have no stage-1 twiddles
    v2f *dst = (LGN % 2) ? b : a;
    // identical bits
    {
        const uint64_t ns = 1ull << 24, bt = ns >> LGN;
        std::vector<v2f> ref(ns), got(ns);
        for (int v = 0; v < nv; ++v) {
            hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, st, a, ns);
            CK(fns[v](a, b, tw, bt, st));
            CK(hipMemcpyAsync((v ? got : ref).data(), b, ns * 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (v && std::memcmp(got.data(), ref.data(), ns * 8) != 0) { std::fprintf(stderr, "lg %d prefetch %d: results differ\n", LGN, pf[v]); return 1; }
        }
    }
    std::vector<float> ms[6];
    for (int r = 0; r < rounds + 1; ++r)
        for (int v = 0; v < nv; ++v)
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, st, a, total);
                CK(hipEventRecord(e0, st));
                CK(fns[v](a, dst, tw, batch, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float t = 0;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (r) ms[v].push_back(t);
            }
    for (int v = 0; v < nv; ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        const double med = ms[v][ms[v].size() / 2];
        std::printf("{\"lg_n\": %d, \"prefetch\": %d, \"placement\": \"%s\", \"ms_median\": %.4f, \"ms_min\": %.4f, \"roofline_frac\": %.4f, \"samples\": %zu}\n",
                    LGN, pf[v], (LGN % 2) ? "out_of_place" : "in_place", med, ms[v].front(), 16.0 * total / (med * 1e-3) / 8e12, ms[v].size());
        std::fflush(stdout);
    }
    CK(hipFree(tw));
    return 0;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 4;
    v2f *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, 8ull << 32));
    CK(hipMalloc(&b, 8ull << 32));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int rc = 0;
    if (!rc) rc = run<9>(a, b, rounds, st, e0, e1);
    if (!rc) rc = run<10>(a, b, rounds, st, e0, e1);
    if (!rc) rc = run<11>(a, b, rounds, st, e0, e1);
    if (!rc) rc = run<12>(a, b, rounds, st, e0, e1);
    if (!rc) rc = run<13>(a, b, rounds, st, e0, e1);
    if (!rc) rc = run<14>(a, b, rounds, st, e0, e1);
    if (!rc) rc = run<15>(a, b, rounds, st, e0, e1);
    return rc;
}
