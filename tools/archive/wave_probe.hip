// The n = 512 wave-private kernel: where the time of the side-by-side form goes (parts knocked out -- no twiddles, no LDS
// exchanges, no arithmetic at all: timing only, nothing meaningful is computed) and the forms that take a wave's four
// transforms one at a time, persistent or not (tools/wave_kernel_variants.h), beside the kernel the library shipped for a
// while (tools/wave_kernel.h), in place and out of place, at a footprint of 2^lg samples.
//   wave_probe [log2_samples = 32] [rounds = 3]
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ifft_wgpu_amd/csrc -Itools tools/wave_probe.hip -o tools/wave_probe
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wave_kernel.h"
#include "wave_kernel_variants.h"

using fwa::v2f;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_fill(v2f *p, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = fwa::gen_sample(0x5EED, i, 1.0f / 1048576.0f);
}

template <int KNOCK>
static void launch(const v2f *src, v2f *dst, const v2f *tw, uint64_t n_samples, hipStream_t st)
{
    hipLaunchKernelGGL((fwa_probe::k_wave512<fwa::FWD, KNOCK>), dim3((uint32_t)(n_samples / 8192)), dim3(256), 0, st, src, dst, tw, n_samples, 1.0f);
}

static void launch_lib(const v2f *src, v2f *dst, const v2f *tw, uint64_t n_samples, hipStream_t st)
{
    hipLaunchKernelGGL((fwa::k_wave512<fwa::FWD>), dim3((uint32_t)(n_samples / 8192)), dim3(256), 0, st, src, dst, tw, n_samples, 1.0f);
}

template <bool PERSIST, int GRID>
static void launch_s(const v2f *src, v2f *dst, const v2f *tw, uint64_t n_samples, hipStream_t st)
{
    const uint32_t chunks = (uint32_t)(n_samples / 8192);
    hipLaunchKernelGGL((fwa_probe::k_wave512s<fwa::FWD, PERSIST, 0>), dim3(PERSIST ? (GRID < (int)chunks ? GRID : chunks) : chunks), dim3(256), 0, st, src, dst, tw,
                       n_samples, 1.0f);
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? std::atoi(argv[1]) : 32;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 3;
    const uint64_t n_samples = 1ull << lg;
    v2f *a = nullptr, *b = nullptr, *tw = nullptr;
    CK(hipMalloc(&a, n_samples * 8));
    CK(hipMalloc(&b, n_samples * 8));
    std::vector<v2f> h(256);
    for (int k = 0; k < 256; ++k) h[k] = v2f{(float)std::cos(-2.0 * M_PI * k / 512.0), (float)std::sin(-2.0 * M_PI * k / 512.0)};
    CK(hipMalloc(&tw, 256 * 8));
    CK(hipMemcpy(tw, h.data(), 256 * 8, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V { const char *name; void (*fn)(const v2f *, v2f *, const v2f *, uint64_t, hipStream_t); };
    const V vs[] = {{"full", launch<0>}, {"no_twiddles", launch<1>}, {"no_exchanges", launch<2>}, {"no_twiddles_no_exchanges", launch<3>},
                    {"loads_and_stores_only", launch<4 | 2>}, {"serial_transforms", launch_s<false, 0>},
                    {"serial_persistent_grid1024", launch_s<true, 1024>}, {"serial_persistent_grid2048", launch_s<true, 2048>},
                    {"serial_persistent_grid4096", launch_s<true, 4096>},
                    {"library_kernel", launch_lib}};
    // the serial variants compute the same bits as the library kernel
    {
        const uint64_t ns = 1ull << 24;
        hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, st, a, ns);
        launch<0>(a, b, tw, ns, st);
        std::vector<v2f> ref(ns), got(ns);
        CK(hipMemcpyAsync(ref.data(), b, ns * 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        for (int k = 5; k < 10; ++k) {
            CK(hipMemsetAsync(b, 0, ns * 8, st));
            vs[k].fn(a, b, tw, ns, st);
            CK(hipMemcpyAsync(got.data(), b, ns * 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            size_t bad = 0;
            for (uint64_t i = 0; i < ns; ++i) bad += (got[i].x != ref[i].x) || (got[i].y != ref[i].y);
            std::printf("{\"check\": \"%s\", \"mismatching_samples\": %zu}\n", vs[k].name, bad);
            if (bad) return 1;
        }
    }
    // interleaved: every round runs every variant (three launches each), so that drift of the box hits all of them alike
    constexpr int NV = sizeof(vs) / sizeof(vs[0]);
    for (int place = 0; place < 2; ++place) {
        std::vector<float> ms[NV];
        for (int r = 0; r < rounds + 1; ++r)
            for (int k = 0; k < NV; ++k)
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, st, a, n_samples);
                    CK(hipEventRecord(e0, st));
                    vs[k].fn(a, place ? b : a, tw, n_samples, st);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float t = 0;
                    CK(hipEventElapsedTime(&t, e0, e1));
                    if (r) ms[k].push_back(t);
                }
        for (int k = 0; k < NV; ++k) {
            std::sort(ms[k].begin(), ms[k].end());
            const double med = ms[k][ms[k].size() / 2];
            std::printf("{\"variant\": \"%s\", \"placement\": \"%s\", \"log2_samples\": %d, \"ms_median\": %.4f, \"ms_min\": %.4f, \"TBps\": %.3f, "
                        "\"roofline_frac\": %.4f, \"samples\": %zu}\n", vs[k].name, place ? "out_of_place" : "in_place", lg, med, ms[k].front(),
                        16.0 * n_samples / (med * 1e-3) / 1e12, 16.0 * n_samples / (med * 1e-3) / 8e12, ms[k].size());
            std::fflush(stdout);
        }
    }
    return 0;
}
