// tools/small32_persist_probe.hip -- VERDICT round 2, item 4 (2^15 at 0.54-0.56, one 1024-thread workgroup per CU): does a
// PERSISTENT form of k_small32<15> help -- one workgroup per CU slot walking through transforms, so that the loads of the
// next transform are issued right behind the stores of the current one (no workgroup launch between them, load latency
// under the store drain)?  Same body (small32_body of small32_kernel.h), same arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/small32_persist_probe tools/small32_persist_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fft_wgpu_amd/csrc/small32_kernel.h"

namespace fwa {
template <int LGN, int DIR>
__global__ __launch_bounds__((LGN <= 13 ? 256 : (1 << (LGN - 5))), 4) void k_small32_persistent(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                                                               const v2f *__restrict__ tw, uint64_t batch,
                                                                                               float scale, uint64_t n_blocks)
{
    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        // opaque per-iteration copy of the thread id: without it LICM hoists ~100 lane-constant LDS / global offsets out of
        // the loop and spills them (the first version of this probe ran 2.4x SLOWER than the plain kernel for that reason)
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        small32_body<LGN, DIR>(src, dst, tw, batch, scale, blk, tid);
        __syncthreads();  // the last exchange reads of this transform are done before the next one's first writes
    }
}
}  // namespace fwa

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using fwa::v2f;
__global__ void k_fill(v2f *d, uint64_t n) { for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) d[i] = fwa::gen_sample(1, i, 1e-6f); }

template <int LGN> static void run(uint64_t total_lg)
{
    const uint64_t n = 1ull << LGN, batch = 1ull << (total_lg - LGN), xpw = fwa::small32_xpw(LGN), blocks = (batch + xpw - 1) / xpw;
    v2f *buf, *tw;
    CK(hipMalloc(&buf, n * batch * 8)); CK(hipMalloc(&tw, n / 2 * 8));
    std::vector<v2f> h(n / 2, v2f{0.6f, 0.8f});
    CK(hipMemcpy(tw, h.data(), n / 2 * 8, hipMemcpyHostToDevice));
    const int threads = LGN <= 13 ? 256 : (1 << (LGN - 5));
    const size_t lds = fwa::small32_lds(LGN);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwa::k_small32<LGN, fwa::FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwa::k_small32_persistent<LGN, fwa::FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&fwa::k_small32_persistent<LGN, fwa::FWD>), threads, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, buf, n * batch);
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        return best;
    };
    const float a = timed([&] { hipLaunchKernelGGL((fwa::k_small32<LGN, fwa::FWD>), dim3((uint32_t)blocks), dim3(threads), lds, 0, buf, buf, tw, batch, 1.0f); });
    printf("n = 2^%d x %llu (%.0f GiB): one workgroup per %llu transform(s), %d resident per CU: %.3f ms = %.3f of the 8 TB/s roofline\n", LGN,
           (unsigned long long)batch, n * batch * 8.0 / (1ull << 30), (unsigned long long)xpw, per_cu, a, 16.0 * n * batch / (a * 1e-3) / 8e12);
    for (int mult : {1, 2}) {
        const uint32_t grid = 256u * per_cu * mult;
        const float b = timed([&] {
            hipLaunchKernelGGL((fwa::k_small32_persistent<LGN, fwa::FWD>), dim3(grid), dim3(threads), lds, 0, buf, buf, tw, batch, 1.0f, blocks);
        });
        printf("    persistent, grid %u (%d x the resident slots): %.3f ms = %.3f\n", grid, mult, b, 16.0 * n * batch / (b * 1e-3) / 8e12);
    }
    CK(hipFree(buf)); CK(hipFree(tw));
}

int main(int argc, char **argv)
{
    const uint64_t total_lg = argc > 1 ? atoi(argv[1]) : 30;  // samples per size: 2^30 = 8 GiB
    run<15>(total_lg); run<14>(total_lg); run<13>(total_lg); run<10>(total_lg);
    return 0;
}
