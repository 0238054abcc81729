// tools/stream_overlap_probe.hip -- do two freshly created HIP streams run kernels concurrently on this stack?
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_overlap_probe tools/stream_overlap_probe.hip
// Background (round 3): the two-chain pipelines of the library ran 15 % slower for some PLAN INSTANCES of one process --
// the time of a single chain -- and the same for every exec of that instance: the two internal streams of that instance did
// not overlap.  This probe creates stream pairs the way a plan does and measures, per pair, the time of two 30-us spin
// kernels (one per stream): ~30 us = concurrent, ~60 us = serialised.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(unsigned ticks, unsigned *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0xffffffffu) sink[0] = 1;
}

static float pair_time(hipStream_t caller, hipStream_t a, hipStream_t b, hipEvent_t fork, hipEvent_t da, hipEvent_t db, hipEvent_t e0, hipEvent_t e1,
                       unsigned *sink, int blocks_in)
{
    const int blocks = blocks_in < 0 ? -blocks_in : blocks_in;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, caller));
        CK(hipEventRecord(fork, caller));
        CK(hipStreamWaitEvent(a, fork, 0)); CK(hipStreamWaitEvent(b, fork, 0));
        hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, a, 3000u, sink);
        if (blocks_in > 0) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, b, 3000u, sink);
        CK(hipEventRecord(da, a)); CK(hipEventRecord(db, b));
        CK(hipStreamWaitEvent(caller, da, 0)); CK(hipStreamWaitEvent(caller, db, 0));
        CK(hipEventRecord(e1, caller));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main(int argc, char **argv)
{
    const int pairs = argc > 1 ? atoi(argv[1]) : 24;
    const int keep = argc > 2 ? atoi(argv[2]) : 0;   // 1: never destroy the pairs
    const int blocks = argc > 3 ? atoi(argv[3]) : 256;
    unsigned *sink; CK(hipMalloc(&sink, 64));
    hipStream_t caller; CK(hipStreamCreateWithFlags(&caller, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("pairs %d keep %d blocks %d: us single/pair\n", pairs, keep, blocks);
    for (int i = 0; i < pairs; ++i) {
        hipStream_t a, b; hipEvent_t fork, da, db;
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&da, hipEventDisableTiming));
        CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&db, hipEventDisableTiming));
        printf(" %.0f/%.0f", pair_time(caller, a, b, fork, da, db, e0, e1, sink, -blocks), pair_time(caller, a, b, fork, da, db, e0, e1, sink, blocks));
        fflush(stdout);
        if (!keep) {
            CK(hipDeviceSynchronize());
            CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
            CK(hipEventDestroy(fork)); CK(hipEventDestroy(da)); CK(hipEventDestroy(db));
        }
    }
    printf("\n");
    return 0;
}
