// tools/fabric_probe3.hip -- measurement tool (round 3): does the ALLOCATION kind of the streamed
// buffer or of the ring change what the 256-MiB Infinity Cache keeps?
//   hipcc --offload-arch=gfx950 -O3 -o tools/fabric_probe3 tools/fabric_probe3.hip
// Same linear 16-B/lane streams as fabric_probe2.hip over a "big" (2 GiB, HBM) and a "small" (ring-sized) region, but
// each region is allocated in turn with hipMalloc (default, coarse-grained), hipExtMallocWithFlags(...Finegrained) and
// hipExtMallocWithFlags(...Uncached).  Columns: "rd big+wr small" = the traffic mix of pass 1 (HBM read + ring write),
// "wr big+rd small" = pass 2 (ring read + HBM write), "FFT mix" = both at once.
// Question: does any combination hold the <= 64-MiB-ring rates (8.2-8.7 / 10.6 TB/s) with a 256-MiB ring?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode bits: 1 read big, 2 write big, 4 read small, 8 write small
template <int LB, int SB, int LS, int SS>
__global__ __launch_bounds__(256) void k_probe(void *big_r, void *big_w, void *small_r, void *small_w,
                                               uint32_t big_bytes, uint32_t small_bytes, uint32_t iters, int mode,
                                               unsigned *sink)
{
    constexpr int U = 4;
    auto rbr = __builtin_amdgcn_make_buffer_rsrc(big_r, 0, big_bytes + (1 << 20), 0x00020000);
    auto rbw = __builtin_amdgcn_make_buffer_rsrc(big_w, 0, big_bytes + (1 << 20), 0x00020000);
    auto rsr = __builtin_amdgcn_make_buffer_rsrc(small_r, 0, small_bytes + (1 << 20), 0x00020000);
    auto rsw = __builtin_amdgcn_make_buffer_rsrc(small_w, 0, small_bytes + (1 << 20), 0x00020000);
    v4u acc = {0, 0, 0, 0};
    const uint32_t stride = gridDim.x * 256 * U * 16;
    uint32_t i = (blockIdx.x * 256 * U + threadIdx.x) * 16;
    for (uint32_t it = 0; it < iters; ++it, i += stride) {
        const uint32_t ib = i & (big_bytes - 1), is = i & (small_bytes - 1);
        v4u a[U], b[U];
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = __builtin_amdgcn_raw_buffer_load_b128(rbr, ib, u * 4096, LB);
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = __builtin_amdgcn_raw_buffer_load_b128(rsr, is, u * 4096, LS);
        if (mode & 1)
#pragma unroll
            for (int u = 0; u < U; ++u) acc += a[u];
        if (mode & 4)
#pragma unroll
            for (int u = 0; u < U; ++u) acc += b[u];
        if (mode & 2)
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)u, rbw, ib, u * 4096, SB);
        if (mode & 8)
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)u, rsw, is, u * 4096, SS);
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y + acc.z + acc.w;
}

typedef void (*kern_t)(void *, void *, void *, void *, uint32_t, uint32_t, uint32_t, int, unsigned *);
struct Variant { const char *name; kern_t k; };

static const char *kind_name(int k) { return k == 0 ? "default" : (k == 1 ? "finegrained" : "uncached"); }
static void alloc_kind(void **p, size_t bytes, int kind)
{
    if (kind == 0) CK(hipMalloc(p, bytes));
    else CK(hipExtMallocWithFlags(p, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
}

int main(int argc, char **argv)
{
    const uint32_t big_bytes = 1u << 31;
    const uint32_t small_bytes = (uint32_t)((argc > 1 ? strtoull(argv[1], 0, 10) : 256ull) << 20);
    const int blocks = argc > 2 ? atoi(argv[2]) : 2048;
    unsigned *sink;
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t per_iter = (uint64_t)blocks * 256 * 4 * 16;
    const uint32_t iters = (uint32_t)((8ull << 30) / per_iter);
    Variant vs[] = {
        {"default policies     ", k_probe<0, 0, 0, 0>},
        {"nt big, sc1 small    ", k_probe<2, 2, 16, 16>},   // the shipped pipeline's policies
        {"nt big, sc1 small st ", k_probe<2, 2, 0, 16>},
        {"nt big, plain small  ", k_probe<2, 2, 0, 0>},
    };
    int modes[] = {9, 6, 15};
    const char *mn[] = {"rd big+wr small", "wr big+rd small", "FFT mix(15)"};
    printf("small (ring) %u MiB, big 2048 MiB, %d blocks; GB/s total per mode\n", small_bytes >> 20, blocks);
    printf("%-14s%-14s%-24s", "big alloc", "small alloc", "policy");
    for (auto m : mn) printf("%18s", m);
    printf("\n");
    for (int kb = 0; kb < 3; ++kb)
        for (int ks = 0; ks < 3; ++ks) {
            void *big_a, *big_b, *small_a, *small_b;
            alloc_kind(&big_a, (size_t)big_bytes + (1 << 20), kb); alloc_kind(&big_b, (size_t)big_bytes + (1 << 20), kb);
            alloc_kind(&small_a, (size_t)small_bytes + (1 << 20), ks); alloc_kind(&small_b, (size_t)small_bytes + (1 << 20), ks);
            CK(hipMemset(big_a, 1, big_bytes)); CK(hipMemset(big_b, 1, big_bytes));
            CK(hipMemset(small_a, 1, small_bytes)); CK(hipMemset(small_b, 1, small_bytes));
            CK(hipDeviceSynchronize());
            for (auto &v : vs) {
                printf("%-14s%-14s%-24s", kind_name(kb), kind_name(ks), v.name);
                for (int mi = 0; mi < 3; ++mi) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 2; ++rep) {
                        CK(hipEventRecord(e0));
                        hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, big_a, big_b, small_a, small_b, big_bytes,
                                           small_bytes, iters, modes[mi], sink);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    double gb = (double)__builtin_popcount(modes[mi]) * iters * per_iter / 1e9;
                    printf("%18.0f", gb / (best * 1e-3));
                }
                printf("\n"); fflush(stdout);
            }
            CK(hipFree(big_a)); CK(hipFree(big_b)); CK(hipFree(small_a)); CK(hipFree(small_b));
        }
    return 0;
}
