// tools/tile_probe4.hip -- measurement tool: does a PERSISTENT tile loop that prefetches tile i+1 while tile
// i "computes" and stores lift the per-CU bytes in flight enough to approach the fabric ceiling?
// Same four streams and tile shapes as tile_probe3 (8 B/lane, tile-contiguous ring), no real arithmetic;
// `work` iterations of dependent FMAs per element stand in for the FFT between load and store.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile_probe4 tools/tile_probe4.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 2, SC1 = 16;
constexpr uint32_t TB = 8u << 20;

struct Item { uint32_t role, tile, t; };
__device__ __forceinline__ Item decode(uint32_t it) { return Item{it & 1, (it >> 1) & 63, it >> 7}; }

__device__ __forceinline__ void load_tile(v2u (&x)[32], Item w, const char *big_in, const char *ring, uint32_t ring_slots, uint32_t tid)
{
    if (w.role == 0) {
        auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(big_in) + (size_t)w.t * TB, 0, TB, 0x00020000);
        const uint32_t vo = ((tid >> 4) * 1024 + (tid & 15)) * 8;
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(r, vo, w.tile * 128 + j * 262144, NT);
    } else {
        auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(ring) + (size_t)(w.t % ring_slots) * TB, 0, TB, 0x00020000);
        const uint32_t np = tid & 31, rr = tid >> 5;
        const uint32_t vi = (np >> 4) * 131072 + rr * 128 + (np & 15) * 8;
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(r, vi, w.tile * 2048 + j * 262144, 0);
    }
}
__device__ __forceinline__ void store_tile(v2u (&x)[32], Item w, char *big_out, char *ring, uint32_t ring_slots, uint32_t tid)
{
    if (w.role == 0) {
        auto r = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)(w.t % ring_slots) * TB, 0, TB, 0x00020000);
        const uint32_t vs = ((tid >> 4) * 16 + (tid & 15)) * 8;
#pragma unroll
        for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], r, vs, w.tile * 131072 + j * 4096, SC1);
    } else {
        auto r = __builtin_amdgcn_make_buffer_rsrc(big_out + (size_t)w.t * TB, 0, TB, 0x00020000);
        const uint32_t vo = ((tid >> 4) * 1024 + (tid & 15)) * 8;
#pragma unroll
        for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], r, vo, w.tile * 128 + j * 262144, NT);
    }
}
__device__ __forceinline__ void fake_work(v2u (&x)[32], int work)
{
    for (int w = 0; w < work; ++w)
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            v2f f = __builtin_bit_cast(v2f, x[j]);
            f = f * 1.0001f + 0.5f;
            x[j] = __builtin_bit_cast(v2u, f);
        }
}

template <bool PREFETCH, int WPS>
__global__ __launch_bounds__(512, WPS) void k(const char *big_in, char *big_out, char *ring, uint32_t ring_slots,
                                              uint32_t n_items, int work)
{
    uint32_t tid = threadIdx.x;
    const uint32_t G = gridDim.x;
    if constexpr (!PREFETCH) {
        for (uint32_t it = blockIdx.x; it < n_items; it += G) {
            asm volatile("" : "+v"(tid));
            v2u x[32];
            Item w = decode(it);
            load_tile(x, w, big_in, ring, ring_slots, tid);
            fake_work(x, work);
            store_tile(x, w, big_out, ring, ring_slots, tid);
        }
    } else {
        v2u a[32], b[32];
        uint32_t it = blockIdx.x;
        if (it >= n_items) return;
        load_tile(a, decode(it), big_in, ring, ring_slots, tid);
        for (;;) {
            asm volatile("" : "+v"(tid));
            const uint32_t nxt = it + G;
            if (nxt < n_items) load_tile(b, decode(nxt), big_in, ring, ring_slots, tid);
            fake_work(a, work);
            store_tile(a, decode(it), big_out, ring, ring_slots, tid);
            if (nxt >= n_items) break;
            const uint32_t nxt2 = nxt + G;
            if (nxt2 < n_items) load_tile(a, decode(nxt2), big_in, ring, ring_slots, tid);
            fake_work(b, work);
            store_tile(b, decode(nxt), big_out, ring, ring_slots, tid);
            if (nxt2 >= n_items) break;
            it = nxt2;
        }
    }
}

int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 1024, ring_slots = argc > 2 ? atoi(argv[2]) : 16;
    char *a, *b, *ring;
    CK(hipMalloc(&a, (size_t)batch * TB)); CK(hipMalloc(&b, (size_t)batch * TB)); CK(hipMalloc(&ring, (size_t)ring_slots * TB));
    CK(hipMemset(a, 1, (size_t)batch * TB)); CK(hipMemset(b, 1, (size_t)batch * TB)); CK(hipMemset(ring, 1, (size_t)ring_slots * TB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char *name; int pre, wgs, work; } vs[] = {
        {"no prefetch, 512 WGs (2/CU), work 0", 0, 512, 0},   {"no prefetch, 512 WGs (2/CU), work 8", 0, 512, 8},
        {"no prefetch, 512 WGs (2/CU), work 16", 0, 512, 16}, {"no prefetch, 256 WGs (1/CU), work 8", 0, 256, 8},
        {"prefetch,    256 WGs (1/CU), work 0", 1, 256, 0},   {"prefetch,    256 WGs (1/CU), work 8", 1, 256, 8},
        {"prefetch,    256 WGs (1/CU), work 16", 1, 256, 16}, {"prefetch,    256 WGs (1/CU), work 32", 1, 256, 32},
        {"prefetch,    512 WGs (spills?), work 8", 2, 512, 8},
    };
    const uint32_t n_items = batch * 128;
    for (auto &v : vs) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (v.pre == 0) hipLaunchKernelGGL((k<false, 4>), dim3(v.wgs), dim3(512), 0, 0, a, b, ring, ring_slots, n_items, v.work);
            else if (v.pre == 1) hipLaunchKernelGGL((k<true, 2>), dim3(v.wgs), dim3(512), 0, 0, a, b, ring, ring_slots, n_items, v.work);
            else hipLaunchKernelGGL((k<true, 4>), dim3(v.wgs), dim3(512), 0, 0, a, b, ring, ring_slots, n_items, v.work);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (t < best) best = t;
        }
        const double bytes = 4.0 * batch * TB;
        printf("%-42s %8.3f ms  %7.0f GB/s -> %6.2f ms at batch 4096 (%4.1f%%)\n", v.name, best, bytes / (best * 1e-3) / 1e9,
               best * 4096.0 / batch, 100.0 * 16.0 * batch * (1 << 20) / (best * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
