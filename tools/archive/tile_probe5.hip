// tools/tile_probe5.hip -- measurement tool: walk from the linear 4-stream probe (8.6 TB/s) to the tile
// structure (6.4 TB/s) one property at a time: lane width, accesses per thread, workgroup size/occupancy,
// and the HBM-side address pattern.  Persistent grid; items alternate pass-1-like (big -> ring) and
// pass-2-like (ring -> big2); each item moves THREADS*U*LB bytes per stream.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile_probe5 tools/tile_probe5.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 2, SC1 = 16;

template <int LB> struct Vec;
template <> struct Vec<8> { typedef v2u T; };
template <> struct Vec<16> { typedef v4u T; };
template <int LB, int AUX> __device__ __forceinline__ typename Vec<LB>::T ld(__amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s)
{
    if constexpr (LB == 8) return __builtin_amdgcn_raw_buffer_load_b64(r, v, s, AUX);
    else return __builtin_amdgcn_raw_buffer_load_b128(r, v, s, AUX);
}
template <int LB, int AUX> __device__ __forceinline__ void st(typename Vec<LB>::T x, __amdgpu_buffer_rsrc_t r, uint32_t v, uint32_t s)
{
    if constexpr (LB == 8) __builtin_amdgcn_raw_buffer_store_b64(x, r, v, s, AUX);
    else __builtin_amdgcn_raw_buffer_store_b128(x, r, v, s, AUX);
}

// big regions are processed in 8-MiB "transforms" (buffer descriptors are 32-bit); ring = ring_slots transforms.
// TILE: HBM side addressed as 16-column tiles of a 1024x1024 8-byte matrix (needs LB == 8, THREADS*U == 16384);
// otherwise the item's THREADS*U*LB bytes are contiguous.  Ring side always contiguous.
template <int LB, int U, int THREADS, int WPS, bool TILE>
__global__ __launch_bounds__(THREADS, WPS) void k(const char *big_in, char *big_out, char *ring, uint32_t ring_slots,
                                                  uint32_t n_items)
{
    constexpr uint32_t TB = 8u << 20, ITEM = THREADS * U * LB, IPT = TB / ITEM;  // items per transform
    uint32_t tid = threadIdx.x;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        asm volatile("" : "+v"(tid));
        const uint32_t role = it & 1, idx = it >> 1, t = idx / IPT, sub = idx % IPT;
        auto rbig = __builtin_amdgcn_make_buffer_rsrc((role ? big_out : const_cast<char *>(big_in)) + (size_t)t * TB, 0, TB, 0x00020000);
        auto rring = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)(t % ring_slots) * TB, 0, TB, 0x00020000);
        typename Vec<LB>::T x[U];
        const uint32_t vlin = tid * LB, slin = sub * ITEM;
        uint32_t vb = vlin, sb = slin, stepb = THREADS * LB;
        if constexpr (TILE) { vb = ((tid >> 4) * 1024 + (tid & 15)) * 8; sb = sub * 128; stepb = (THREADS / 16) * 8192; }
        if (role == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld<LB, NT>(rbig, vb, sb + u * stepb);
#pragma unroll
            for (int u = 0; u < U; ++u) st<LB, SC1>(x[u], rring, vlin, slin + u * THREADS * LB);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld<LB, 0>(rring, vlin, slin + u * THREADS * LB);
#pragma unroll
            for (int u = 0; u < U; ++u) st<LB, NT>(x[u], rbig, vb, sb + u * stepb);
        }
    }
}
typedef void (*kern_t)(const char *, char *, char *, uint32_t, uint32_t);
struct V { const char *name; kern_t k; int threads, item_bytes, blocks_per_cu; };
int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 1024, ring_slots = argc > 2 ? atoi(argv[2]) : 16;
    constexpr size_t TB = 8u << 20;
    char *a, *b, *ring;
    CK(hipMalloc(&a, batch * TB)); CK(hipMalloc(&b, batch * TB)); CK(hipMalloc(&ring, ring_slots * TB));
    CK(hipMemset(a, 1, batch * TB)); CK(hipMemset(b, 1, batch * TB)); CK(hipMemset(ring, 1, ring_slots * TB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    V vs[] = {
        {"16B x4,  256 thr, 8 blk/CU, linear ", k<16, 4, 256, 8, false>, 256, 16 * 4 * 256, 8},
        {"8B  x4,  256 thr, 8 blk/CU, linear ", k<8, 4, 256, 8, false>, 256, 8 * 4 * 256, 8},
        {"8B  x8,  256 thr, 8 blk/CU, linear ", k<8, 8, 256, 8, false>, 256, 8 * 8 * 256, 8},
        {"16B x16, 256 thr, 4 blk/CU, linear ", k<16, 16, 256, 4, false>, 256, 16 * 16 * 256, 4},
        {"8B  x32, 256 thr, 4 blk/CU, linear ", k<8, 32, 256, 4, false>, 256, 8 * 32 * 256, 4},
        {"8B  x32, 512 thr, 2 blk/CU, linear ", k<8, 32, 512, 4, false>, 512, 8 * 32 * 512, 2},
        {"16B x16, 512 thr, 2 blk/CU, linear ", k<16, 16, 512, 4, false>, 512, 16 * 16 * 512, 2},
        {"8B  x32, 512 thr, 2 blk/CU, TILE   ", k<8, 32, 512, 4, true>, 512, 8 * 32 * 512, 2},
        {"8B  x8,  512 thr, 4 blk/CU, linear ", k<8, 8, 512, 8, false>, 512, 8 * 8 * 512, 4},
    };
    for (auto &v : vs) {
        const uint32_t n_items = (uint32_t)(2 * batch * (TB / v.item_bytes));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(256 * v.blocks_per_cu), dim3(v.threads), 0, 0, a, b, ring, ring_slots, n_items);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (t < best) best = t;
        }
        const double bytes = 4.0 * batch * TB;
        printf("%-40s %8.3f ms  %7.0f GB/s -> %6.2f ms at batch 4096 (%4.1f%%)\n", v.name, best, bytes / (best * 1e-3) / 1e9,
               best * 4096.0 / batch, 100.0 * 16.0 * batch * (1 << 20) / (best * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
