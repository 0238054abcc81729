// tools/tile_probe2.hip -- measurement tool: is the HBM-side column-tile pattern limited by the power-of-two
// row pitch (channel camping) or by the 128-byte segment width?  Pass-1/pass-2-like skeleton as tile_probe,
// ring tile-contiguous, 8 B per lane, with the `big` row pitch and the tile width as parameters.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile_probe2 tools/tile_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 2, SC1 = 16;

// W columns per tile (16 or 32), threads = W*32, each thread 32 elements.  PITCH = bytes per matrix row in `big`.
template <int W, uint32_t PITCH>
__global__ __launch_bounds__(W * 32) void k_tiles(const char *big_in, char *big_out, char *ring, uint32_t ring_slots,
                                                 uint32_t batch)
{
    constexpr uint32_t TBIG = PITCH * 1024, TRING = 8u << 20, TILES = 1024 / W;
    const uint32_t tid = threadIdx.x;
    const uint32_t role = blockIdx.x & 1, idx = blockIdx.x >> 1;
    const uint32_t tile = idx % TILES, t = idx / TILES;
    if (t >= batch) return;
    auto rring = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)(t % ring_slots) * TRING, 0, TRING, 0x00020000);
    const uint32_t c = tid % W, q = tid / W;  // 32 rows per pass
    v2u x[32];
    if (role == 0) {
        auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(big_in) + (size_t)t * TBIG, 0, TBIG, 0x00020000);
        const uint32_t vo = q * PITCH + c * 8;
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, vo, tile * (W * 8) + j * (32 * PITCH), NT);
        const uint32_t vs = (q * W + c) * 8;  // tile-contiguous ring: [row][W cols]
#pragma unroll
        for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rring, vs, tile * (W * 8192) + j * (32 * W * 8), SC1);
    } else {
        auto rout = __builtin_amdgcn_make_buffer_rsrc(big_out + (size_t)t * TBIG, 0, TBIG, 0x00020000);
        // rows [W*tile, W*tile+W) of source tile s = one contiguous W*W*8-byte chunk at s*(W*8192) + tile*(W*W*8)
        constexpr uint32_t CH = W * W * 8, LPC = CH / 8;  // lanes per chunk
        const uint32_t e = tid % LPC, part = tid / LPC;   // (W*32)/LPC chunks per pass
        constexpr uint32_t CPP = (W * 32) / LPC;
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(rring, e * 8, (j * CPP + part) * (W * 8192) + tile * CH, 0);
        const uint32_t vo = q * PITCH + c * 8;
#pragma unroll
        for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rout, vo, tile * (W * 8) + j * (32 * PITCH), NT);
    }
}
typedef void (*kern_t)(const char *, char *, char *, uint32_t, uint32_t);
struct Variant { const char *name; kern_t k; int w; uint32_t pitch; };
int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 1024, ring_slots = argc > 2 ? atoi(argv[2]) : 16;
    const size_t maxT = (size_t)(8192 + 512) * 1024;
    char *a, *b, *ring;
    CK(hipMalloc(&a, batch * maxT)); CK(hipMalloc(&b, batch * maxT)); CK(hipMalloc(&ring, (size_t)ring_slots << 23));
    CK(hipMemset(a, 1, batch * maxT)); CK(hipMemset(b, 1, batch * maxT)); CK(hipMemset(ring, 1, (size_t)ring_slots << 23));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Variant vs[] = {
        {"W=16 (128-B segments), pitch 8192       ", k_tiles<16, 8192>, 16, 8192},
        {"W=16 (128-B segments), pitch 8192+128   ", k_tiles<16, 8320>, 16, 8320},
        {"W=16 (128-B segments), pitch 8192+256   ", k_tiles<16, 8448>, 16, 8448},
        {"W=16 (128-B segments), pitch 8192+512   ", k_tiles<16, 8704>, 16, 8704},
        {"W=32 (256-B segments), pitch 8192       ", k_tiles<32, 8192>, 32, 8192},
        {"W=32 (256-B segments), pitch 8192+256   ", k_tiles<32, 8448>, 32, 8448},
    };
    printf("batch %u, ring %u slots\n", batch, ring_slots);
    for (auto &v : vs) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(batch * 2 * (1024 / v.w)), dim3(v.w * 32), 0, 0, a, b, ring, ring_slots, batch);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = 4.0 * batch * (8u << 20);
        printf("%-44s %8.3f ms  %7.0f GB/s total -> %6.2f ms at batch 4096 (%5.1f%% roofline)\n", v.name, best,
               bytes / (best * 1e-3) / 1e9, best * 4096.0 / batch, 100.0 * (16.0 * batch * (1 << 20)) / (best * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
