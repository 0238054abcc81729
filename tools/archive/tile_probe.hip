// tools/tile_probe.hip -- measurement tool: which tile ACCESS PATTERN sustains the most bandwidth for
// the two-pass 2^20 FFT traffic (compute is fully hidden: see DESIGN.md "what limits the tiles").
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile_probe tools/tile_probe.hip
// One launch = pass-1-like workgroups (read a column tile of transform t from `big`, write it to the
// ring) interleaved with pass-2-like workgroups (read a row tile from the ring, write a column tile to
// `big`).  No arithmetic, no hazards handled: a bandwidth probe only.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t TB = 8u << 20;  // bytes per transform
constexpr int NT = 2, SC1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, TB, 0x00020000); }

// V = bytes per lane on the HBM side (8/16), Y = bytes per lane on the ring side (8/16),
// LIN = ring layout tile-contiguous (each pass-1 tile writes one contiguous 128 KiB block [row][16 cols]).
template <int V, int Y, bool LIN, int WPS /*min waves per SIMD*/>
__global__ __launch_bounds__(512, WPS) void k_tiles(const char *big_in, char *big_out, char *ring, uint32_t ring_slots,
                                                    uint32_t batch)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t role = blockIdx.x & 1, idx = blockIdx.x >> 1;
    const uint32_t tile = idx & 63, t = idx >> 6;
    if (t >= batch) return;
    auto rring = rsrc(ring + (size_t)(t % ring_slots) * TB);
    if (role == 0) {
        // ---- pass-1-like: column tile (16 cols x 1024 rows) of big_in[t] -> ring
        auto rin = rsrc(big_in + (size_t)t * TB);
        v4u x16[16]; v2u x8[32];
        if constexpr (V == 8) {
            const uint32_t c = tid & 15, q = tid >> 4;            // 16 cols x 32 rows per pass, 32 passes
            const uint32_t vo = (q * 1024 + c) * 8;
#pragma unroll
            for (int j = 0; j < 32; ++j) x8[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, vo, tile * 128 + j * 262144, NT);
        } else {
            const uint32_t cp = tid & 7, q = tid >> 3;            // 8 col-pairs x 64 rows per pass, 16 passes
            const uint32_t vo = (q * 1024 + cp * 2) * 8;
#pragma unroll
            for (int j = 0; j < 16; ++j) x16[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, vo, tile * 128 + j * 524288, NT);
        }
        // ring store
        if constexpr (Y == 8) {
            v2u y[32];
            if constexpr (V == 8) { for (int j = 0; j < 32; ++j) y[j] = x8[j]; }
            else { for (int j = 0; j < 16; ++j) { y[2 * j] = x16[j].xy; y[2 * j + 1] = x16[j].zw; } }
            const uint32_t c = tid & 15, q = tid >> 4;
            const uint32_t vo = LIN ? (q * 16 + c) * 8 : (q * 1024 + c) * 8;
#pragma unroll
            for (int j = 0; j < 32; ++j)
                __builtin_amdgcn_raw_buffer_store_b64(y[j], rring, vo, LIN ? tile * 131072 + j * 4096 : tile * 128 + j * 262144, SC1);
        } else {
            v4u y[16];
            if constexpr (V == 16) { for (int j = 0; j < 16; ++j) y[j] = x16[j]; }
            else { for (int j = 0; j < 16; ++j) y[j] = v4u{x8[2 * j].x, x8[2 * j].y, x8[2 * j + 1].x, x8[2 * j + 1].y}; }
            const uint32_t cp = tid & 7, q = tid >> 3;
            const uint32_t vo = LIN ? (q * 16 + cp * 2) * 8 : (q * 1024 + cp * 2) * 8;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(y[j], rring, vo, LIN ? tile * 131072 + j * 8192 : tile * 128 + j * 524288, SC1);
        }
    } else {
        // ---- pass-2-like: row tile (16 rows x 1024 cols) of the ring -> column tile of big_out[t]
        auto rout = rsrc(big_out + (size_t)t * TB);
        v2u y8[32]; v4u y16[16];
        if constexpr (Y == 8) {
            if constexpr (!LIN) {
                const uint32_t np = tid & 31, r = tid >> 5;       // 256 B contiguous per half-wave
                const uint32_t vo = (r * 1024 + np) * 8;
#pragma unroll
                for (int j = 0; j < 32; ++j) y8[j] = __builtin_amdgcn_raw_buffer_load_b64(rring, vo, tile * 131072 + j * 256, 0);
            } else {
                // tile-contiguous ring: rows [16*tile,16*tile+16) of source tile s are one 2-KiB chunk at s*128KiB + tile*2KiB
                const uint32_t e = tid & 255, half = tid >> 8;    // 256 lanes x 8 B = one 2-KiB chunk
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    y8[j] = __builtin_amdgcn_raw_buffer_load_b64(rring, e * 8, (2 * j + half) * 131072 + tile * 2048, 0);
            }
        } else {
            if constexpr (!LIN) {
                const uint32_t np = tid & 31, r = tid >> 5;       // 512 B contiguous per half-wave
                const uint32_t vo = (r * 1024 + np * 2) * 8;
#pragma unroll
                for (int j = 0; j < 16; ++j) y16[j] = __builtin_amdgcn_raw_buffer_load_b128(rring, vo, tile * 131072 + j * 512, 0);
            } else {
                const uint32_t e = tid & 127, qq = tid >> 7;      // 128 lanes x 16 B = one 2-KiB chunk
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    y16[j] = __builtin_amdgcn_raw_buffer_load_b128(rring, e * 16, (4 * j + qq) * 131072 + tile * 2048, 0);
            }
        }
        if constexpr (V == 8) {
            v2u x[32];
            if constexpr (Y == 8) { for (int j = 0; j < 32; ++j) x[j] = y8[j]; }
            else { for (int j = 0; j < 16; ++j) { x[2 * j] = y16[j].xy; x[2 * j + 1] = y16[j].zw; } }
            const uint32_t c = tid & 15, q = tid >> 4;
            const uint32_t vo = (q * 1024 + c) * 8;
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rout, vo, tile * 128 + j * 262144, NT);
        } else {
            v4u x[16];
            if constexpr (Y == 16) { for (int j = 0; j < 16; ++j) x[j] = y16[j]; }
            else { for (int j = 0; j < 16; ++j) x[j] = v4u{y8[2 * j].x, y8[2 * j].y, y8[2 * j + 1].x, y8[2 * j + 1].y}; }
            const uint32_t cp = tid & 7, q = tid >> 3;
            const uint32_t vo = (q * 1024 + cp * 2) * 8;
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_amdgcn_raw_buffer_store_b128(x[j], rout, vo, tile * 128 + j * 524288, NT);
        }
    }
}

typedef void (*kern_t)(const char *, char *, char *, uint32_t, uint32_t);
struct Variant { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 1024;          // transforms (8 MiB each)
    const uint32_t ring_slots = argc > 2 ? atoi(argv[2]) : 16;
    char *a, *b, *ring;
    CK(hipMalloc(&a, (size_t)batch * TB)); CK(hipMalloc(&b, (size_t)batch * TB)); CK(hipMalloc(&ring, (size_t)ring_slots * TB));
    CK(hipMemset(a, 1, (size_t)batch * TB)); CK(hipMemset(b, 1, (size_t)batch * TB)); CK(hipMemset(ring, 1, (size_t)ring_slots * TB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Variant vs[] = {
        {"hbm 8B, ring 8B strided  (current)   2wg/cu", k_tiles<8, 8, false, 4>},
        {"hbm 16B, ring 8B strided             2wg/cu", k_tiles<16, 8, false, 4>},
        {"hbm 8B, ring 16B strided             2wg/cu", k_tiles<8, 16, false, 4>},
        {"hbm 16B, ring 16B strided            2wg/cu", k_tiles<16, 16, false, 4>},
        {"hbm 8B, ring 8B tile-contiguous      2wg/cu", k_tiles<8, 8, true, 4>},
        {"hbm 16B, ring 16B tile-contiguous    2wg/cu", k_tiles<16, 16, true, 4>},
        {"hbm 8B, ring 8B strided              3wg/cu", k_tiles<8, 8, false, 6>},
        {"hbm 16B, ring 16B strided            3wg/cu", k_tiles<16, 16, false, 6>},
        {"hbm 16B, ring 16B tile-contiguous    3wg/cu", k_tiles<16, 16, true, 6>},
        {"hbm 16B, ring 16B tile-contiguous    4wg/cu", k_tiles<16, 16, true, 8>},
    };
    printf("batch %u transforms, ring %u slots (%u MiB); time scaled to batch 4096\n", batch, ring_slots, ring_slots * 8);
    for (auto &v : vs) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(batch * 128), dim3(512), 0, 0, a, b, ring, ring_slots, batch);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = 4.0 * batch * TB;  // hbm read + ring write + ring read + hbm write
        printf("%-48s %8.3f ms  %7.0f GB/s total  -> %6.2f ms at batch 4096 (%5.1f%% of 8 TB/s roofline)\n", v.name, best,
               bytes / (best * 1e-3) / 1e9, best * 4096.0 / batch, 100.0 * (16.0 * batch * (1 << 20)) / (best * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
