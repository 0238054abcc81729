// tools/tile_probe3.hip -- measurement tool: where does the tile pattern lose bandwidth?  Isolates the four
// streams of the two-pass FFT (HBM column-tile read, ring write, ring read, HBM column-tile write) and
// their pairings, with the real tile shapes (16 cols x 1024 rows, 8 B/lane, tile-contiguous ring).
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile_probe3 tools/tile_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 2, SC1 = 16;
constexpr uint32_t TB = 8u << 20;

// mask bits: 1 HBM column-tile read, 2 ring write, 4 ring read, 8 HBM column-tile write.
// LINEAR_HBM: replace the column tile by a contiguous 128-KiB block of the same transform (pattern control).
template <bool LINEAR_HBM>
__global__ __launch_bounds__(512, 4) void k(const char *big_in, char *big_out, char *ring, uint32_t ring_slots,
                                            uint32_t batch, int mask, int both_roles, unsigned *sink)
{
    const uint32_t tid = threadIdx.x;
    uint32_t role, idx;
    if (both_roles) { role = blockIdx.x & 1; idx = blockIdx.x >> 1; } else { role = (mask & 3) ? 0 : 1; idx = blockIdx.x; }
    const uint32_t tile = idx & 63, t = idx >> 6;
    if (t >= batch) return;
    auto rring = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)(t % ring_slots) * TB, 0, TB, 0x00020000);
    const uint32_t c = tid & 15, q = tid >> 4;
    const uint32_t vo_col = LINEAR_HBM ? tid * 8 : (q * 1024 + c) * 8;
    const uint32_t so_col = LINEAR_HBM ? tile * 131072 : tile * 128;
    constexpr uint32_t jstep = LINEAR_HBM ? 4096 : 262144;
    v2u x[32];
    v2u acc = {0, 0};
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = v2u{tid + j, tile};
    if (role == 0) {
        if (mask & 1) {
            auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(big_in) + (size_t)t * TB, 0, TB, 0x00020000);
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, vo_col, so_col + j * jstep, NT);
        }
        if (mask & 2) {
            const uint32_t vs = (q * 16 + c) * 8;
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rring, vs, tile * 131072 + j * 4096, SC1);
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += x[j];
        }
    } else {
        if (mask & 4) {
            const uint32_t np = tid & 31, r = tid >> 5;
            const uint32_t vi = (np >> 4) * 131072 + r * 128 + (np & 15) * 8;
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(rring, vi, tile * 2048 + j * 262144, 0);
        }
        if (mask & 8) {
            auto rout = __builtin_amdgcn_make_buffer_rsrc(big_out + (size_t)t * TB, 0, TB, 0x00020000);
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rout, vo_col, so_col + j * jstep, NT);
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += x[j];
        }
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}

int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 1024, ring_slots = argc > 2 ? atoi(argv[2]) : 16;
    char *a, *b, *ring; unsigned *sink;
    CK(hipMalloc(&a, (size_t)batch * TB)); CK(hipMalloc(&b, (size_t)batch * TB)); CK(hipMalloc(&ring, (size_t)ring_slots * TB)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, (size_t)batch * TB)); CK(hipMemset(b, 1, (size_t)batch * TB)); CK(hipMemset(ring, 1, (size_t)ring_slots * TB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct M { int mask, both; const char *name; } ms[] = {
        {1, 0, "HBM read only (pass-1 loads)"},        {8, 0, "HBM write only (pass-2 stores)"},
        {2, 0, "ring write only (sc1)"},               {4, 0, "ring read only"},
        {3, 0, "pass 1 alone: HBM read + ring write"}, {12, 0, "pass 2 alone: ring read + HBM write"},
        {9, 1, "HBM read (p1 WGs) + HBM write (p2 WGs)"},
        {15, 1, "all four (the FFT mix)"},
    };
    for (int lin = 0; lin < 2; ++lin) {
        printf("---- HBM-side pattern: %s\n", lin ? "contiguous 128-KiB blocks (control)" : "column tiles (128-B segments, 8-KiB pitch)");
        for (auto &m : ms) {
            float best = 1e30f;
            const uint32_t grid = batch * 64 * (m.both ? 2 : 1);
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (lin) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(512), 0, 0, a, b, ring, ring_slots, batch, m.mask, m.both, sink);
                else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(512), 0, 0, a, b, ring, ring_slots, batch, m.mask, m.both, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (t < best) best = t;
            }
            const double bytes = (double)__builtin_popcount(m.mask) * batch * TB;
            printf("%-44s %8.3f ms  %7.0f GB/s\n", m.name, best, bytes / (best * 1e-3) / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
