// kernels.h -- launch wrappers implemented in kernels_*.hip (internal to the library).
#pragma once
#include "cplx.h"

namespace fwa {

hipError_t launch_r2_stage(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint32_t stage,
                           uint64_t batch, float scale, hipStream_t st);
// 2 <= n <= 256: contiguous 64-KiB chunks per workgroup, every wave walks 16 KiB linearly, operands staged in LDS half by half
// (kernels_chunk.hip: k_chunk); in place allowed
hipError_t launch_chunk(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                        hipStream_t st);
// 512 <= n <= 32768: 32 points per thread, one exchange (512, 1024) or two (kernels_small.hip: k_small32); in place allowed
hipError_t launch_small32(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                          hipStream_t st);
// last pass of a two-pass plan n = n1 * 2^lg_l (lg_l = 9 .. 12, n <= 2^28): 16 adjacent rows per workgroup (8 from 2048-point rows), 32 points
// per thread, transposed store out[k1 + n1*k2] (kernels_rows32.hip: k_rows32); tw = half table of W_{2^lg_l}
bool rows32_supported(uint32_t lg_l);
hipError_t prepare_rows32(uint32_t lg_l);
bool rows32_ring_supported(uint32_t lg_l, uint32_t in_cw);
// in_cw = 0: `in` is the n1 x 2^lg_l matrix; in_cw = 32 / 64: the tile-contiguous ring written by k_colsw
hipError_t launch_rows32(int dir, uint32_t lg_l, const v2f *in, v2f *out, const v2f *tw, uint32_t n1, uint64_t in_sb,
                         uint64_t out_sb, uint32_t n_transforms, float scale, uint32_t xcd_swizzle, uint32_t in_cw,
                         hipStream_t st);
// pass A with short columns and wide tiles (lg_l = 9: 512 rows x 32 columns, lg_l = 8: 256 x 64; 512 threads, two
// workgroups per CU; kernels_rows32.hip: k_colsw); tile_ring: tile-contiguous output [tile][k1][width]
bool colsw_supported(uint32_t lg_l);
uint32_t colsw_width(uint32_t lg_l);
hipError_t prepare_colsw(uint32_t lg_l);
hipError_t launch_colsw(int dir, uint32_t lg_l, bool out_is_ring, bool tile_ring, const v2f *in, v2f *out, const v2f *tw,
                        const v2f *tw_lo, const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb,
                        uint32_t n_transforms, uint32_t xcd_swizzle, hipStream_t st);
// pass A with a 2048-point first factor (lg_l = 11, n = 2048 * pitch <= 2^28): 16 adjacent columns
// per workgroup, matrix layout out, four-step twiddle of domain n (kernels_rows32.hip: k_cols32); tw = half table of W_{2^lg_l}
bool cols32_supported(uint32_t lg_l);
hipError_t prepare_cols32(uint32_t lg_l);
hipError_t launch_cols32(int dir, uint32_t lg_l, bool out_is_ring, const v2f *in, v2f *out, const v2f *tw, const v2f *tw_lo,
                         const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb, uint32_t n_transforms,
                         uint32_t xcd_swizzle, hipStream_t st);
enum { TILE_COLS = 0, TILE_ROWS_T = 1 };
enum { ROLE_FIRST = 1, ROLE_MIDDLE = 2, ROLE_LAST = 3 };  // cache-policy role of a tiled pass

struct TileArgs {
    const v2f *in;
    v2f *out;
    const v2f *tw;      // W_L table (L/2 entries) for the inner stages
    const v2f *tw_lo;   // four-step twiddle tables (COLS only)
    const v2f *tw_hi;
    uint64_t in_sb, in_s1, in_st;     // input base = b*in_sb + d1*in_s1 + tile*in_st
    uint64_t out_sb, out_s1, out_st;  // output base likewise
    uint64_t pitch;                   // COLS: element pitch of the FFT axis; ROWS_T: pitch between the rows of a tile
    uint64_t out_stride;              // ROWS_T: element stride between consecutive outputs of a row
    uint32_t d1_count, tile_count;    // blockIdx.x = (b*d1_count + d1)*tile_count + tile
    float scale;
    uint32_t role;                    // ROLE_FIRST (user buffer -> ring), ROLE_MIDDLE (ring -> ring), ROLE_LAST
    uint32_t cw;                      // FFTs per workgroup (tile width): 16 or 32
    uint32_t xcd_swizzle;             // XCD-aware block -> tile mapping (blocks % 8 == 0 only)
};

// `cw` FFTs of length 2^lg_l per workgroup along one axis (kernels_tiled.hip: k_tile); blocks = batch*d1_count*tile_count
bool tile_supported(uint32_t lg_l, uint32_t cw);
hipError_t prepare_tile(uint32_t lg_l, uint32_t cw);
hipError_t launch_tile(int dir, int mode, uint32_t lg_l, const TileArgs &a, uint64_t batch, hipStream_t st);
hipError_t setup_small_kernels();
hipError_t setup_1m_kernels();
// One pass of the 2^20 pipeline over `n_transforms` transforms; transform i of the launch uses ring slot i
// (1024 x 16-column tiles: 512-thread workgroups, 128-B segments).
hipError_t launch_p1_1m(int dir, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t n_transforms, uint32_t xcd_swizzle, hipStream_t st);
hipError_t launch_p2_1m(int dir, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t n_transforms,
                        float scale, uint32_t xcd_swizzle, hipStream_t st);
// 1024-point column pass for n = 1024 * pitch (pitch = 2^4 .. 2^20 columns), matrix layout in and out, four-step
// twiddle of domain n from the two-level table (tw_lo, tw_hi); transform i at src + i*in_sb / dst + i*out_sb.
hipError_t launch_p1_gen(int dir, bool out_is_ring, const v2f *src, v2f *dst, const v2f *tw_inner, const v2f *tw_lo,
                         const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb, uint32_t n_transforms,
                         uint32_t xcd_swizzle, hipStream_t st);
hipError_t launch_scale(const v2f *a, v2f *b, uint64_t n_samples, float scale, hipStream_t st);
hipError_t launch_fill(v2f *dst, uint64_t seed, uint64_t g0, uint64_t n_samples, float scale, hipStream_t st);
hipError_t launch_copy(const void *src, void *dst, uint64_t bytes, hipStream_t st);
// `blocks` one-wave workgroups that spin for ticks x 10 ns (<= 1 ms) without touching memory
hipError_t launch_spin(uint32_t ticks, uint32_t blocks, hipStream_t st);

#ifdef FWA_LAB
// ---- laboratory build only (kernels_lab_*.hip): kernel families that measured slower than the shipped ones ----
// 16 <= n <= 4096: register radix-16 Stockham with direct addressing (one launch); src == dst allowed (a transform is read
// completely before any of it is written)
// wave_shuffle: n = 32/64/128 exchange between the two stages with __shfl_xor instead of LDS (opt-in, slower)
hipError_t launch_small16(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                          bool wave_shuffle, hipStream_t st);
// Persistent form of the 2^20 pipeline: one launch per exec, ring of `ring_slots` transforms (>= depth + 1); `ctl` =
// ring_ctl_bytes(batch) bytes of device memory (zeroed here per call); ctl[1] != 0 afterwards means a bounded spin timed out.
size_t ring_ctl_bytes(uint64_t batch);
hipError_t launch_ring_1m(int dir, const v2f *src, v2f *dst, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                          uint32_t *ctl, uint32_t batch, uint32_t depth, uint32_t ring_slots, uint32_t n_workgroups,
                          float scale, hipStream_t st);
hipError_t setup_lab_1m_kernels();
#endif

}  // namespace fwa
