// kernels.h -- launch wrappers implemented in kernels_{small,tiled,1m}.hip (internal to the library).
#pragma once
#include "cplx.h"

namespace fwa {

hipError_t launch_r2_stage(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint32_t stage,
                           uint64_t batch, float scale, hipStream_t st);
hipError_t launch_lds_small(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                            hipStream_t st);
// 16 <= n <= 16384: register radix-16 Stockham (one launch); src == dst allowed (a transform is read
// completely before any of it is written)
// wave_shuffle: n = 32/64/128 exchange between the two stages with __shfl_xor instead of LDS (opt-in, slower)
hipError_t launch_small16(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                          bool wave_shuffle, hipStream_t st);
enum { TILE_COLS = 0, TILE_ROWS_T = 1 };

struct TileArgs {
    const v2f *in;
    v2f *out;
    const v2f *tw;      // W_L table (L/2 entries) for the inner stages
    const v2f *tw_lo;   // four-step twiddle tables (COLS only)
    const v2f *tw_hi;
    uint64_t in_sb, in_s1, in_st;     // input base = b*in_sb + d1*in_s1 + tile*in_st
    uint64_t out_sb, out_s1, out_st;  // output base likewise
    uint64_t pitch;                   // COLS: element pitch of the FFT axis; ROWS_T: pitch between the 16 rows
    uint64_t out_stride;              // ROWS_T: element stride between consecutive outputs of a row
    uint32_t d1_count, tile_count;    // blockIdx.x = (b*d1_count + d1)*tile_count + tile
    float scale;
    uint32_t flags;                   // bit 0: timing-only ablation (twiddles = 1); bits 8-9: cache policy role
                                      // (0 default, 1 first pass, 2 middle pass, 3 last pass)
};

// 16 FFTs of length 2^lg_l per workgroup along one axis (kernels_tiled.hip: k_tile16); blocks = batch*d1_count*tile_count
hipError_t prepare_tile16(uint32_t lg_l);
hipError_t launch_tile16(int dir, int mode, uint32_t lg_l, const TileArgs &a, uint64_t batch, hipStream_t st);
// n = 2, 4, 8 (in place allowed)
hipError_t launch_tiny(int dir, const v2f *src, v2f *dst, uint32_t n, uint64_t batch, float scale, hipStream_t st);
hipError_t setup_small_kernels();
hipError_t setup_1m_kernels();
hipError_t launch_p1_1m(int dir, int policy, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t ring_slots, uint64_t t_first, uint32_t n_transforms, hipStream_t st);
hipError_t launch_p2_1m(int dir, int policy, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t ring_slots,
                        uint64_t t_first, uint32_t n_transforms, float scale, hipStream_t st);
size_t fused_ctl_bytes(uint64_t batch);
// In-place persistent pipeline; `ctl` = fused_ctl_bytes(batch) bytes of device memory (zeroed here per call).
// Needs >= 64 resident workgroups to be deadlock-free (see kernels_1m.hip); batch*128 tickets must fit in u32.
hipError_t launch_fused_1m(int dir, int policy, v2f *data, const v2f *tw_inner, const v2f *tw_outer, uint32_t *ctl,
                           uint32_t batch, uint32_t depth, uint32_t n_workgroups, float scale, uint32_t dbg,
                           hipStream_t st);
// One launch = pass-1 tiles of n1 transforms (p1_src -> p1_ring) side by side with pass-2 tiles of n2
// transforms (p2_ring -> p2_dst); the two sets are independent of each other.
hipError_t launch_mix_1m(int dir, int policy, const v2f *p1_src, v2f *p1_ring, uint32_t n1, const v2f *p2_ring,
                         v2f *p2_dst, uint32_t n2, const v2f *tw_inner, const v2f *tw_outer, float scale,
                         uint32_t dbg, hipStream_t st);
// n = R * S split: radix-R butterflies over stride S with twiddle W_(R*S)^{n2 k1} = hi[e>>10]*lo[e&1023]; `n_sub`
// independent arrays of length R*S; in == out allowed (each thread owns its R positions).
hipError_t launch_radix_pass(int dir, int R, const v2f *in, v2f *out, const v2f *tw_lo, const v2f *tw_hi,
                             uint32_t lg_s, uint64_t n_sub, hipStream_t st);
// digit-reversal permute of `batch` transforms of length R1*R2*M (out of place).
hipError_t launch_permute(const v2f *in, v2f *out, uint32_t lg_r1, uint32_t lg_r2, uint32_t lg_m, uint64_t batch,
                          float scale, hipStream_t st);
hipError_t launch_scale(const v2f *a, v2f *b, uint64_t n_samples, float scale, hipStream_t st);
hipError_t launch_fill(v2f *dst, uint64_t seed, uint64_t g0, uint64_t n_samples, float scale, hipStream_t st);
hipError_t launch_copy(const void *src, void *dst, uint64_t bytes, hipStream_t st);

}  // namespace fwa
