// tile_kernel.h -- k_tile, the generic 16-point-per-thread tile kernel of the multi-pass paths, as a template shared by its two
// translation units (kernels_tiled.hip: launcher + forward; kernels_tiled_inv.hip: inverse).
#pragma once
#include "tile_body.h"

namespace fwa {

template <int LGL, int CW, int DIR, int MODE, bool BUF, int ROLE>
__global__ __launch_bounds__(((1 << LGL) / 16) * CW) void k_tile(TileArgs a)
{
    constexpr int AOUT = (ROLE == ROLE_FIRST || ROLE == ROLE_MIDDLE) ? AUX_SC1 : (ROLE == ROLE_LAST ? AUX_NT : AUX_DEFAULT);
    constexpr int AIN = (ROLE == ROLE_FIRST) ? AUX_NT : AUX_DEFAULT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware block -> tile mapping, bit 0 only: each XCD gets a contiguous run of tiles (xcd_map, device_common.h).  Bit 2
    // (adjacent tiles for the two residents of a CU) means nothing for these small workgroups and is ignored, as the
    // header's description of "xcd_swizzle" says: values without bit 0 leave every kernel of the plan unswizzled.
    const uint32_t bid = xcd_map(a.xcd_swizzle & 1u);
    const uint32_t tile = bid % a.tile_count;
    const uint32_t rest = bid / a.tile_count;
    const uint32_t d1 = rest % a.d1_count;
    const uint64_t b = rest / a.d1_count;
    const v2f *in = a.in + b * a.in_sb + d1 * a.in_s1 + tile * a.in_st;
    v2f *out = a.out + b * a.out_sb + d1 * a.out_s1 + tile * a.out_st;
    tile_body<LGL, CW, DIR, MODE, BUF, AIN, AOUT>(in, out, tile * CW, a.tw, a.tw_lo, a.tw_hi, a.pitch, a.out_stride, a.scale,
                                                   reinterpret_cast<v2f *>(smem), threadIdx.x);
}

template <int CW, int DIR, int MODE, bool BUF, int ROLE>
static const void *tile_kernel_p(uint32_t lg_l)
{
    switch (lg_l) {
        case 6: return reinterpret_cast<const void *>(&k_tile<6, CW, DIR, MODE, BUF, ROLE>);
        case 7: return reinterpret_cast<const void *>(&k_tile<7, CW, DIR, MODE, BUF, ROLE>);
        case 8: return reinterpret_cast<const void *>(&k_tile<8, CW, DIR, MODE, BUF, ROLE>);
        case 9: return reinterpret_cast<const void *>(&k_tile<9, CW, DIR, MODE, BUF, ROLE>);
        case 10: return reinterpret_cast<const void *>(&k_tile<10, CW, DIR, MODE, BUF, ROLE>);
        default: return nullptr;
    }
}
// COLS passes come as first or middle pass, ROWS_T is always the last; the 64-bit-pointer form (BUF = false,
// only above 4-GiB tiles) has no policy bits.
template <int CW, int DIR, int MODE>
static const void *tile_kernel_m(uint32_t lg_l, bool buf, int role)
{
    if (!buf) return tile_kernel_p<CW, DIR, MODE, false, 0>(lg_l);
    if constexpr (MODE == TILE_COLS) {
        if (role == ROLE_MIDDLE) return tile_kernel_p<CW, DIR, MODE, true, ROLE_MIDDLE>(lg_l);
        return tile_kernel_p<CW, DIR, MODE, true, ROLE_FIRST>(lg_l);
    } else {
        return tile_kernel_p<CW, DIR, MODE, true, ROLE_LAST>(lg_l);
    }
}
// entry points of one direction (dir = FWD: kernels_tiled.hip, INV: kernels_tiled_inv.hip)
const void *tile_kernel_fwd(int mode, uint32_t lg_l, bool buf, int role);
const void *tile_kernel_inv(int mode, uint32_t lg_l, bool buf, int role);

}  // namespace fwa
