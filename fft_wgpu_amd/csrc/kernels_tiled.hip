// kernels_tiled.hip -- (2/3) the building block of the 2- and 3-pass paths: k_tile.
#include "tile_body.h"

namespace fwa {

template <int LGL, int CW, int DIR, int MODE, bool BUF, int ROLE>
__global__ __launch_bounds__(((1 << LGL) / 16) * CW) void k_tile(TileArgs a)
{
    constexpr int AOUT = (ROLE == ROLE_FIRST || ROLE == ROLE_MIDDLE) ? AUX_SC1 : (ROLE == ROLE_LAST ? AUX_NT : AUX_DEFAULT);
    constexpr int AIN = (ROLE == ROLE_FIRST) ? AUX_NT : AUX_DEFAULT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware block -> tile mapping: each XCD gets a contiguous run of tiles (see kernels_1m.hip xcd_block)
    const uint32_t bid = a.xcd_swizzle ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t tile = bid % a.tile_count;
    const uint32_t rest = bid / a.tile_count;
    const uint32_t d1 = rest % a.d1_count;
    const uint64_t b = rest / a.d1_count;
    const v2f *in = a.in + b * a.in_sb + d1 * a.in_s1 + tile * a.in_st;
    v2f *out = a.out + b * a.out_sb + d1 * a.out_s1 + tile * a.out_st;
    tile_body<LGL, CW, DIR, MODE, BUF, AIN, AOUT>(in, out, tile * CW, a.tw, a.tw_lo, a.tw_hi, a.pitch, a.out_stride, a.scale,
                                                   reinterpret_cast<v2f *>(smem), threadIdx.x);
}

// Launchable tile widths: 16 FFTs per workgroup.  (32-wide tiles -- 256-byte segments, workgroups twice as large --
// measured no faster at any size, profiles/round2/sizes_cw16_vs_cw32.jsonl, and are only instantiated inside k_team.)
bool tile_supported(uint32_t lg_l, uint32_t cw) { return cw == 16 && lg_l >= 6 && lg_l <= 10; }

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
static const void *tile_kernel(int dir, int mode, uint32_t cw, uint32_t lg_l, bool buf, int role)
{
    if (!tile_supported(lg_l, cw)) return nullptr;
#define FWA_TK(CWV)                                                                                          \
    (dir == FWD ? (mode == TILE_COLS ? tile_kernel_m<CWV, FWD, TILE_COLS>(lg_l, buf, role)                   \
                                     : tile_kernel_m<CWV, FWD, TILE_ROWS_T>(lg_l, buf, role))                \
                : (mode == TILE_COLS ? tile_kernel_m<CWV, INV, TILE_COLS>(lg_l, buf, role)                   \
                                     : tile_kernel_m<CWV, INV, TILE_ROWS_T>(lg_l, buf, role)))
    return FWA_TK(16);
#undef FWA_TK
}

// called at plan creation: raises the dynamic-LDS limit of the kernels a plan will launch
hipError_t prepare_tile(uint32_t lg_l, uint32_t cw)
{
    if (!tile_supported(lg_l, cw)) return hipErrorInvalidValue;
    const size_t lds = tile_lds(lg_l, cw);
    if (lds <= 65536) return hipSuccess;
    for (int dir : {FWD, INV})
        for (int mode : {TILE_COLS, TILE_ROWS_T})
            for (int buf = 0; buf < 2; ++buf)
                for (int role : {ROLE_FIRST, ROLE_MIDDLE}) {
                    const void *k = tile_kernel(dir, mode, cw, lg_l, buf != 0, role);
                    if (!k) return hipErrorInvalidValue;
                    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                }
    return hipSuccess;
}

hipError_t launch_tile(int dir, int mode, uint32_t lg_l, const TileArgs &a, uint64_t batch, hipStream_t st)
{
    const uint64_t blocks = batch * a.d1_count * a.tile_count;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one tile?  COLS: L rows of `pitch`; ROWS_T: cw rows of `pitch` in, L outputs of out_stride
    const uint64_t L = 1ull << lg_l, cw = a.cw;
    const uint64_t span_in = (cw * a.pitch + L) * 8, span_out = (L * a.out_stride + cw) * 8;
    const uint64_t span = (mode == TILE_COLS) ? L * a.pitch * 8 + cw * 8 : (span_in > span_out ? span_in : span_out);
    const void *k = tile_kernel(dir, mode, a.cw, lg_l, span < (1ull << 32), (int)a.role);
    if (!k) return hipErrorInvalidValue;
    TileArgs copy = a;
    if (blocks % 8) copy.xcd_swizzle = 0;
    void *args[] = {&copy};
    return hipLaunchKernel(k, dim3((uint32_t)blocks), dim3((uint32_t)((L / 16) * cw)), args, tile_lds(lg_l, a.cw), st);
}


}  // namespace fwa
