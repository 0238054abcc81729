// kernels_tiled.hip -- the building block of the 2- and 3-pass paths: k_tile (template: tile_kernel.h); launcher and the
// forward instantiations (inverse: kernels_tiled_inv.hip).
#include "tile_kernel.h"

namespace fwa {

// Launchable tile widths: 16 FFTs per workgroup.  (32-wide tiles -- 256-byte segments, workgroups twice as large --
// measured no faster at any size, profiles/round2/sizes_cw16_vs_cw32.jsonl.)
bool tile_supported(uint32_t lg_l, uint32_t cw) { return cw == 16 && lg_l >= 6 && lg_l <= 10; }

const void *tile_kernel_fwd(int mode, uint32_t lg_l, bool buf, int role)
{
    return mode == TILE_COLS ? tile_kernel_m<16, FWD, TILE_COLS>(lg_l, buf, role) : tile_kernel_m<16, FWD, TILE_ROWS_T>(lg_l, buf, role);
}
static const void *tile_kernel(int dir, int mode, uint32_t cw, uint32_t lg_l, bool buf, int role)
{
    if (!tile_supported(lg_l, cw)) return nullptr;
    return dir == FWD ? tile_kernel_fwd(mode, lg_l, buf, role) : tile_kernel_inv(mode, lg_l, buf, role);
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
