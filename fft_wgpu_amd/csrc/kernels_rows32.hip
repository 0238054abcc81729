// kernels_rows32.hip -- k_rows32, the last pass of the two-pass plans (512 .. 4096-point rows with the transposed store):
// launcher and the 512 / 1024-point instantiations (template: rows32.h; 2048 / 4096: kernels_rows32b.hip).
#include "rows32.h"

namespace fwa {

bool rows32_supported(uint32_t lg_l) { return lg_l >= 9 && lg_l <= 12; }
// tile-contiguous ring input of width in_cw (written by k_colsw): rows of >= 32*in_cw points
bool rows32_ring_supported(uint32_t lg_l, uint32_t in_cw) { return (in_cw == 32 || in_cw == 64) && lg_l >= 10 && lg_l <= 12 && (1u << (lg_l - 5)) >= in_cw; }

const void *rows32_kernel_small(uint32_t lg_l, int dir, uint32_t in_cw)
{
    return lg_l == 9 ? rows32_kernel_of<9>(dir, 0) : rows32_kernel_of<10>(dir, in_cw);
}
static const void *rows32_pick(uint32_t lg_l, int dir, uint32_t in_cw)
{
    if (lg_l == 9 || lg_l == 10) return rows32_kernel_small(lg_l, dir, in_cw);
    if (lg_l == 11 || lg_l == 12) return rows32_kernel_big(lg_l, dir, in_cw);
    return nullptr;
}
static void rows32_geometry(uint32_t lg_l, uint32_t *rw, uint32_t *threads, int *lds)
{
    *rw = (uint32_t)rows32_rows((int)lg_l);
    *threads = *rw << (lg_l - 5);
    switch (lg_l) {
        case 9: *lds = Rows32<9, rows32_rows(9)>::LDS_BYTES; break;
        case 10: *lds = Rows32<10, rows32_rows(10)>::LDS_BYTES; break;
        case 11: *lds = Rows32<11, rows32_rows(11)>::LDS_BYTES; break;
        default: *lds = Rows32<12, rows32_rows(12)>::LDS_BYTES; break;
    }
}

// > 64 KiB of dynamic LDS needs the attribute once per device (plan setup)
hipError_t prepare_rows32(uint32_t lg_l)
{
    if (!rows32_supported(lg_l)) return hipErrorInvalidValue;
    uint32_t rw, th;
    int lds;
    rows32_geometry(lg_l, &rw, &th, &lds);
    hipError_t e = hipSuccess;
    for (int dir : {FWD, INV})
        for (uint32_t cw : {0u, 32u, 64u})
            if (e == hipSuccess && (cw == 0 || rows32_ring_supported(lg_l, cw)))
                e = hipFuncSetAttribute(rows32_pick(lg_l, dir, cw), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}

// n = n1 * 2^lg_l per transform, n * 8 < 2^32; `in`: n1 rows of 2^lg_l contiguous samples (in_cw = 0) or the
// tile-contiguous ring of k_colsw (in_cw = its tile width); out[k1 + n1*k2]
hipError_t launch_rows32(int dir, uint32_t lg_l, const v2f *in, v2f *out, const v2f *tw, uint32_t n1, uint64_t in_sb,
                         uint64_t out_sb, uint32_t n_transforms, float scale, uint32_t xcd_swizzle, uint32_t in_cw,
                         hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (!rows32_supported(lg_l) || n1 < 16 || (n1 & (n1 - 1)) || ((uint64_t)n1 << lg_l) > (1ull << 28))
        return hipErrorInvalidValue;
    if (in_cw && !rows32_ring_supported(lg_l, in_cw)) return hipErrorInvalidValue;
    uint32_t rw, th;
    int lds;
    rows32_geometry(lg_l, &rw, &th, &lds);
    const uint64_t blocks = (uint64_t)n_transforms * (n1 / rw);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (blocks % 8) xcd_swizzle = 0;
    void *args[] = {&in, &out, &tw, &n1, &in_sb, &out_sb, &scale, &xcd_swizzle};
    return hipLaunchKernel(rows32_pick(lg_l, dir, in_cw), dim3((uint32_t)blocks), dim3(th), args, (size_t)lds, st);
}

}  // namespace fwa
