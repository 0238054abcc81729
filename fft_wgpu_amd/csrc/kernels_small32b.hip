// kernels_small32b.hip -- the 8192 .. 32768-point instantiations of k_small32 (small32_kernel.h), a translation unit of
// their own so that the library builds in parallel.
#include "small32_kernel.h"

namespace fwa {

hipError_t launch_small32_big(int dir, uint32_t lg_n, const v2f *src, v2f *dst, const v2f *tw, uint64_t batch, float scale,
                              hipStream_t st)
{
#define FWA_S32(L)                                                                               \
    case L:                                                                                      \
        return dir == FWD ? launch_small32_n<L, FWD>(src, dst, tw, batch, scale, st)             \
                          : launch_small32_n<L, INV>(src, dst, tw, batch, scale, st)
    switch (lg_n) {
        FWA_S32(13); FWA_S32(14); FWA_S32(15);
        default: return hipErrorInvalidValue;
    }
#undef FWA_S32
}

hipError_t setup_small_kernels()
{
    // 16384 / 32768-point transforms need 66 / 132 KiB of dynamic LDS (8192: 33 KiB, inside the default limit)
    hipError_t e = hipSuccess;
    const void *ks[4] = {reinterpret_cast<const void *>(&k_small32<14, FWD>), reinterpret_cast<const void *>(&k_small32<14, INV>),
                         reinterpret_cast<const void *>(&k_small32<15, FWD>), reinterpret_cast<const void *>(&k_small32<15, INV>)};
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)small32_lds(i < 2 ? 14 : 15));
    return e;
}

}  // namespace fwa
