// kernels_small32.hip -- one-launch kernels for n = 512 .. 32768, 32 points per thread: launcher and the 512 .. 4096
// instantiations (template: small32_kernel.h; 8192 .. 32768: kernels_small32b.hip).
#include "small32_kernel.h"

namespace fwa {

hipError_t launch_small32(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                          hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    uint32_t lg_n = 0;
    while ((1u << lg_n) < n) ++lg_n;
#define FWA_S32(L)                                                                               \
    case L:                                                                                      \
        return dir == FWD ? launch_small32_n<L, FWD>(src, dst, tw, batch, scale, st)             \
                          : launch_small32_n<L, INV>(src, dst, tw, batch, scale, st)
    switch (lg_n) {
        FWA_S32(9); FWA_S32(10); FWA_S32(11); FWA_S32(12);
        case 13: case 14: case 15: return launch_small32_big(dir, lg_n, src, dst, tw, batch, scale, st);
        default: return hipErrorInvalidValue;
    }
#undef FWA_S32
}

}  // namespace fwa
