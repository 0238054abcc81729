// kernels_rows32b.hip -- the 2048-point instantiations of k_rows32 (rows32.h), a translation unit of their own so that the
// library builds in parallel (4096: kernels_rows32c.hip).
#include "rows32.h"

namespace fwa {

const void *rows32_kernel_2048(int dir, uint32_t in_cw) { return rows32_kernel_of<11>(dir, in_cw); }
const void *rows32_kernel_big(uint32_t lg_l, int dir, uint32_t in_cw)
{
    return lg_l == 11 ? rows32_kernel_2048(dir, in_cw) : rows32_kernel_4096(dir, in_cw);
}

}  // namespace fwa
