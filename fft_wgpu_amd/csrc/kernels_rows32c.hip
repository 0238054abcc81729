// kernels_rows32c.hip -- the 4096-point instantiations of k_rows32 (rows32.h).
#include "rows32.h"

namespace fwa {

const void *rows32_kernel_4096(int dir, uint32_t in_cw) { return rows32_kernel_of<12>(dir, in_cw); }

}  // namespace fwa
