// kernels_tiled_inv.hip -- the inverse-direction instantiations of k_tile (tile_kernel.h), a translation unit of their own so
// that the library builds in parallel.
#include "tile_kernel.h"

namespace fwa {

const void *tile_kernel_inv(int mode, uint32_t lg_l, bool buf, int role)
{
    return mode == TILE_COLS ? tile_kernel_m<16, INV, TILE_COLS>(lg_l, buf, role) : tile_kernel_m<16, INV, TILE_ROWS_T>(lg_l, buf, role);
}

}  // namespace fwa
