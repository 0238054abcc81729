// kernels_wave.hip -- n = 512: launcher and instantiations of the wave-private kernel (template and description: wave_kernel.h).
#include "wave_kernel.h"

namespace fwa {

hipError_t launch_wave512(int dir, const v2f *src, v2f *dst, const v2f *tw, uint64_t batch, float scale, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    const uint64_t n_samples = batch * 512, blocks = (n_samples + 8191) / 8192;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (dir == FWD) hipLaunchKernelGGL((k_wave512<FWD>), dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, tw, n_samples, scale);
    else hipLaunchKernelGGL((k_wave512<INV>), dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, tw, n_samples, scale);
    return hipGetLastError();
}

}  // namespace fwa
