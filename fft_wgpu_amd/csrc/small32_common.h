// small32_common.h -- twiddle helper shared by the 32-point-per-thread kernels (kernels_small32.hip, kernels_rows32.hip).
#pragma once
#include "device_common.h"

namespace fwa {

// x[brev<R>(q)] *= W_N^{e*q} for q = 1 .. R-1 with 7 + R/8 - 1 table look-ups instead of R - 1:
// W^{e(8a + b)} = W^{8ea} * W^{eb} (one extra rounding on the twiddles that are products, as in k_tile).
// In two halves so that a kernel can issue the look-ups long before it needs them (they depend on the thread index only):
// issued at the point of use, behind the data loads, each costs its wave an exposed cache latency.
template <int R, int N>
struct Twiddles {
    v2f pb[8], pa[R / 8];
};
template <int R, int N>
__device__ __forceinline__ void twiddle_fetch(Twiddles<R, N> &w, const v2f *__restrict__ tw, uint32_t e)
{
    static_assert(R == 16 || R == 32, "radix");
    static_for<1, 8>([&](auto b_) { constexpr int b = decltype(b_)::value; w.pb[b] = tw_lookup<N>(tw, e * b); });
    static_for<1, R / 8>([&](auto a_) { constexpr int a = decltype(a_)::value; w.pa[a] = tw_lookup<N>(tw, e * (8 * a)); });
}
template <int R, int N, int DIR>
__device__ __forceinline__ void twiddle_apply(v2f (&x)[R], const Twiddles<R, N> &w)
{
    static_for<1, R>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        constexpr int a = q / 8, b = q % 8, r = brev<R>(q);
        if constexpr (a == 0) x[r] = cmul_tw<DIR>(x[r], w.pb[b]);
        else if constexpr (b == 0) x[r] = cmul_tw<DIR>(x[r], w.pa[a]);
        else x[r] = cmul_tw<DIR>(x[r], cmul(w.pa[a], w.pb[b]));
    });
}
template <int R, int N, int DIR>
__device__ __forceinline__ void twiddle_outputs(v2f (&x)[R], const v2f *__restrict__ tw, uint32_t e)
{
    Twiddles<R, N> w;
    twiddle_fetch<R, N>(w, tw, e);
    twiddle_apply<R, N, DIR>(x, w);
}

}  // namespace fwa
