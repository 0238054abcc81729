// cplx.h -- device-side complex helpers and the in-register radix-R FFT.
// gfx950 only.  No reference analogue: the reference is radix-2 everywhere
// (SURVEY.md F4); these replace R/2 * log2(R) of its butterflies
// (src/kernel/fft.wgsl:27-62) by one register-resident DIF network.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fwa {

typedef float v2f __attribute__((ext_vector_type(2)));  // {re, im}: layout of reference src/lib.rs:10-15
typedef float v4f __attribute__((ext_vector_type(4)));

// Forward transforms use exp(-2*pi*i*k/n) (reference processor.rs:43-49); the
// inverse uses the conjugate (ifft.wgsl:41-42).  Tables always hold the forward
// value; DIR = +1 conjugates on use.
constexpr int FWD = -1;
constexpr int INV = +1;

__device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
    return v2f{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f w)  // a * conj(w)
{
    return v2f{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y};
}
template <int DIR>
__device__ __forceinline__ v2f cmul_tw(v2f a, v2f w_fwd)
{
    if constexpr (DIR == FWD) return cmul(a, w_fwd);
    else return cmul_conj(a, w_fwd);
}

// ---- compile-time helpers ------------------------------------------------
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

template <int R>
constexpr int brev(int k)
{
    int r = 0;
    for (int b = 0; b < ilog2c(R); ++b) r |= ((k >> b) & 1) << (ilog2c(R) - 1 - b);
    return r;
}

// constexpr cos/sin(2*pi*j/m) for j/m in [0, 1/2], m a power of two: reduce to
// [0, pi/4] exactly in integers, then Taylor in double (error < 1e-17).
constexpr double cx_poly_sin(double x)
{
    double x2 = x * x, t = x, s = x;
    for (int i = 1; i < 12; ++i) { t *= -x2 / ((2 * i) * (2 * i + 1)); s += t; }
    return s;
}
constexpr double cx_poly_cos(double x)
{
    double x2 = x * x, t = 1.0, s = 1.0;
    for (int i = 1; i < 12; ++i) { t *= -x2 / ((2 * i - 1) * (2 * i)); s += t; }
    return s;
}
constexpr double CX_PI = 3.14159265358979323846264338327950288;
constexpr double cx_cos2pi(int j, int m)  // cos(2*pi*j/m), 0 <= j <= m/2
{
    // octants of width m/8
    if (8 * j <= m) return cx_poly_cos(2.0 * CX_PI * j / m);
    if (8 * j <= 3 * m) return -cx_poly_sin(2.0 * CX_PI * (4 * j - m) / (4.0 * m));  // cos(pi/2 + d) = -sin d
    return -cx_poly_cos(2.0 * CX_PI * (m - 2 * j) / (2.0 * m));                        // cos(pi - d) = -cos d
}
constexpr double cx_sin2pi(int j, int m)
{
    if (8 * j <= m) return cx_poly_sin(2.0 * CX_PI * j / m);
    if (8 * j <= 3 * m) return cx_poly_cos(2.0 * CX_PI * (4 * j - m) / (4.0 * m));    // sin(pi/2 + d) = cos d
    return cx_poly_sin(2.0 * CX_PI * (m - 2 * j) / (2.0 * m));                         // sin(pi - d) = sin d
}

// d * W_M^J, W_M = exp(DIR * 2*pi*i / M), 0 <= J < M/2.
template <int M, int J, int DIR>
__device__ __forceinline__ v2f tw_const(v2f d)
{
    constexpr float h = 0.70710678118654752440f;
    if constexpr (J == 0) {
        return d;
    } else if constexpr (4 * J == M) {  // -i (fwd) / +i (inv)
        return DIR == FWD ? v2f{d.y, -d.x} : v2f{-d.y, d.x};
    } else if constexpr (8 * J == M) {
        return DIR == FWD ? v2f{(d.x + d.y) * h, (d.y - d.x) * h}
                          : v2f{(d.x - d.y) * h, (d.x + d.y) * h};
    } else if constexpr (8 * J == 3 * M) {
        return DIR == FWD ? v2f{(d.y - d.x) * h, -(d.x + d.y) * h}
                          : v2f{-(d.x + d.y) * h, (d.x - d.y) * h};
    } else {
        constexpr float c = (float)cx_cos2pi(J, M);
        constexpr float s = (float)cx_sin2pi(J, M);
        // fwd: d * (c - i s); inv: d * (c + i s)
        return DIR == FWD ? v2f{d.x * c + d.y * s, d.y * c - d.x * s}
                          : v2f{d.x * c - d.y * s, d.y * c + d.x * s};
    }
}

// In-place radix-R decimation-in-frequency FFT on registers.
// Input natural order x[0..R-1]; output X[k] is left in x[brev<R>(k)].
template <int R, int DIR>
__device__ __forceinline__ void fft_reg(v2f (&x)[R])
{
    constexpr int LG = ilog2c(R);
    static_for<0, LG>([&](auto s_) {
        constexpr int half = R >> (decltype(s_)::value + 1);
        static_for<0, R / (2 * half)>([&](auto b_) {
            constexpr int base = decltype(b_)::value * 2 * half;
            static_for<0, half>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                v2f a = x[base + j], b = x[base + j + half];
                x[base + j] = a + b;
                x[base + j + half] = tw_const<2 * half, j, DIR>(a - b);
            });
        });
    });
}

// ---- generator, bit-identical to oracle/ref_fft.c:fwo_sample --------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ v2f gen_sample(uint64_t seed, uint64_t g, float scale)
{
    uint64_t h = mix64(seed + (g + 1) * 0x9E3779B97F4A7C15ULL);
    int32_t r = (int32_t)((h >> 40) & 0xFFFFFFu) - 8388608;
    int32_t i = (int32_t)((h >> 16) & 0xFFFFFFu) - 8388608;
    return v2f{((float)r * 0x1p-23f) * scale, ((float)i * 0x1p-23f) * scale};
}

}  // namespace fwa
