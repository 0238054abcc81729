/*
 * oracle/ref_fft.c -- TEST INFRASTRUCTURE ONLY (see ref_fft.h).
 *
 * Build with -ffp-contract=off so that the complex multiply keeps the
 * reference's 4-mul / 2-add form (src/kernel/fft.wgsl:71-73).
 */
#define _POSIX_C_SOURCE 199309L
#include "ref_fft.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* generator                                                           */
/* ------------------------------------------------------------------ */
static inline uint64_t fwo_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline fwo_c32 fwo_sample(uint64_t seed, uint64_t g, float scale)
{
    uint64_t h = fwo_mix64(seed + (g + 1) * 0x9E3779B97F4A7C15ULL);
    int32_t r = (int32_t)((h >> 40) & 0xFFFFFFu) - 8388608;
    int32_t i = (int32_t)((h >> 16) & 0xFFFFFFu) - 8388608;
    fwo_c32 v;
    v.re = ((float)r * 0x1p-23f) * scale;
    v.im = ((float)i * 0x1p-23f) * scale;
    return v;
}

void fwo_gen_input(fwo_c32 *dst, uint64_t seed, uint64_t first_transform,
                   uint64_t n_transforms, uint32_t n, float scale)
{
    uint64_t total = n_transforms * (uint64_t)n;
    uint64_t g0 = first_transform * (uint64_t)n;
    for (uint64_t k = 0; k < total; ++k)
        dst[k] = fwo_sample(seed, g0 + k, scale);
}

int fwo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* fp32 restatement of the reference                                   */
/* ------------------------------------------------------------------ */

/* src/kernel/fft.wgsl:71-73 (same text in fft4.wgsl:114-119, ifft.wgsl:77-79) */
static inline fwo_c32 cmul(fwo_c32 a, fwo_c32 b)
{
    fwo_c32 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

static int ilog2_u32(uint32_t n)
{
    int l = 0;
    while ((1u << l) < n) ++l;
    return l;
}

/* Host twiddle table: src/processor.rs:43-49.  f64 math, rounded to f32. */
static fwo_c32 *forward_table(uint32_t n)
{
    const double PI = 3.14159265358979323846; /* std::f64::consts::PI */
    uint32_t half = n / 2 ? n / 2 : 1;
    fwo_c32 *tw = (fwo_c32 *)malloc(sizeof(fwo_c32) * half);
    for (uint32_t k = 0; k < n / 2; ++k) {
        double theta = -2.0 * PI * (double)k / (double)n;
        tw[k].re = (float)cos(theta);
        tw[k].im = (float)sin(theta);
    }
    return tw;
}

enum { MODE_FWD = 0, MODE_INV_SCALED = 1, MODE_INV_UNSCALED = 2 };

/* One transform, all log2(n) stages.  Stage recurrence:
 *   src/kernel/fft.wgsl:27-62  (forward, table twiddle at index block_idx*J)
 *   src/kernel/ifft.wgsl:25-64 (inverse, f32 cos/sin of 2*PI*f32(s*J)/f32(n))
 *   src/kernel/ifft.wgsl:65-74 (last stage divides its two outputs by f32(n))
 * Even stages read `a` and write `b`, odd stages the reverse. */
static void one_transform(fwo_c32 *a, fwo_c32 *b, uint32_t n, int rounds, int mode,
                          const fwo_c32 *tw)
{
    const float PI_F = 3.14159265358979323846f; /* ifft.wgsl:6 */
    const uint32_t half = n / 2;
    for (int stage = 0; stage < rounds; ++stage) {
        const uint32_t J = 1u << stage;
        const uint32_t block_size = 2u * J;
        fwo_c32 *src = (stage % 2 == 0) ? a : b;
        fwo_c32 *dst = (stage % 2 == 0) ? b : a;
        for (uint32_t idx = 0; idx < half; ++idx) {
            uint32_t s = idx / J;
            uint32_t j = idx % J;
            fwo_c32 w;
            if (mode == MODE_FWD) {
                w = tw[s * J];
            } else {
                float theta = 2.0f * PI_F * (float)(s * J) / (float)n;
                w.re = cosf(theta);
                w.im = sinf(theta);
            }
            uint32_t idx1 = s * J + j;
            uint32_t idx2 = idx1 + half;
            uint32_t out1 = s * block_size + j;
            uint32_t out2 = out1 + J;
            fwo_c32 x = src[idx1], y = src[idx2];
            fwo_c32 sum = { x.re + y.re, x.im + y.im };
            fwo_c32 dif = { x.re - y.re, x.im - y.im };
            fwo_c32 p = cmul(dif, w);
            if (mode == MODE_INV_SCALED && stage == rounds - 1) {
                float fn = (float)n;
                sum.re = sum.re / fn; sum.im = sum.im / fn;
                p.re = p.re / fn;     p.im = p.im / fn;
            }
            dst[out1] = sum;
            dst[out2] = p;
        }
    }
}

static int run_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads, int mode)
{
    int rounds = ilog2_u32(n);
    fwo_c32 *tw = (mode == MODE_FWD) ? forward_table(n) : NULL;
    if (threads < 1) threads = 1;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t t = 0; t < (int64_t)batch; ++t)
        one_transform(a + (uint64_t)t * n, b + (uint64_t)t * n, n, rounds, mode, tw);
    free(tw);
    /* src/processor.rs:153-157 (Forward), :335-339 (Inverse), :664-668 (Onlyinverse) */
    return (rounds % 2 == 0) ? 0 : 1;
}

int fwo_forward_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads)
{ return run_ref(a, b, n, batch, threads, MODE_FWD); }

int fwo_inverse_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads)
{ return run_ref(a, b, n, batch, threads, MODE_INV_SCALED); }

int fwo_onlyinverse_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads)
{ return run_ref(a, b, n, batch, threads, MODE_INV_UNSCALED); }

void fwo_normalize_ref(const fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t total)
{
    float fn = (float)n;
    for (uint64_t i = 0; i < total; ++i) {
        b[i].re = a[i].re / fn;
        b[i].im = a[i].im / fn;
    }
}

/* ------------------------------------------------------------------ */
/* fp64 truth                                                          */
/* ------------------------------------------------------------------ */
void fwo_dft_f64_naive(const fwo_c32 *in, fwo_c64 *out, uint32_t n, int dir)
{
    const double TWO_PI = 6.283185307179586476925286766559;
    double *c = (double *)malloc(sizeof(double) * n);
    double *s = (double *)malloc(sizeof(double) * n);
    for (uint32_t k = 0; k < n; ++k) {
        /* octant-exact values where it matters */
        double ang = TWO_PI * (double)k / (double)n;
        c[k] = cos(ang);
        s[k] = (dir < 0 ? -1.0 : 1.0) * sin(ang);
    }
    for (uint32_t k = 0; k < n; ++k) {
        double re = 0.0, im = 0.0, cre = 0.0, cim = 0.0; /* Kahan */
        for (uint32_t j = 0; j < n; ++j) {
            uint32_t m = (uint32_t)(((uint64_t)j * k) % n);
            double xr = in[j].re, xi = in[j].im;
            double pr = xr * c[m] - xi * s[m];
            double pi = xr * s[m] + xi * c[m];
            double y = pr - cre, t = re + y; cre = (t - re) - y; re = t;
            y = pi - cim; t = im + y; cim = (t - im) - y; im = t;
        }
        out[k].re = re; out[k].im = im;
    }
    free(c); free(s);
}

static void dft64_radix2(const fwo_c32 *in, fwo_c64 *out, uint32_t n, int dir,
                         const fwo_c64 *tw /* n/2 entries exp(dir*2*pi*i*k/n) */)
{
    int lg = ilog2_u32(n);
    /* bit-reversed load, then iterative DIT */
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1u) << (lg - 1 - b);
        out[r].re = in[i].re; out[r].im = in[i].im;
    }
    (void)dir;
    for (uint32_t len = 2; len <= n; len <<= 1) {
        uint32_t step = n / len, h = len / 2;
        for (uint32_t base = 0; base < n; base += len) {
            for (uint32_t j = 0; j < h; ++j) {
                fwo_c64 w = tw[j * step];
                fwo_c64 u = out[base + j], v = out[base + j + h];
                double vr = v.re * w.re - v.im * w.im;
                double vi = v.re * w.im + v.im * w.re;
                out[base + j].re = u.re + vr;     out[base + j].im = u.im + vi;
                out[base + j + h].re = u.re - vr; out[base + j + h].im = u.im - vi;
            }
        }
    }
}

void fwo_dft_f64(const fwo_c32 *in, fwo_c64 *out, uint32_t n, uint64_t batch,
                 int dir, int threads)
{
    if (threads < 1) threads = 1;
    if (n <= 64) {
        for (uint64_t t = 0; t < batch; ++t)
            fwo_dft_f64_naive(in + t * n, out + t * n, n, dir);
        return;
    }
    const double TWO_PI = 6.283185307179586476925286766559;
    fwo_c64 *tw = (fwo_c64 *)malloc(sizeof(fwo_c64) * (n / 2));
    for (uint32_t k = 0; k < n / 2; ++k) {
        double ang = TWO_PI * (double)k / (double)n;
        tw[k].re = cos(ang);
        tw[k].im = (dir < 0 ? -1.0 : 1.0) * sin(ang);
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t t = 0; t < (int64_t)batch; ++t)
        dft64_radix2(in + (uint64_t)t * n, out + (uint64_t)t * n, n, dir, tw);
    free(tw);
}

void fwo_compare(const fwo_c32 *y, const fwo_c64 *r, uint32_t n,
                 double *max_rel, double *rel_l2)
{
    double maxd = 0.0, maxr = 0.0, sd = 0.0, sr = 0.0;
    for (uint32_t k = 0; k < n; ++k) {
        double dr = (double)y[k].re - r[k].re, di = (double)y[k].im - r[k].im;
        double d = sqrt(dr * dr + di * di);
        double m = sqrt(r[k].re * r[k].re + r[k].im * r[k].im);
        if (!(d <= maxd)) maxd = d;   /* NaN-propagating max */
        if (m > maxr) maxr = m;
        sd += d * d; sr += m * m;
    }
    *max_rel = (maxr > 0.0) ? maxd / maxr : maxd;
    *rel_l2 = (sr > 0.0) ? sqrt(sd / sr) : sqrt(sd);
}

/* ------------------------------------------------------------------ */
/* CPU baseline                                                        */
/* ------------------------------------------------------------------ */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double fwo_bench_forward(uint32_t n, uint64_t batch, int threads, double min_seconds,
                         int *reps_out)
{
    uint64_t total = (uint64_t)n * batch;
    fwo_c32 *a = (fwo_c32 *)malloc(sizeof(fwo_c32) * total);
    fwo_c32 *b = (fwo_c32 *)malloc(sizeof(fwo_c32) * total);
    if (!a || !b) { free(a); free(b); return 0.0; }
    double best = 1e300, t_all = 0.0;
    int reps = 0;
    do {
        fwo_gen_input(a, 0x5EEDULL, 0, batch, n, 1.0f);
        double t0 = now_s();
        fwo_forward_ref(a, b, n, batch, threads);
        double dt = now_s() - t0;
        if (dt < best) best = dt;
        t_all += dt;
        ++reps;
    } while (t_all < min_seconds);
    free(a); free(b);
    if (reps_out) *reps_out = reps;
    return (double)total / best;
}
