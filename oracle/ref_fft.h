/*
 * oracle/ref_fft.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the fft_wgpu reference algorithm (radix-2 Stockham DIF,
 * ping-pong buffers) plus an independent fp64 DFT.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's `cpu_baseline` leg may load this.
 * The product path (fft_wgpu_amd/csrc) never links or calls it.
 *
 * Parity status: the reference's own tests pin only constant-input inverse
 * transforms at n=512 (examples/basic_inverse.rs:238-253,
 * examples/basic_inverse2.rs:269-284); those cases are checked in
 * tests/test_oracle.py.  The forward transform is NOT pinned by any reference
 * test ("forward parity unpinned by the reference"); it is anchored on the
 * mathematical DFT (fwo_dft_f64 / numpy.fft fixtures) which is what the
 * reference's comparison library (rustfft, unpinned "*", Cargo.toml:12) computes.
 */
#ifndef FWO_REF_FFT_H
#define FWO_REF_FFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Wire layout of one sample: reference src/lib.rs:10-15 (#[repr(C)] {real, imag}). */
typedef struct { float re, im; } fwo_c32;
typedef struct { double re, im; } fwo_c64;

/* Counter-based input generator shared bit-for-bit with the device generator
 * (fft_wgpu_amd/csrc/gen.hip).  Sample i of transform t (n samples each) is a
 * pure function of (seed, t*n + i): re, im uniform on the 2^-23 grid in
 * [-1, 1), then multiplied by `scale` (use a power of two to stay exact). */
void fwo_gen_input(fwo_c32 *dst, uint64_t seed, uint64_t first_transform,
                   uint64_t n_transforms, uint32_t n, float scale);

/* Forward, unnormalised.  Follows src/kernel/fft.wgsl:27-62 == fft4.wgsl:53-91
 * with the table twiddles of src/processor.rs:43-49.  `a` holds the input and
 * is clobbered; `b` is the ping-pong partner (same size).  Returns 0 when the
 * result is in `a`, 1 when it is in `b` (src/processor.rs:153-157). */
int fwo_forward_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads);

/* Inverse with the 1/n scale folded into the last stage: src/kernel/ifft.wgsl:25-75. */
int fwo_inverse_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads);

/* Inverse without the scale: src/kernel/onlyifft.wgsl:25-65. */
int fwo_onlyinverse_ref(fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t batch, int threads);

/* b[i] = a[i] / f32(n): src/kernel/normalize.wgsl:9-12. */
void fwo_normalize_ref(const fwo_c32 *a, fwo_c32 *b, uint32_t n, uint64_t total);

/* Independent truth: fp64 DFT of fp32 input.  dir = -1 forward, +1 inverse
 * (unscaled).  Naive O(n^2) for n <= 64, fp64 radix-2 otherwise. */
void fwo_dft_f64(const fwo_c32 *in, fwo_c64 *out, uint32_t n, uint64_t batch,
                 int dir, int threads);
/* Always-naive O(n^2) variant used to validate the fast fp64 path. */
void fwo_dft_f64_naive(const fwo_c32 *in, fwo_c64 *out, uint32_t n, int dir);

/* max_k |y_k - r_k| / max_k |r_k| and rel-L2 for one transform (SURVEY 8(c) metric). */
void fwo_compare(const fwo_c32 *y, const fwo_c64 *r, uint32_t n,
                 double *max_rel, double *rel_l2);

/* CPU baseline for bench.py: repeats fwo_forward_ref on `batch` transforms of
 * length n with `threads` OpenMP threads until `min_seconds` elapsed (at least
 * once) and returns samples per second of the best repetition. */
double fwo_bench_forward(uint32_t n, uint64_t batch, int threads, double min_seconds,
                         int *reps_out);

int fwo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
