// kernels_chunk.hip -- n = 2 .. 256: whole transforms inside a contiguous 64-KiB chunk per workgroup (gfx950 only).
//
// Below n = 512 a transform is shorter than what a wavefront moves with one instruction, so a kernel whose lanes
// address "their" points directly (k_tiny16, k_small16: laboratory build) issues loads whose lanes are 16 .. 128 bytes apart:
// every instruction touches up to 64 cache lines and the same line is touched by up to 16 instructions (0.52-0.67 of the
// roofline).  Here global memory is only ever addressed linearly, in the launch shape that streams fastest on this part
// (tools/stream_probe.hip, profiles/round4/probe_stream_shapes.txt: 6.3 TB/s against 5.5-5.9 for every shape whose instructions
// hop through the chunk -- round 2's 32-KiB chunks included): one 256-thread workgroup per 64-KiB-aligned chunk, every WAVE
// walks its own 16 KiB front to back with 32 loads of 512 contiguous bytes, all in flight before the first use.  The
// arithmetic runs on one HALF of the chunk at a time (the first 8 KiB of every wave's range, then the second: 4096 samples
// = whole transforms, n divides 1024): the half is parked in LDS (one pad element per 16: conflict-free on both sides) and
// transformed there -- 16 points per thread, radix 16 [x 2, 4, 8, 16] with one exchange, the recurrence of fft.wgsl:27-62
// generalised to radix R, twiddle table of processor.rs:43-49 -- and read back into the registers it came from; 34 KiB of LDS,
// four workgroups per CU.  32 linear stores follow.  The buffer descriptor ends with the data, so a ragged last chunk needs
// no bounds code (loads return 0, stores are dropped).  In place allowed: a workgroup reads its chunk completely before it
// writes any of it.
#include "device_common.h"

// Where the 15 stage-0 twiddle look-ups of n >= 32 (they depend on the thread index only and are the same for both halves)
// are issued: 0 = at the point of use, once per half -- each an exposed cache latency inside the barrier-to-barrier phase;
// 1 = once, before the data loads; 4 = once, right behind them.  Per size at the 32-GiB footprint, one library per value,
// interleaved (tools/ab_libs.py, profiles/round5/ab_twiddle_prefetch_chunk.jsonl; identical bits): 32: 0.758 -> 0.800 (behind),
// 64: 0.740 -> 0.785 (before), 128: 0.764 -> 0.780 (before; behind: 0.729), 256: 0.751 -> 0.777 (behind).
#ifndef FWA_PF_CHUNK
#define FWA_PF_CHUNK -1
#endif
constexpr int chunk_prefetch_default(int lgn) { return (lgn == 6 || lgn == 7) ? 1 : 4; }

namespace fwa {

template <int LGN, int DIR, int AIN = AUX_NT, int AOUT = AUX_NT>
__global__ __launch_bounds__(256, (LGN >= 6 ? 3 : 4)) void k_chunk(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                  const v2f *__restrict__ tw, uint64_t n_samples, float scale)
{
    constexpr int N = 1 << LGN;
    constexpr int T = 256;
    constexpr uint32_t CH = 32 * T;   // samples per workgroup (64 KiB)
    constexpr uint32_t HALF = 16 * T;  // samples transformed at a time
    constexpr bool EARLY = LGN >= 6;   // store a half as soon as it is done (register pressure: see below)
    __shared__ v2f lds_all[HALF + HALF / 16];
    const uint32_t tid = threadIdx.x;
    const uint64_t e0 = (uint64_t)one_launch_block() * CH;
    const uint64_t left = n_samples - e0;
    const uint32_t valid = left < CH ? (uint32_t)left * 8u : CH * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(src + e0), 0, valid, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dst + e0, 0, valid, 0x00020000);
    auto pad = [](uint32_t p) { return p + (p >> 4); };
    // wave w walks samples [2048 w, 2048 w + 2048) of the chunk: access i covers samples 2048 w + 64 i + lane
    const uint32_t wv = tid >> 6, lane = tid & 63;
    const uint32_t voff = (wv * 2048 + lane) * 8;
    // ... and access i of half h = i / 16 sits at position 1024 w + 64 (i % 16) + lane of the half's LDS space: the four
    // 1024-sample blocks of a half are whole transforms, back to back
    const uint32_t park = wv * 1024 + lane;

    constexpr int PF = LGN > 4 ? (FWA_PF_CHUNK < 0 ? chunk_prefetch_default(LGN) : FWA_PF_CHUNK) : 0;
    v2f wpre[16];
    auto prefetch = [&] {
        if constexpr (LGN > 4)
            static_for<1, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; wpre[q] = tw_lookup<N>(tw, (tid % (N / 16)) * q); });
    };
    if constexpr (PF & 1) prefetch();
    v2f y[32];
    static_for<0, 32>([&](auto i_) { constexpr int i = decltype(i_)::value; y[i] = buf_load<AIN>(rin, voff, i * 512); });
    if constexpr (PF & 4) prefetch();
  static_for<0, 2>([&](auto h_) {
    constexpr int h = decltype(h_)::value;
    v2f x[16], v[16];
    if constexpr (h == 1) __syncthreads();  // the first half has been read back
    static_for<0, 16>([&](auto u_) { constexpr int u = decltype(u_)::value; lds_all[pad(park + 64 * u)] = y[16 * h + u]; });
    __syncthreads();

    if constexpr (LGN <= 4) {
        // samples 16*tid .. 16*tid + 15 = 16/N whole transforms; the thread reads and rewrites only its own 17 slots
        v2f *mine = lds_all + 17 * tid;
        static_for<0, 16>([&](auto i_) { constexpr int i = decltype(i_)::value; x[i] = mine[i]; });
        static_for<0, 16 / N>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            v2f z[N];
            static_for<0, N>([&](auto i_) { constexpr int i = decltype(i_)::value; z[i] = x[g * N + i]; });
            fft_reg<N, DIR>(z);
            static_for<0, N>([&](auto k_) { constexpr int k = decltype(k_)::value; mine[g * N + k] = z[brev<N>(k)] * scale; });
        });
    } else {
        constexpr int TPX = N / 16;        // threads per transform
        constexpr int RL = 1 << (LGN % 4);  // radix of the second stage when it is not 16
        v2f *lds = lds_all + (tid / TPX) * (N + N / 16);
        const uint32_t t = tid % TPX;
        // stage 0 (J = 1, s = t): inputs t + m*N/16, output q at t*16 + q, twiddle W_n^{t*q}
        static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(t + m * TPX)]; });
        fft_reg<16, DIR>(x);
        static_for<0, 16>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            v[q] = x[brev<16>(q)];
            if constexpr (q != 0) v[q] = cmul_tw<DIR>(v[q], (PF & 5) ? wpre[q] : tw_lookup<N>(tw, t * q));
        });
        __syncthreads();  // every stage-0 operand has been read
        static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; lds[pad(t * 16 + q)] = v[q]; });
        __syncthreads();
        if constexpr (RL == 1) {  // n = 256: second radix-16 stage (s = 0, J = 16): inputs t + 16*m, output q at t + 16*q
            static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(t + m * 16)]; });
            __syncthreads();
            fft_reg<16, DIR>(x);
            static_for<0, 16>([&](auto q_) { constexpr int q = decltype(q_)::value; lds[pad(t + 16 * q)] = x[brev<16>(q)] * scale; });
        } else {  // n = 32, 64, 128: 16/RL butterflies of radix RL (idx = t + b*TPX, s = 0), output q at idx + 16*q
            static_for<0, 16>([&](auto i_) {
                constexpr uint32_t i = decltype(i_)::value;
                x[i] = lds[pad(t + (i / RL) * TPX + (i % RL) * (N / RL))];
            });
            __syncthreads();
            static_for<0, 16 / RL>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                v2f z[RL];
                static_for<0, RL>([&](auto m_) { constexpr int m = decltype(m_)::value; z[m] = x[b * RL + m]; });
                fft_reg<RL, DIR>(z);
                static_for<0, RL>([&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    lds[pad(t + b * TPX + q * 16)] = z[brev<RL>(q)] * scale;
                });
            });
        }
    }
    __syncthreads();
    static_for<0, 16>([&](auto u_) { constexpr int u = decltype(u_)::value; y[16 * h + u] = lds_all[pad(park + 64 * u)]; });
    // from n = 64 on the two-stage arithmetic needs the registers of the finished half: its 16 stores go out at once
    if constexpr (EARLY)
        static_for<0, 16>([&](auto u_) { constexpr int i = 16 * h + decltype(u_)::value; buf_store<AOUT>(y[i], rout, voff, i * 512); });
  });
    if constexpr (!EARLY)
        static_for<0, 32>([&](auto i_) { constexpr int i = decltype(i_)::value; buf_store<AOUT>(y[i], rout, voff, i * 512); });
}

template <int DIR>
static hipError_t launch_chunk_dir(const v2f *src, v2f *dst, const v2f *tw, uint32_t lg_n, uint64_t n_samples, float scale,
                                   hipStream_t st)
{
    const uint64_t blocks = (n_samples + 8191) / 8192;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 g((uint32_t)blocks), b(256);
    switch (lg_n) {
        case 1: hipLaunchKernelGGL((k_chunk<1, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 2: hipLaunchKernelGGL((k_chunk<2, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 3: hipLaunchKernelGGL((k_chunk<3, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 4: hipLaunchKernelGGL((k_chunk<4, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 5: hipLaunchKernelGGL((k_chunk<5, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 6: hipLaunchKernelGGL((k_chunk<6, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 7: hipLaunchKernelGGL((k_chunk<7, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        case 8: hipLaunchKernelGGL((k_chunk<8, DIR>), g, b, 0, st, src, dst, tw, n_samples, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_chunk(int dir, const v2f *src, v2f *dst, const v2f *tw, uint32_t n, uint64_t batch, float scale,
                        hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    uint32_t lg_n = 0;
    while ((1u << lg_n) < n) ++lg_n;
    return dir == FWD ? launch_chunk_dir<FWD>(src, dst, tw, lg_n, batch * n, scale, st)
                      : launch_chunk_dir<INV>(src, dst, tw, lg_n, batch * n, scale, st);
}

}  // namespace fwa
