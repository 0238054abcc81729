// kernels_lab_fused.hip -- LABORATORY build only: config C2 (one 2^20 transform) as ONE launch (the idea
// and its prediction: HISTORY.md, "C2 in one launch").  The three passes of the latency plan 64 x 64 x 256 -- the same tile_body
// arithmetic as the three k_tile launches, so results are bit-identical -- separated by two grid barriers.
//
// Hand-offs (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md, visibility, first row of the measured-forms table): every
// slab store is sc1 (write-through); every storing wave drains (`s_waitcnt vmcnt(0)`); workgroup barrier; ONE lane stores the
// workgroup's flag with a relaxed agent-scope atomic (= sc1 store); wave 0 polls all 256 flags (one 16-byte sc1 load per lane);
// workgroup barrier; EVERY slab load of the next pass is an sc1 buffer load -- no acquire fence.
// Epochs: a workgroup reads its own flag at entry (v: what the previous exec left, equal in every flag once an exec has completed)
// and the two barriers wait for v + 1 and v + 2 -- no memset node in front of the launch, and the same under graph replay.
// Progress: 256 workgroups of 256 threads and 42 KiB of LDS are all resident on 256 CUs; spins are bounded (0.2 s) and set ctl[1].
#include "tile_body.h"

namespace fwa {

typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
constexpr uint32_t FUSED_WGS = 256, FUSED_FLAG_WORD = 64;   // ctl[1] = error word, flags at ctl + 64 (256 B in), one per workgroup

__device__ __forceinline__ void grid_barrier_256(uint32_t *ctl, uint32_t wg, uint32_t epoch, uint32_t tid)
{
    uint32_t *flags = ctl + FUSED_FLAG_WORD;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave: its sc1 stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(flags, 0, FUSED_WGS * 4, 0x00020000);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        for (;;) {
            const v4u32 f = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rf, tid * 16, 0, AUX_SC1));
            const bool ok = (int32_t)(f.x - epoch) >= 0 && (int32_t)(f.y - epoch) >= 0 && (int32_t)(f.z - epoch) >= 0
                && (int32_t)(f.w - epoch) >= 0;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {   // 0.2 s: never in a healthy run
                if (tid == 0) __hip_atomic_fetch_or(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the next pass's loads below the poll
    __syncthreads();
}

struct FusedArgs {
    const v2f *in;
    v2f *slab, *out;
    const v2f *tw_a, *tw_b, *tw_c, *lo1, *hi1, *lo_b, *hi_b;
    uint32_t *ctl;
    float scale;
    uint32_t stamps;   // diagnostic: 1 = every workgroup records six s_memrealtime stamps (100 MHz) behind the flags
};

template <int DIR>
__global__ __launch_bounds__(256) void k_fused_c2(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t N1 = 64, N2 = 64, N3 = 256, N = N1 * N2 * N3;
    const uint32_t wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t v = __builtin_amdgcn_readfirstlane(
        __hip_atomic_load(a.ctl + FUSED_FLAG_WORD + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    v2f *lds_w = reinterpret_cast<v2f *>(smem) + wave * (16 * tile_pstr(64));
    uint64_t *stamp = reinterpret_cast<uint64_t *>(a.ctl + FUSED_FLAG_WORD + FUSED_WGS) + wg * 8;
    auto mark = [&](int slot) {
        if (a.stamps && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[slot] = __builtin_amdgcn_s_memrealtime(); }
    };
    mark(0);
    const uint32_t bid = wg * 4 + wave;   // passes A and B: one 64-point x 16-column tile per wave
    {   // pass A: FFT over n1 (pitch N / N1), four-step twiddle of domain n, user buffer -> slab
        tile_body<6, 16, DIR, TILE_COLS, true, AUX_NT, AUX_SC1>(a.in + bid * 16, a.slab + bid * 16, bid * 16, a.tw_a, a.lo1, a.hi1,
                                                                 N / N1, 0, 1.0f, lds_w, lane);
    }
    mark(1);
    grid_barrier_256(a.ctl, wg, v + 1, tid);
    mark(2);
    {   // pass B: FFT over n2 inside every k1-plane (pitch N3), twiddle of domain N2 * N3, in place in the slab
        const uint32_t tile = bid % (N3 / 16), d1 = bid / (N3 / 16);
        v2f *p = a.slab + d1 * (N2 * N3) + tile * 16;
        tile_body<6, 16, DIR, TILE_COLS, true, AUX_SC1, AUX_SC1>(p, p, tile * 16, a.tw_b, a.lo_b, a.hi_b, N3, 0, 1.0f, lds_w, lane);
    }
    mark(3);
    grid_barrier_256(a.ctl, wg, v + 2, tid);
    mark(4);
    {   // pass C: 256-point rows of the last axis, 16 adjacent k1 per workgroup, transposed store into the result buffer
        const uint32_t tile = wg % (N1 / 16), d1 = wg / (N1 / 16);
        tile_body<8, 16, DIR, TILE_ROWS_T, true, AUX_SC1, AUX_NT>(a.slab + d1 * N3 + tile * (16 * (N / N1)), a.out + d1 * N1 + tile * 16,
                                                                   tile * 16, a.tw_c, nullptr, nullptr, N / N1, N1 * N2, a.scale,
                                                                   reinterpret_cast<v2f *>(smem), tid);
    }
    mark(5);
}

size_t fused_c2_ctl_bytes() { return (FUSED_FLAG_WORD + FUSED_WGS) * sizeof(uint32_t) + FUSED_WGS * 8 * sizeof(uint64_t); }

hipError_t launch_fused_c2(int dir, const v2f *in, v2f *slab, v2f *out, const v2f *tw_a, const v2f *tw_b, const v2f *tw_c,
                           const v2f *lo1, const v2f *hi1, const v2f *lo_b, const v2f *hi_b, uint32_t *ctl, float scale,
                           bool stamps, hipStream_t st)
{
    FusedArgs a{in, slab, out, tw_a, tw_b, tw_c, lo1, hi1, lo_b, hi_b, ctl, scale, stamps ? 1u : 0u};
    const size_t lds_ab = 4 * tile_lds(6, 16), lds_c = tile_lds(8, 16);
    const size_t lds = lds_ab > lds_c ? lds_ab : lds_c;
    void *args[] = {&a};
    const void *k = dir == FWD ? reinterpret_cast<const void *>(&k_fused_c2<FWD>) : reinterpret_cast<const void *>(&k_fused_c2<INV>);
    return hipLaunchKernel(k, dim3(FUSED_WGS), dim3(256), args, lds, st);
}

}  // namespace fwa
