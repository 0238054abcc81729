// kernels_lab_1m.hip -- LABORATORY build only (libfft_wgpu_amd_lab.so): the persistent single-launch ring form of the 2^20
// pipeline (path 5), which measured slower than the shipped per-group launches of 1024 x 16 tiles (23.2-24.4 ms against
// 21.1: DESIGN.md 4, item 2 cites it) and is kept for A/B timing and bit-identity tests.  (The 32-column tile, tile_w = 32,
// left in round 6: profiles/round6/lab_pruned_families.patch.)
#include "tile_1m.h"

namespace fwa {

// ---------------------------------------------------------------------------
// Persistent form of the same pipeline: ONE launch per exec, a small ring.
//
// Why: the ring of the two-launch form holds group x chains = 32 transforms (256 MiB), and a ring that large gets
// almost nothing from the 256-MiB Infinity Cache while 32 GiB of HBM traffic stream through it (measured with linear
// streams, profiles/round2/probe_fabric_ring_size.txt: the same traffic mix sustains 8.7 TB/s with a 64-MiB ring,
// 7.3 TB/s with 128-512 MiB).  Smaller launches cannot shrink it (launch gaps and tails dominate below ~16
// transforms per launch), a persistent kernel can: workgroups pull tickets from one counter; ticket order interleaves
// pass-1 tiles of transform t with pass-2 tiles of transform t - depth, so a ring of depth + a few slots suffices.
//
// Hand-offs (cdna_hip_programming.md Guideline 16, R1 counter form): pass 1 stores the ring write-through (sc1),
// every wave drains its stores, workgroup barrier, one lane adds to done1[t]; a pass-2 tile polls done1[t] == 64 with
// one lane (relaxed agent-scope load), workgroup barrier, then EVERY ring load is an sc1 buffer load.  Slot reuse: a
// pass-2 tile adds to rdone[t] once all its loads have landed; pass-1 tiles of transform t + ring_slots poll it before
// their first store.
// Progress: tickets are handed out in order and a ticket only ever waits for lower tickets, each of which is held by
// a workgroup that is running (no co-residency assumption, any grid size).  Spins are bounded (2 s) and set ctl[1].
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool spin_until_64(uint32_t *p, uint32_t *err)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 64u) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s: never in a healthy run
            __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

constexpr int RING_CTL_WORDS = 32;  // ctl[0] ticket, ctl[1] error; then done1[batch], rdone[batch]

template <int DIR>
__global__ __launch_bounds__(512, 4) void k_ring_1m(const v2f *src, v2f *dst, v2f *ring, const v2f *__restrict__ tw_inner,
                                                    const v2f *__restrict__ tw_outer, uint32_t *ctl, uint32_t batch,
                                                    uint32_t depth, uint32_t ring_slots, float scale)
{
    using G = Geom<16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + G::XCH_BYTES + G::TWI_BYTES);
    uint32_t *s_next = reinterpret_cast<uint32_t *>(smem + G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES);
    uint32_t *ticket = ctl, *err = ctl + 1, *done1 = ctl + RING_CTL_WORDS, *rdone = done1 + batch;

    reinterpret_cast<v4f *>(twi)[threadIdx.x] = reinterpret_cast<const v4f *>(tw_inner)[threadIdx.x];
    if (threadIdx.x == 0) *s_next = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    uint32_t k = __builtin_amdgcn_readfirstlane(*s_next);

    const uint32_t total = 128u * batch;
    const uint32_t prologue = 64u * depth;           // pass-1 tiles of transforms 0 .. depth-1
    const uint32_t steady = 128u * (batch - depth);  // pass-1 tiles of t + depth interleaved with pass-2 tiles of t

    while (k < total) {
        // Opaque per-iteration copy of the thread id: without it LICM hoists ~100 lane-constant LDS/global offsets out
        // of the persistent loop and spills them.
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        uint32_t nxt = 0;  // the next ticket is requested now; its latency hides behind this tile
        if (tid == 0) nxt = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        uint32_t pass, t, tile;
        if (k < prologue) {
            pass = 1; t = k >> 6; tile = k & 63;
        } else if (k - prologue < steady) {
            const uint32_t kk = k - prologue;
            const uint32_t s = kk >> 7, r = kk & 127;
            tile = r >> 1;
            if ((r & 1) == 0) { pass = 1; t = s + depth; } else { pass = 2; t = s; }
        } else {
            const uint32_t kk = k - prologue - steady;
            pass = 2; t = (batch - depth) + (kk >> 6); tile = kk & 63;
        }
        v2f *slab = ring + (uint64_t)(t % ring_slots) * (1ull << 20);

        if (pass == 1) {
            if (t >= ring_slots) {  // slot still being read by transform t - ring_slots?
                if (tid == 0) spin_until_64(&rdone[t - ring_slots], err);
                __syncthreads();
            }
            p1_tile<DIR, 16>(src + (uint64_t)t * (1ull << 20), slab, tile, tw_outer + (size_t)tile * 1024, xch, twi, two, tid);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&done1[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (tid == 0) spin_until_64(&done1[t], err);
            __syncthreads();
            p2_tile<DIR, 16, AUX_SC1>(slab, dst + (uint64_t)t * (1ull << 20), tile, scale, xch, twi, tid, [&] {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's rows are in registers
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(&rdone[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
        }
        if (tid == 0) *s_next = nxt;
        __syncthreads();
        k = __builtin_amdgcn_readfirstlane(*s_next);
    }
}

size_t ring_ctl_bytes(uint64_t batch) { return sizeof(uint32_t) * (RING_CTL_WORDS + 2 * batch); }

hipError_t launch_ring_1m(int dir, const v2f *src, v2f *dst, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                          uint32_t *ctl, uint32_t batch, uint32_t depth, uint32_t ring_slots, uint32_t n_workgroups,
                          float scale, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    if (depth < 1) depth = 1;
    if (depth > batch) depth = batch;
    if (ring_slots < depth + 1) return hipErrorInvalidValue;  // pass-1 of t + depth runs beside pass-2 of t
    hipError_t e = hipMemsetAsync(ctl, 0, ring_ctl_bytes(batch), st);
    if (e != hipSuccess) return e;
    if (n_workgroups > 128u * batch) n_workgroups = 128u * batch;
    using G = Geom<16>;
    void *args[] = {&src, &dst, &ring, &tw_inner, &tw_outer, &ctl, &batch, &depth, &ring_slots, &scale};
    const void *k = dir == FWD ? reinterpret_cast<const void *>(&k_ring_1m<FWD>) : reinterpret_cast<const void *>(&k_ring_1m<INV>);
    return hipLaunchKernel(k, dim3(n_workgroups), dim3(512), args, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES + 16, st);
}

hipError_t setup_lab_1m_kernels()
{
    hipError_t e = hipSuccess;
    using G = Geom<16>;
    const int lds = G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES + 16;
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_1m<FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_1m<INV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}

}  // namespace fwa
