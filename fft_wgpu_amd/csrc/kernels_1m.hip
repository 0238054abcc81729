// kernels_1m.hip -- (3/3) the n = 2^20 two-pass pipeline (headline config) and its experimental fused variant.
#include "device_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// n = 2^20 = 1024 x 1024, two passes.
//
// Index algebra (n = 1024*n1 + n2, k = K1 + 1024*K2):
//   X[K1 + 1024 K2] = sum_{n2} W_N^{n2 K1} * ( sum_{n1} x[1024 n1 + n2] W_1024^{n1 K1} ) * W_1024^{n2 K2}
// pass 1: tile = 16 adjacent columns n2; 1024-point FFT over n1 per column; multiply by W_N^{n2 K1};
//         store Y[K1][n2] into the scratch ring (same row-major shape).
// pass 2: tile = 16 adjacent rows K1; 1024-point FFT over n2 per row; store X[K1 + 1024 K2]
//         (16 adjacent K1 = one 128-byte segment per K2).
// Each 1024-point FFT = radix-32 (registers) -> twiddle W_1024^{n' k1} -> LDS exchange -> radix-32.
// 512 threads, 32 points per thread, 64 data VGPRs, one 64-KiB exchange buffer used twice
// (real parts, then imaginary parts) so that two workgroups fit in a CU's 160 KiB.
// ---------------------------------------------------------------------------
constexpr int XCH_BYTES = 65536;
constexpr int TWI_BYTES = 8192;
constexpr int TWO_BYTES = 8192;

template <int DIR>
__device__ __forceinline__ void stage1_fft_twiddle(v2f (&x)[32], const v2f *twi, uint32_t q)
{
    fft_reg<32, DIR>(x);
    // x[brev(k1)] = Z[k1]; multiply by W_1024^{q*k1}; table layout [k1][q]
    static_for<1, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        constexpr int r = brev<32>(k1);
        x[r] = cmul_tw<DIR>(x[r], twi[k1 * 32 + q]);
    });
}

// One pass-1 tile: column FFTs.  `in`/`out` are the (wave-uniform) bases of a 1024x1024 row-major
// transform; they may be the same transform (in-place: every load of the workgroup completes before the
// first exchange barrier, every store is issued after the last one).
// Requires: twi loaded; `two` free to overwrite (all threads past their previous use).
// OUT_LIN: the output is a scratch slab in tile-contiguous layout -- tile s owns bytes [s*128 KiB, +128 KiB)
// as [K1 (1024)][column (16)]: every pass-1 store instruction of the workgroup covers 4 KiB contiguous,
// and a pass-2 tile finds its 16 rows of a source tile as ONE 2-KiB chunk (measured +5 % over the strided
// matrix layout, tools/tile_probe.hip).
template <int DIR, int AUX_IN, int AUX_OUT, bool OUT_LIN = false>
__device__ __forceinline__ void p1_tile(const v2f *in, v2f *out, uint32_t tile, const v2f *tw_outer_tile,
                                        float *xch, const v2f *twi, v2f *two, uint32_t tid, bool skeleton = false)
{
    const uint32_t c = tid & 15;  // column inside the tile
    const uint32_t q = tid >> 4;  // n' before the exchange, k1 after it
    const uint32_t voff = (q * 1024 + c) * 8;
    const uint32_t soff = tile * 128;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_IN>(rin, voff, soff + j * 262144);
    });
    reinterpret_cast<v4f *>(two)[tid] = reinterpret_cast<const v4f *>(tw_outer_tile)[tid];
    __syncthreads();
    const uint32_t voff_o = OUT_LIN ? (q * 16 + c) * 8 : voff;
    const uint32_t soff_o = OUT_LIN ? tile * 131072 : soff;
    constexpr uint32_t kstep = OUT_LIN ? 4096 : 262144;  // bytes between K1 = q + 32*k2 and q + 32*(k2+1)
    if (skeleton) {  // measurement only: same loads and stores, no arithmetic, no LDS exchange
        static_for<0, 32>([&](auto k_) {
            constexpr int k2 = decltype(k_)::value;
            buf_store<AUX_OUT>(x[k2], rout, voff_o, soff_o + k2 * kstep);
        });
        return;
    }

    stage1_fft_twiddle<DIR>(x, twi, q);

    // exchange: word address c + 16*(k1*32 + (n' ^ (k1&1))) -- conflict-free on both sides
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + 16 * (k1 * 32 + (q ^ (k1 & 1)))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].x = xch[c + 16 * (q * 32 + (np ^ (q & 1)))];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + 16 * (k1 * 32 + (q ^ (k1 & 1)))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].y = xch[c + 16 * (q * 32 + (np ^ (q & 1)))];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = FFT1024 output K1 = q + 32*k2

    // four-step twiddle W_N^{n2*K1} = A[q][c] * B[k2][c]
    const v2f A = two[q * 16 + c];
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        const v2f w = cmul(A, two[512 + k2 * 16 + c]);
        buf_store<AUX_OUT>(cmul_tw<DIR>(x[brev<32>(k2)], w), rout, voff_o, soff_o + k2 * kstep);
    });
}

// One pass-2 tile: row FFTs + transposed store.  `in`/`out` are wave-uniform transform bases; the tile
// reads rows [16*tile, 16*tile+16) and writes columns [16*tile, 16*tile+16).  after_load() runs once every load of the calling thread
// has been issued and before the first barrier; before_store() runs right before the first store.
// The in-place fused kernel uses them for the "all 64 tiles loaded" hand-shake.
template <int DIR, int AUX_IN, int AUX_OUT, bool IN_LIN = false, class AfterLoad, class BeforeStore>
__device__ __forceinline__ void p2_tile(const v2f *in, v2f *out, uint32_t tile, float scale, float *xch,
                                        const v2f *twi, uint32_t tid, AfterLoad after_load,
                                        BeforeStore before_store, bool skeleton = false)
{
    // before the exchange: lane = n' (32 consecutive samples of one row), r = row in the tile
    const uint32_t np = tid & 31;
    const uint32_t r = tid >> 5;
    // IN_LIN (tile-contiguous slab, see p1_tile): sample n2 = 32*j + n' lives in source tile 2*j + (n'>>4),
    // whose rows [16*tile, 16*tile+16) form one 2-KiB chunk [row][16 columns].
    const uint32_t voff_in = IN_LIN ? (np >> 4) * 131072 + r * 128 + (np & 15) * 8 : (r * 1024 + np) * 8;
    const uint32_t soff_in = IN_LIN ? tile * 2048 : tile * 131072;
    constexpr uint32_t jstep = IN_LIN ? 262144 : 256;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_IN>(rin, voff_in, soff_in + j * jstep);
    });
    after_load();
    if (skeleton) {  // measurement only
        before_store();
        const uint32_t vo = ((tid >> 4) * 1024 + (tid & 15)) * 8;
        static_for<0, 32>([&](auto k_) {
            constexpr int k2 = decltype(k_)::value;
            buf_store<AUX_OUT>(x[k2], rout, vo, tile * 128 + k2 * 262144);
        });
        return;
    }

    stage1_fft_twiddle<DIR>(x, twi, np);

    // after the exchange: lane = r' (16 adjacent K1 = one 128-B output segment), k1' = tid >> 4
    const uint32_t r2 = tid & 15;
    const uint32_t k1p = tid >> 4;
    // word address (r*32 + k1)*32 + (n' ^ ((r + 16*(k1&1)) & 31))
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ ((r + 16 * (k1 & 1)) & 31))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    const uint32_t rd_base = (r2 * 32 + k1p) * 32;
    const uint32_t rd_xor = (r2 + 16 * (k1p & 1)) & 31;
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].x = xch[rd_base + (n ^ rd_xor)];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ ((r + 16 * (k1 & 1)) & 31))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].y = xch[rd_base + (n ^ rd_xor)];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = row FFT output K2 = k1p + 32*k2

    before_store();
    const uint32_t voff_out = (k1p * 1024 + r2) * 8;
    const uint32_t soff_out = tile * 128;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        buf_store<AUX_OUT>(x[brev<32>(k2)] * scale, rout, voff_out, soff_out + k2 * 262144);
    });
}

template <int DIR, int AIN = AUX_DEFAULT, int AOUT = AUX_DEFAULT>
__global__ __launch_bounds__(512, 4) void k_p1_1m(const v2f *__restrict__ src, v2f *__restrict__ ring,
                                                  const v2f *__restrict__ tw_inner,
                                                  const v2f *__restrict__ tw_outer, uint32_t ring_slots,
                                                  uint64_t t_first)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + XCH_BYTES + TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x & 63;
    const uint64_t t = t_first + (blockIdx.x >> 6);
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p1_tile<DIR, AIN, AOUT, true>(src + t * (1ull << 20), ring + (t % ring_slots) * (1ull << 20), tile,
                                           tw_outer + (size_t)tile * 1024, xch, twi, two, tid);
}

template <int DIR, int AIN = AUX_DEFAULT, int AOUT = AUX_DEFAULT>
__global__ __launch_bounds__(512, 4) void k_p2_1m(const v2f *__restrict__ ring, v2f *__restrict__ dst,
                                                  const v2f *__restrict__ tw_inner, uint32_t ring_slots,
                                                  uint64_t t_first, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x & 63;
    const uint64_t t = t_first + (blockIdx.x >> 6);
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p2_tile<DIR, AIN, AOUT, true>(ring + (t % ring_slots) * (1ull << 20), dst + t * (1ull << 20), tile, scale,
                                           xch, twi, tid, [] { __syncthreads(); }, [] {});
}

// Mixed launch: even workgroups run pass-1 tiles of one group of transforms, odd workgroups run pass-2
// tiles of the PREVIOUS group of the same chain (whose pass 1 finished in the previous launch on this
// stream).  No dependency exists inside a launch, so there is nothing to wait for; every CU hosts
// pass-1 (HBM-read heavy) and pass-2 (HBM-write heavy) workgroups side by side and both HBM directions
// stay busy across the whole launch.
template <int DIR, int A1IN, int A1OUT, int A2IN, int A2OUT>
__global__ __launch_bounds__(512, 4) void k_mix_1m(const v2f *__restrict__ p1_src, v2f *__restrict__ p1_ring,
                                                   uint32_t n1, const v2f *__restrict__ p2_ring,
                                                   v2f *__restrict__ p2_dst, uint32_t n2,
                                                   const v2f *__restrict__ tw_inner,
                                                   const v2f *__restrict__ tw_outer, float scale, uint32_t dbg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + XCH_BYTES + TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    // role assignment: blocks are dealt round-robin over the 8 XCDs, so `blockIdx & 1` would put every pass-1
    // tile on four XCDs and every pass-2 tile on the other four; bit 3 instead gives each XCD both roles
    // (dbg & 128 selects the old split, for A/B timing).
    uint32_t role, idx;
    if (dbg & 256) {  // pass-2 tiles first, then pass-1 tiles: one HBM direction at a time inside the launch
        const uint32_t n2w = n2 * 64;
        if (blockIdx.x < n2w) { role = 1; idx = blockIdx.x; } else { role = 0; idx = blockIdx.x - n2w; }
    } else if (dbg & 128) { role = blockIdx.x & 1; idx = blockIdx.x >> 1; }
    else { role = (blockIdx.x >> 3) & 1; idx = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7); }
    const bool skel = (dbg & 32) != 0;  // timing-only: memory skeleton
    const uint32_t tile = idx & 63;
    const uint64_t t = idx >> 6;
    if (role == 0) {
        if (t >= n1) return;
        reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
        p1_tile<DIR, A1IN, A1OUT, true>(p1_src + t * (1ull << 20), p1_ring + t * (1ull << 20), tile,
                                  tw_outer + (size_t)tile * 1024, xch, twi, two, tid, skel);
    } else {
        if (t >= n2) return;
        reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
        p2_tile<DIR, A2IN, A2OUT, true>(p2_ring + t * (1ull << 20), p2_dst + t * (1ull << 20), tile, scale, xch, twi, tid,
                                  [] { __syncthreads(); }, [] {}, skel);
    }
}

// ---------------------------------------------------------------------------
// Fused, in-place, persistent 2^20 pipeline: ONE launch per exec, no scratch.
//
// Workgroups pull tickets from one counter.  Ticket order interleaves pass-1 tiles of transform t
// with pass-2 tiles of transform t-D, so HBM reads (pass 1), cache-resident intermediate traffic and
// HBM writes (pass 2) overlap continuously.  Pass 1 overwrites its column tile in place with Y; pass 2
// reads 16 rows of Y and writes the 16-column tile of X over the same transform.  Because pass 2
// transposes, a pass-2 tile may only store once ALL 64 pass-2 tiles of that transform hold their rows in
// registers: `loaded[t]`.  `done1[t]` counts finished pass-1 tiles.
//
// Progress: tickets are handed out in order; pass-1 tiles wait for nothing; a pass-2 tile waits only for
// tickets of its own transform or lower.  The dequeued tickets always form a prefix, at most one
// transform is partially dequeued, so at most 63 workgroups can be parked in the loaded[] wait: any
// launch with >= 64 resident workgroups makes progress.  Spins are bounded (ctl->error) regardless.
// Visibility between workgroups follows cdna_hip_programming.md Guideline 16 (agent-scope release by the
// producer after every wave drained its stores; relaxed poll + one agent-scope acquire by the consumer).
// ---------------------------------------------------------------------------
struct FusedCtl {
    uint32_t ticket;
    uint32_t error;
    uint32_t pad[30];
    // followed by done1[batch], loaded[batch]
};

template <bool RMW = false>
__device__ __forceinline__ bool spin_until_64(uint32_t *p, uint32_t *err)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    // RMW: poll with a returning atomic (served where agent-scope atomics execute, never by a cached copy)
    while ((RMW ? __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 64u) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s: never in a healthy run
            __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

template <int DIR, bool FENCES, int P1_IN, int P1_OUT, int P2_IN, int P2_OUT>
__global__ __launch_bounds__(512, 4) void k_fused_1m(v2f *data, const v2f *__restrict__ tw_inner,
                                                     const v2f *__restrict__ tw_outer, uint32_t *ctl_words,
                                                     uint32_t batch, uint32_t depth, float scale, uint32_t dbg)
{
    // dbg: timing-only ablation switches (results are WRONG when any is set; never set by the product path)
    //   1 skip release fence, 2 skip acquire fence, 4 skip loaded[] wait, 8 skip done1[] wait, 16 skip counters
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + XCH_BYTES + TWI_BYTES);
    uint32_t *ticket = ctl_words;
    uint32_t *err = ctl_words + 1;
    uint32_t *done1 = ctl_words + 32;
    uint32_t *loaded = done1 + batch;

    reinterpret_cast<v4f *>(twi)[threadIdx.x] = reinterpret_cast<const v4f *>(tw_inner)[threadIdx.x];

    const uint32_t total = 128u * batch;
    const uint32_t prologue = 64u * depth;             // pass-1 tiles of transforms 0..depth-1
    const uint32_t steady = 128u * (batch - depth);    // interleaved region

    for (;;) {
        // Opaque per-iteration copy of the thread id: without it LICM hoists ~100 lane-constant LDS/global
        // offsets out of the persistent loop and spills them (cdna_hip_programming.md, attention pitfalls).
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid == 0)
            reinterpret_cast<uint32_t *>(xch)[0] =
                __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // readfirstlane: the ticket is wave-uniform, keep everything derived from it in SGPRs
        const uint32_t k = __builtin_amdgcn_readfirstlane(reinterpret_cast<uint32_t *>(xch)[0]);
        __syncthreads();  // xch is reused by the exchange below
        if (k >= total) break;

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
        v2f *base = data + (uint64_t)t * (1ull << 20);

        if (pass == 1) {
            p1_tile<DIR, P1_IN, P1_OUT>(base, base, tile, tw_outer + (size_t)tile * 1024, xch, twi, two, tid,
                                        (dbg & 32) != 0);
            // publish: every wave drains its stores, then one lane releases at agent scope and counts
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (FENCES && !(dbg & 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (!(dbg & 16)) __hip_atomic_fetch_add(&done1[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (tid == 0) {
                if (!(dbg & 8)) { if (dbg & 64) spin_until_64<true>(&done1[t], err); else spin_until_64<false>(&done1[t], err); }
                if (FENCES && !(dbg & 2)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            p2_tile<DIR, P2_IN, P2_OUT, false>(
                base, base, tile, scale, xch, twi, tid,
                [&] {
                    // this tile's rows are in registers: tell the other 63 tiles of the transform
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0 && !(dbg & 16))
                        __hip_atomic_fetch_add(&loaded[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                },
                [&] {
                    if (tid == 0 && !(dbg & 4)) { if (dbg & 64) spin_until_64<true>(&loaded[t], err); else spin_until_64<false>(&loaded[t], err); }
                    __syncthreads();
                },
                (dbg & 32) != 0);
            // the next iteration's first barrier orders these stores' issue after everything above;
            // nothing in this launch reads X, the kernel boundary publishes it.
        }
    }
}

// Cache-policy variants (measured with tools/fabric_probe2: write-through `sc1` stores for the
// cache-resident intermediate and `nt` on the HBM-facing side lift the mixed-traffic ceiling).
//   policy 0: default loads/stores; fused kernel publishes with agent-scope release/acquire fences
//   policy 1: HBM side nt, intermediate stored sc1 (write-through) and loaded sc1: no fences needed
//             (Guideline 16 form "every store of the handed-off bytes sc1, drained, then counter;
//              every load of them an sc1 load after the poll + workgroup barrier")
//   policy 2: as 1 with sc0|sc1 stores      policy 3: as 1 without nt      policy 4: nt only, fences kept
constexpr int N_POLICIES = 8;  // 5..7: more ring-side variants (two-launch kernels only)

template <int DIR>
static const void *fused_kernel(int policy)
{
    switch (policy) {
        case 1: case 5: case 6: case 7:
            return reinterpret_cast<const void *>(&k_fused_1m<DIR, false, AUX_NT, AUX_SC1, AUX_SC1, AUX_NT>);
        case 2: return reinterpret_cast<const void *>(&k_fused_1m<DIR, false, AUX_NT, AUX_SC1 | 1, AUX_SC1, AUX_NT>);
        case 3: return reinterpret_cast<const void *>(&k_fused_1m<DIR, false, AUX_DEFAULT, AUX_SC1, AUX_SC1, AUX_DEFAULT>);
        case 4: return reinterpret_cast<const void *>(&k_fused_1m<DIR, true, AUX_NT, AUX_DEFAULT, AUX_DEFAULT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_fused_1m<DIR, true, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT>);
    }
}
template <int DIR>
static const void *p1_kernel(int policy)
{
    switch (policy) {
        case 1: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_SC1>);
        case 2: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_SC1 | 1>);
        case 3: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_DEFAULT, AUX_SC1>);
        case 4: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_DEFAULT>);
        case 5: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_SC1>);
        case 6: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_SC1 | AUX_NT>);
        case 7: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_NT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_p1_1m<DIR, AUX_DEFAULT, AUX_DEFAULT>);
    }
}
template <int DIR>
static const void *p2_kernel(int policy)
{
    switch (policy) {
        case 1: case 2: case 4: return reinterpret_cast<const void *>(&k_p2_1m<DIR, AUX_DEFAULT, AUX_NT>);
        case 5: case 6: case 7: return reinterpret_cast<const void *>(&k_p2_1m<DIR, AUX_NT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_p2_1m<DIR, AUX_DEFAULT, AUX_DEFAULT>);
    }
}

template <int DIR>
static const void *mix_kernel(int policy)
{
    switch (policy) {
        case 1: case 5: case 6: case 7:
            return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_NT, AUX_SC1, AUX_DEFAULT, AUX_NT>);
        case 2: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_NT, AUX_SC1 | 1, AUX_DEFAULT, AUX_NT>);
        case 3: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_DEFAULT, AUX_SC1, AUX_DEFAULT, AUX_DEFAULT>);
        case 4: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_NT, AUX_DEFAULT, AUX_DEFAULT, AUX_NT>);
        default: return reinterpret_cast<const void *>(&k_mix_1m<DIR, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT, AUX_DEFAULT>);
    }
}

hipError_t launch_mix_1m(int dir, int policy, const v2f *p1_src, v2f *p1_ring, uint32_t n1, const v2f *p2_ring,
                         v2f *p2_dst, uint32_t n2, const v2f *tw_inner, const v2f *tw_outer, float scale,
                         uint32_t dbg, hipStream_t st)
{
    const uint32_t nmax = n1 > n2 ? n1 : n2;
    if (nmax == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    void *args[] = {&p1_src, &p1_ring, &n1, &p2_ring, &p2_dst, &n2, &tw_inner, &tw_outer, &scale, &dbg};
    const void *k = (dir == FWD) ? mix_kernel<FWD>(policy) : mix_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(nmax * 128), dim3(512), args, XCH_BYTES + TWI_BYTES + TWO_BYTES, st);
}

hipError_t setup_1m_kernels()
{
    const int big = XCH_BYTES + TWI_BYTES + TWO_BYTES, small = XCH_BYTES + TWI_BYTES;
    for (int pol = 0; pol < N_POLICIES; ++pol) {
        const void *ks[8] = {fused_kernel<FWD>(pol), fused_kernel<INV>(pol), p1_kernel<FWD>(pol),
                             p1_kernel<INV>(pol),    mix_kernel<FWD>(pol),   mix_kernel<INV>(pol),
                             p2_kernel<FWD>(pol),    p2_kernel<INV>(pol)};
        for (int i = 0; i < 8; ++i) {
            hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 6 ? big : small);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

size_t fused_ctl_bytes(uint64_t batch) { return sizeof(uint32_t) * (32 + 2 * batch); }

hipError_t launch_fused_1m(int dir, int policy, v2f *data, const v2f *tw_inner, const v2f *tw_outer, uint32_t *ctl,
                           uint32_t batch, uint32_t depth, uint32_t n_workgroups, float scale, uint32_t dbg,
                           hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    if (depth < 1) depth = 1;
    if (depth > batch) depth = batch;
    hipError_t e = hipMemsetAsync(ctl, 0, fused_ctl_bytes(batch), st);
    if (e != hipSuccess) return e;
    const uint32_t max_useful = 128u * batch;
    if (n_workgroups > max_useful) n_workgroups = max_useful;
    void *args[] = {&data, &tw_inner, &tw_outer, &ctl, &batch, &depth, &scale, &dbg};
    const void *k = (dir == FWD) ? fused_kernel<FWD>(policy) : fused_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(n_workgroups), dim3(512), args, XCH_BYTES + TWI_BYTES + TWO_BYTES, st);
}

hipError_t launch_p1_1m(int dir, int policy, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t ring_slots, uint64_t t_first, uint32_t n_transforms, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    void *args[] = {&src, &ring, &tw_inner, &tw_outer, &ring_slots, &t_first};
    const void *k = (dir == FWD) ? p1_kernel<FWD>(policy) : p1_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(n_transforms * 64), dim3(512), args, XCH_BYTES + TWI_BYTES + TWO_BYTES, st);
}

hipError_t launch_p2_1m(int dir, int policy, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t ring_slots,
                        uint64_t t_first, uint32_t n_transforms, float scale, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (policy < 0 || policy >= N_POLICIES) return hipErrorInvalidValue;
    void *args[] = {&ring, &dst, &tw_inner, &ring_slots, &t_first, &scale};
    const void *k = (dir == FWD) ? p2_kernel<FWD>(policy) : p2_kernel<INV>(policy);
    return hipLaunchKernel(k, dim3(n_transforms * 64), dim3(512), args, XCH_BYTES + TWI_BYTES, st);
}

}  // namespace fwa
