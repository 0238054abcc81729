// kernels_1m.hip -- (3/3) the n = 2^20 two-pass pipeline (headline config C3).
//
// Replaces the 20 global-memory radix-2 passes of reference src/kernel/fft4.wgsl:36-101 by two passes of
// register-resident 32 x 32 FFTs: one HBM round trip plus one round trip through a cache-sized ring.
#include "device_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// n = 2^20 = 1024 x 1024, two passes.
//
// Index algebra (n = 1024*n1 + n2, k = K1 + 1024*K2):
//   X[K1 + 1024 K2] = sum_{n2} W_N^{n2 K1} * ( sum_{n1} x[1024 n1 + n2] W_1024^{n1 K1} ) * W_1024^{n2 K2}
// pass 1: tile = W adjacent columns n2 (one W*8-byte segment per matrix row); 1024-point FFT over n1 per
//         column; multiply by W_N^{n2 K1}; store Y[K1][n2] into the scratch ring, tile-contiguous.
// pass 2: tile = W adjacent rows K1; 1024-point FFT over n2 per row; store X[K1 + 1024 K2]
//         (W adjacent K1 = one W*8-byte segment per K2).
// Each 1024-point FFT = radix-32 (registers) -> twiddle W_1024^{n' k1} -> LDS exchange -> radix-32.
// A workgroup has 32*W threads with 32 points each (64 data VGPRs); the exchange buffer holds the real
// parts, then the imaginary parts (W*4 KiB).
//   W = 16: 512 threads, 80 KiB LDS, two workgroups per CU, 128-B HBM segments.
//   W = 32: 1024 threads, 152 KiB LDS, one workgroup per CU, 256-B HBM segments (the column-tile stream
//           sustains more with 256-B segments: profiles/round1/probe_tile_pitch_width.txt).
// Cache policy (measured, profiles/round1/probe_fabric_cache_policies.txt): user-buffer accesses `nt`,
// ring stores `sc1` (write-through), ring loads default.
// ---------------------------------------------------------------------------
template <int W>
struct Geom {
    static_assert(W == 16 || W == 32, "tile width");
    static constexpr int LGW = (W == 16) ? 4 : 5;
    static constexpr int THREADS = 32 * W;
    static constexpr int TILES = 1024 / W;
    static constexpr int XCH_BYTES = W * 4096;        // one float per point of the tile
    static constexpr int TWI_BYTES = 8192;            // [k1][n'] = W_1024^{n' k1}
    static constexpr int TWO_BYTES = 2 * 32 * W * 8;  // per tile A[32][W], B[32][W]
    static constexpr uint32_t TILE_BYTES = W * 8192;  // one tile of the ring slab
    // XOR swizzles that make both sides of the exchange conflict-free (bank = word address mod 32)
    static __device__ __forceinline__ constexpr uint32_t sw1(uint32_t k1) { return W == 16 ? (k1 & 1) : 0; }
    static __device__ __forceinline__ uint32_t sw2(uint32_t r, uint32_t k1)
    {
        return W == 16 ? ((r + 16 * (k1 & 1)) & 31) : r;
    }
};

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

// One pass-1 tile: column FFTs.  `in` is the (wave-uniform) base of a 1024x1024 row-major transform, `out`
// the base of its ring slab: tile s owns bytes [s*W*8 KiB, +W*8 KiB) as [K1 (1024)][column (W)], so every
// store instruction of a wave covers 512 contiguous bytes and a pass-2 tile finds its W rows of a source
// tile as ONE contiguous W*W*8-byte chunk.
template <int DIR, int W>
__device__ __forceinline__ void p1_tile(const v2f *in, v2f *out, uint32_t tile, const v2f *tw_outer_tile,
                                        float *xch, const v2f *twi, v2f *two, uint32_t tid)
{
    using G = Geom<W>;
    const uint32_t c = tid & (W - 1);  // column inside the tile
    const uint32_t q = tid >> G::LGW;  // n' before the exchange, k1 after it
    const uint32_t voff = (q * 1024 + c) * 8;
    const uint32_t soff = tile * (W * 8);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP(0);
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_NT>(rin, voff, soff + j * 262144);
    });
    FWA_STAMP(1);
    reinterpret_cast<v4f *>(two)[tid] = reinterpret_cast<const v4f *>(tw_outer_tile)[tid];
    __syncthreads();

    stage1_fft_twiddle<DIR>(x, twi, q);

    // exchange: word address c + W*(k1*32 + (n' ^ sw1(k1)))
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + W * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].x = xch[c + W * (q * 32 + (np ^ G::sw1(q)))];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[c + W * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int np = decltype(n_)::value;
        x[np].y = xch[c + W * (q * 32 + (np ^ G::sw1(q)))];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = FFT1024 output K1 = q + 32*k2

    // four-step twiddle W_N^{n2*K1} = A[q][c] * B[k2][c]
    const v2f A = two[q * W + c];
    const uint32_t voff_o = (q * W + c) * 8;
    const uint32_t soff_o = tile * G::TILE_BYTES;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        const v2f w = cmul(A, two[32 * W + k2 * W + c]);
        buf_store<AUX_SC1>(cmul_tw<DIR>(x[brev<32>(k2)], w), rout, voff_o, soff_o + k2 * (32 * W * 8));
    });
    FWA_STAMP(3);
}

// One pass-2 tile: row FFTs + transposed store.  `in` = ring slab of the transform, `out` = its 1024x1024
// result matrix; the tile reads rows [W*tile, W*tile+W) and writes columns [W*tile, W*tile+W).
// AUX_IN: cache policy of the ring loads; after_load() runs once every load of the calling thread has been issued
// and must contain a workgroup barrier (it also makes the twiddle table visible).
template <int DIR, int W, int AUX_IN, class AfterLoad>
__device__ __forceinline__ void p2_tile(const v2f *in, v2f *out, uint32_t tile, float scale, float *xch,
                                        const v2f *twi, uint32_t tid, AfterLoad after_load)
{
    using G = Geom<W>;
    // before the exchange: lane = n' (32 consecutive samples of one row), r = row in the tile.
    // Sample n2 = 32*j + n' lives in source tile n2 / W, whose rows [W*tile, +W) are one chunk [row][W columns].
    const uint32_t np = tid & 31;
    const uint32_t r = tid >> 5;
    const uint32_t voff_in = (np >> G::LGW) * G::TILE_BYTES + r * (W * 8) + (np & (W - 1)) * 8;
    const uint32_t soff_in = tile * (W * W * 8);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in), rout = make_rsrc(out);
    v2f x[32];
    FWA_ENTRY_HOOK();
    FWA_STAMP(0);
    static_for<0, 32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        x[j] = buf_load<AUX_IN>(rin, voff_in, soff_in + j * 262144);
    });
    FWA_STAMP(1);
    after_load();

    stage1_fft_twiddle<DIR>(x, twi, np);

    // after the exchange: lane = r' (W adjacent K1 = one output segment), k1' = tid / W
    const uint32_t r2 = tid & (W - 1);
    const uint32_t k1p = tid >> G::LGW;
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ G::sw2(r, k1))] = x[brev<32>(k1)].x;
    });
    __syncthreads();
    const uint32_t rd_base = (r2 * 32 + k1p) * 32;
    const uint32_t rd_xor = G::sw2(r2, k1p);
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].x = xch[rd_base + (n ^ rd_xor)];
    });
    __syncthreads();
    static_for<0, 32>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        xch[(r * 32 + k1) * 32 + (np ^ G::sw2(r, k1))] = x[brev<32>(k1)].y;
    });
    __syncthreads();
    static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        x[n].y = xch[rd_base + (n ^ rd_xor)];
    });

    fft_reg<32, DIR>(x);  // x[brev(k2)] = row FFT output K2 = k1p + 32*k2

    const uint32_t voff_out = (k1p * 1024 + r2) * 8;
    const uint32_t soff_out = tile * (W * 8);
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        buf_store<AUX_NT>(x[brev<32>(k2)] * scale, rout, voff_out, soff_out + k2 * 262144);
    });
    FWA_STAMP(3);
}

// XCD-aware block -> tile mapping (cdna_hip_programming.md T1).  Blocks are dealt round-robin over the 8 XCDs, so
// with tile = blockIdx % TILES an XCD only ever holds tiles of the same residue mod 8: every one of its resident
// workgroups then streams column tiles whose addresses agree modulo 1 KiB, i.e. they all fall on the same few L2
// channels.  The swizzle hands each XCD a contiguous run of (transform, tile) indices instead: its 64 resident
// workgroups are 64 consecutive tiles and cover whole 8-KiB rows.  Grid sizes are multiples of 8 (TILES is).
__device__ __forceinline__ uint32_t xcd_block(uint32_t swizzle)
{
    const uint32_t b = blockIdx.x;
    return swizzle ? (b & 7u) * (gridDim.x >> 3) + (b >> 3) : b;
}

// ---------------------------------------------------------------------------
// k_p1_gen: the same 1024-point column pass for any n = 1024 * P (P = 2^4 .. 2^20 columns): pass A of the tiled
// plans whose first factor is 1024 (2^18, 2^19 and re-factorised larger sizes).  Differences to p1_tile: the row
// pitch P is a run-time value (scalar offsets), the output keeps the matrix layout (the next pass is k_tile, which
// reads rows at pitch P), and the per-tile four-step factors A[q][c] = W_n^{col*q}, B[k2][c] = W_n^{32*col*k2} are
// computed by the workgroup itself from the two-level table of domain n (one look-up pair per thread each) instead
// of a precomputed per-tile table.  Same 80 KiB of LDS, two workgroups per CU (k_tile at L = 1024 needs 138 KiB and
// two full-complex exchanges).
// ---------------------------------------------------------------------------
template <int DIR, int AUX_OUT>
__global__ __launch_bounds__(512) void k_p1_gen(const v2f *__restrict__ src, v2f *__restrict__ dst,
                                                const v2f *__restrict__ tw_inner, const v2f *__restrict__ tw_lo,
                                                const v2f *__restrict__ tw_hi, uint32_t pitch, uint64_t in_sb,
                                                uint64_t out_sb, uint32_t xcd_swizzle)
{
    using G = Geom<16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + G::XCH_BYTES + G::TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(xcd_swizzle & 1u);
    const uint32_t tiles = pitch >> 4;
    const uint32_t tile = bid % tiles;
    const uint64_t t = bid / tiles;
    const uint32_t c = tid & 15, q = tid >> 4;
    // rows q + 32*j of the matrix: one buffer descriptor per eight j (a quarter of the transform, <= 2 GiB at
    // n = 2^30), so that every byte offset stays below 2^32
    const uint64_t quarter = (uint64_t)pitch * 256;  // elements
    const uint32_t qbytes = pitch * 2048u;
    const v2f *sbase = src + t * in_sb;
    const uint32_t voff = (q * pitch + c) * 8;
    const uint32_t soff = tile * 128;
    const uint32_t jstep = pitch * 256;  // bytes between rows q + 32*j and q + 32*(j+1)
    v2f x[32];
    static_for<0, 4>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(sbase + g * quarter), 0, qbytes, 0x00020000);
        static_for<0, 8>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            x[8 * g + j] = buf_load<AUX_NT>(rin, voff, soff + j * jstep);
        });
    });
    reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    {   // A[q][c] = W_n^{col*q} (thread (q, c)), B[k2][c] = W_n^{32*col*k2} (thread (k2, c)): exponents < n <= 2^30
        const uint32_t col = tile * 16 + c;
        const uint32_t ea = col * q, eb = ea << 5;
        two[q * 16 + c] = cmul(tw_hi[ea >> 10], tw_lo[ea & 1023]);
        two[512 + q * 16 + c] = cmul(tw_hi[eb >> 10], tw_lo[eb & 1023]);
    }
    __syncthreads();

    stage1_fft_twiddle<DIR>(x, twi, q);
    static_for<0, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; xch[c + 16 * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].x; });
    __syncthreads();
    static_for<0, 32>([&](auto n_) { constexpr int np = decltype(n_)::value; x[np].x = xch[c + 16 * (q * 32 + (np ^ G::sw1(q)))]; });
    __syncthreads();
    static_for<0, 32>([&](auto k_) { constexpr int k1 = decltype(k_)::value; xch[c + 16 * (k1 * 32 + (q ^ G::sw1(k1)))] = x[brev<32>(k1)].y; });
    __syncthreads();
    static_for<0, 32>([&](auto n_) { constexpr int np = decltype(n_)::value; x[np].y = xch[c + 16 * (q * 32 + (np ^ G::sw1(q)))]; });
    fft_reg<32, DIR>(x);  // x[brev(k2)] = output K1 = q + 32*k2

    const v2f A = two[q * 16 + c];
    v2f *dbase = dst + t * out_sb;
    static_for<0, 4>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dbase + g * quarter, 0, qbytes, 0x00020000);
        static_for<0, 8>([&](auto j_) {
            constexpr int k2 = 8 * g + decltype(j_)::value;
            const v2f w = cmul(A, two[512 + k2 * 16 + c]);
            buf_store<AUX_OUT>(cmul_tw<DIR>(x[brev<32>(k2)], w), rout, voff, soff + (k2 & 7) * jstep);
        });
    });
}

template <int DIR, int W>
__global__ __launch_bounds__(32 * W) void k_p1_1m(const v2f *__restrict__ src, v2f *__restrict__ ring,
                                                  const v2f *__restrict__ tw_inner,
                                                  const v2f *__restrict__ tw_outer, uint32_t xcd_swizzle)
{
    using G = Geom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    v2f *two = reinterpret_cast<v2f *>(smem + G::XCH_BYTES + G::TWI_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(xcd_swizzle & 1u);
    const uint32_t tile = bid % G::TILES;
    const uint64_t t = bid / G::TILES;  // transform inside the group = ring slot
    if (tid < 512) reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p1_tile<DIR, W>(src + t * (1ull << 20), ring + t * (1ull << 20), tile, tw_outer + (size_t)tile * (64 * W), xch,
                    twi, two, tid);
}

template <int DIR, int W>
__global__ __launch_bounds__(32 * W) void k_p2_1m(const v2f *__restrict__ ring, v2f *__restrict__ dst,
                                                  const v2f *__restrict__ tw_inner, float scale, uint32_t xcd_swizzle)
{
    using G = Geom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem);
    v2f *twi = reinterpret_cast<v2f *>(smem + G::XCH_BYTES);
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = xcd_block(xcd_swizzle & 1u);
    const uint32_t tile = bid % G::TILES;
    // bit 1: newest ring slots first (the transforms pass 1 wrote last are the likeliest to still sit in the
    // Infinity Cache when this launch starts)
    const uint64_t t = (xcd_swizzle & 2u) ? (gridDim.x / G::TILES - 1) - bid / G::TILES : bid / G::TILES;
    if (tid < 512) reinterpret_cast<v4f *>(twi)[tid] = reinterpret_cast<const v4f *>(tw_inner)[tid];
    p2_tile<DIR, W, AUX_DEFAULT>(ring + t * (1ull << 20), dst + t * (1ull << 20), tile, scale, xch, twi, tid,
                                 [] { __syncthreads(); });
}

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

template <int W>
static hipError_t setup_w()
{
    using G = Geom<W>;
    const int p1 = G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES, p2 = G::XCH_BYTES + G::TWI_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p1_1m<FWD, W>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, p1);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p1_1m<INV, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, p1);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p2_1m<FWD, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, p2);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p2_1m<INV, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, p2);
    return e;
}

hipError_t launch_p1_gen(int dir, bool out_is_ring, const v2f *src, v2f *dst, const v2f *tw_inner, const v2f *tw_lo,
                         const v2f *tw_hi, uint32_t pitch, uint64_t in_sb, uint64_t out_sb, uint32_t n_transforms,
                         uint32_t xcd_swizzle, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (pitch < 16 || pitch > (1u << 20) || (pitch & (pitch - 1))) return hipErrorInvalidValue;
    const uint64_t blocks = (uint64_t)n_transforms * (pitch / 16);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    using G = Geom<16>;
    if (blocks % 8) xcd_swizzle = 0;  // the XCD mapping needs a grid that is a multiple of 8
    void *args[] = {&src, &dst, &tw_inner, &tw_lo, &tw_hi, &pitch, &in_sb, &out_sb, &xcd_swizzle};
    const void *k = dir == FWD ? (out_is_ring ? reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_SC1>) : reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_NT>))
                               : (out_is_ring ? reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_SC1>) : reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_NT>));
    return hipLaunchKernel(k, dim3((uint32_t)blocks), dim3(512), args, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES, st);
}

hipError_t setup_1m_kernels()
{
    hipError_t e = setup_w<16>();
    if (e == hipSuccess) e = setup_w<32>();
    using G = Geom<16>;
    for (const void *kg : {reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_SC1>), reinterpret_cast<const void *>(&k_p1_gen<FWD, AUX_NT>),
                           reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_SC1>), reinterpret_cast<const void *>(&k_p1_gen<INV, AUX_NT>)})
        if (e == hipSuccess)
            e = hipFuncSetAttribute(kg, hipFuncAttributeMaxDynamicSharedMemorySize, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES);
    const int lds = G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES + 16;
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_1m<FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_1m<INV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}

template <int DIR, int W>
static hipError_t launch_p1_w(const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer, uint32_t n_transforms,
                              uint32_t swz, hipStream_t st)
{
    using G = Geom<W>;
    void *args[] = {&src, &ring, &tw_inner, &tw_outer, &swz};
    return hipLaunchKernel(reinterpret_cast<const void *>(&k_p1_1m<DIR, W>), dim3(n_transforms * G::TILES),
                           dim3(G::THREADS), args, G::XCH_BYTES + G::TWI_BYTES + G::TWO_BYTES, st);
}
template <int DIR, int W>
static hipError_t launch_p2_w(const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t n_transforms, float scale,
                              uint32_t swz, hipStream_t st)
{
    using G = Geom<W>;
    void *args[] = {&ring, &dst, &tw_inner, &scale, &swz};
    return hipLaunchKernel(reinterpret_cast<const void *>(&k_p2_1m<DIR, W>), dim3(n_transforms * G::TILES),
                           dim3(G::THREADS), args, G::XCH_BYTES + G::TWI_BYTES, st);
}

hipError_t launch_p1_1m(int dir, int tile_w, const v2f *src, v2f *ring, const v2f *tw_inner, const v2f *tw_outer,
                        uint32_t n_transforms, uint32_t swz, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (tile_w == 16)
        return dir == FWD ? launch_p1_w<FWD, 16>(src, ring, tw_inner, tw_outer, n_transforms, swz, st)
                          : launch_p1_w<INV, 16>(src, ring, tw_inner, tw_outer, n_transforms, swz, st);
    if (tile_w == 32)
        return dir == FWD ? launch_p1_w<FWD, 32>(src, ring, tw_inner, tw_outer, n_transforms, swz, st)
                          : launch_p1_w<INV, 32>(src, ring, tw_inner, tw_outer, n_transforms, swz, st);
    return hipErrorInvalidValue;
}

hipError_t launch_p2_1m(int dir, int tile_w, const v2f *ring, v2f *dst, const v2f *tw_inner, uint32_t n_transforms,
                        float scale, uint32_t swz, hipStream_t st)
{
    if (n_transforms == 0) return hipSuccess;
    if (tile_w == 16)
        return dir == FWD ? launch_p2_w<FWD, 16>(ring, dst, tw_inner, n_transforms, scale, swz, st)
                          : launch_p2_w<INV, 16>(ring, dst, tw_inner, n_transforms, scale, swz, st);
    if (tile_w == 32)
        return dir == FWD ? launch_p2_w<FWD, 32>(ring, dst, tw_inner, n_transforms, scale, swz, st)
                          : launch_p2_w<INV, 32>(ring, dst, tw_inner, n_transforms, scale, swz, st);
    return hipErrorInvalidValue;
}

}  // namespace fwa
