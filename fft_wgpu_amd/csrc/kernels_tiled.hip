// kernels_tiled.hip -- (2/3) the building block of the 2- and 3-pass paths: k_tile.
#include "device_common.h"

namespace fwa {

// ---------------------------------------------------------------------------
// k_tile: CW FFTs of length L (64 <= L <= 1024) per workgroup along ONE axis of a multi-dimensional view
// of the transform -- the building block of the 2- and 3-pass paths (n = N1*N2[*N3]).  Same register radix-16
// Stockham stages as k_small16; what differs is addressing:
//   COLS  (strided axis): element i of FFT c at in + i*pitch + c; CW adjacent c = one CW*8-byte segment, so
//         loads and stores are coalesced over c.  Output element o is multiplied by the four-step twiddle
//         W_T^{(col0 + c)*o} = hi[e>>10]*lo[e&1023] and stored at out + o*pitch + c (in place allowed).
//   ROWS_T (last axis): FFT c is a contiguous row at in + c*row_pitch; loads are coalesced along the row,
//         the exchange re-maps threads, and output element o of row c goes to out + o*out_stride + c
//         (CW adjacent rows = one segment): the transposed store that restores natural order.
// LDS: one padded array per FFT (pad(p) = p + p/16, conflict-free over the position index as in k_small16);
// the arrays are PSTR elements apart with PSTR = 17 (mod 32): lanes that differ in the FFT index c (the
// fastest lane index of every stage after the first) then hit distinct banks for b64 writes (16-lane groups)
// and b64 reads (32-lane groups).  With PSTR = L + L/16 (a multiple of 16 for L >= 256) those accesses were
// 8- to 16-way bank conflicts.
// ---------------------------------------------------------------------------
// ROLE: cache policy of the global accesses (measured on the 2^20 pipeline: `nt` on user-buffer accesses and
// write-through `sc1` ring stores): ROLE_FIRST (user buffer -> ring: loads nt, stores sc1), ROLE_MIDDLE
// (ring -> ring: stores sc1), ROLE_LAST (ring -> user buffer: stores nt); BUF = false has no policy bits.
constexpr uint32_t tile_pstr(uint32_t L)
{
    uint32_t p = L + L / 16;
    while (p % 32 != 17) ++p;
    return p;
}

// The tile itself: CW FFTs of length L read at `in`, written at `out` (both already offset to the tile), first FFT of
// the tile = column / row `col0` of its matrix.  AIN / AOUT: cache-policy bits of the global loads / stores.
template <int LGL, int CW, int DIR, int MODE, bool BUF, int AIN, int AOUT>
__device__ __forceinline__ void tile_body(const v2f *in, v2f *out, uint32_t col0, const v2f *__restrict__ tw_l,
                                          const v2f *__restrict__ tw_lo, const v2f *__restrict__ tw_hi, uint64_t pitch,
                                          uint64_t out_stride, float scale, v2f *lds_all, uint32_t tid)
{
    constexpr int L = 1 << LGL;
    constexpr int TPX = L / 16;
    constexpr int NS16 = LGL / 4;
    constexpr int RL = 1 << (LGL % 4);
    constexpr int PSTR = tile_pstr(L);
    auto pad = [](uint32_t p) { return p + (p >> 4); };

    // mapping B (FFT index fastest): coalesces every access whose CW FFTs are adjacent in memory
    const uint32_t cB = tid & (CW - 1), tB = tid / CW;
    // mapping A (position fastest): coalesces along a contiguous row
    const uint32_t cA = tid / TPX, tA = tid % TPX;
    const uint32_t c0 = (MODE == TILE_COLS) ? cB : cA, t0 = (MODE == TILE_COLS) ? tB : tA;
    // Addressing.  BUF (every byte offset of the tile < 2^32, checked by the launcher): buffer loads/stores
    // with one 32-bit per-lane offset and a scalar offset per access -- no 64-bit multiply per element
    // (cdna_hip_programming.md T8); otherwise plain 64-bit pointers (only the largest transforms).
    const uint32_t pitch32 = (uint32_t)pitch, ostride32 = (uint32_t)out_stride;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(in), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFFu, 0x00020000);
    const uint32_t vin = (MODE == TILE_COLS) ? (t0 * pitch32 + c0) * 8 : (c0 * pitch32 + t0) * 8;
    const uint32_t sin_step = (MODE == TILE_COLS) ? (uint32_t)(L / 16) * pitch32 * 8 : (uint32_t)(L / 16) * 8;

    // stage 0: global -> LDS (L >= 64, so there is always a later stage); inputs i = t0 + m*L/16
    {
        v2f *lds = lds_all + c0 * PSTR;
        v2f x[16];
        static_for<0, 16>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if constexpr (BUF) x[m] = buf_load<AIN>(rin, vin, m * sin_step);
            else x[m] = (MODE == TILE_COLS) ? in[(uint64_t)(t0 + m * (L / 16)) * pitch + c0]
                                            : in[(uint64_t)c0 * pitch + t0 + m * (L / 16)];
        });
        fft_reg<16, DIR>(x);
        static_for<0, 16>([&](auto q_) {  // J = 1: s = t0, output position t0*16 + q, twiddle W_L^{t0*q}
            constexpr int q = decltype(q_)::value;
            v2f v = x[brev<16>(q)];
            if constexpr (q != 0) v = cmul_tw<DIR>(v, tw_lookup<L>(tw_l, t0 * q));
            lds[t0 * 17 + q] = v;  // pad(t0*16 + q) = t0*16 + q + t0
        });
    }
    v2f *lds = lds_all + cB * PSTR;
    const uint32_t t = tB;
    // Four-step twiddle (COLS).  Every output of this thread has index o = t + m*TPX, m = 0..15, so
    // W_T^{col*o} = [W^{col*t} * (W^{col*TPX})^(m&3)] * W^{col*TPX*4*(m>>2)}: four table look-ups
    // (hi[e>>10]*lo[e&1023] each) and short products instead of one look-up pair per output.
    v2f pa[4], pb[4];
    if constexpr (MODE == TILE_COLS) {
        const uint32_t col = col0 + cB;
        auto look = [&](uint32_t e) { return cmul(tw_hi[e >> 10], tw_lo[e & 1023]); };
        const v2f wt = look(col * t), p1 = look(col * TPX);
        pa[0] = v2f{1.f, 0.f}; pa[1] = look(col * (4 * TPX)); pa[2] = look(col * (8 * TPX)); pa[3] = cmul(pa[2], pa[1]);
        pb[0] = wt; pb[1] = cmul(wt, p1);
        const v2f p2 = cmul(p1, p1);
        pb[2] = cmul(wt, p2); pb[3] = cmul(pb[2], p1);
    }
    // output m of this thread: index o = t + m*TPX (m is a compile-time constant at every call site)
    auto emit = [&](auto m_, v2f v) {
        constexpr uint32_t m = decltype(m_)::value;
        const uint32_t o = t + m * TPX;
        if constexpr (MODE == TILE_COLS) {
            v = cmul_tw<DIR>(v, cmul(pa[m >> 2], pb[m & 3])) * scale;
            if constexpr (BUF) buf_store<AOUT>(v, rout, (t * pitch32 + cB) * 8, m * (uint32_t)TPX * pitch32 * 8);
            else out[(uint64_t)o * pitch + cB] = v;
        } else {
            v = v * scale;
            if constexpr (BUF) buf_store<AOUT>(v, rout, (t * ostride32 + cB) * 8, m * (uint32_t)TPX * ostride32 * 8);
            else out[(uint64_t)o * out_stride + cB] = v;
        }
    };
    uint32_t J = 16;
    static_for<1, NS16>([&](auto s_) {
        constexpr int st = decltype(s_)::value;
        constexpr bool last = (st == NS16 - 1) && RL == 1;
        __syncthreads();
        v2f x[16];
        static_for<0, 16>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = lds[pad(t + m * (L / 16))]; });
        if constexpr (!last) __syncthreads();
        fft_reg<16, DIR>(x);
        const uint32_t j = t & (J - 1), sJ = t - j;
        static_for<0, 16>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            v2f v = x[brev<16>(q)];
            if constexpr (last) {
                emit(q_, v);  // last stage: J = TPX, s = 0, o = t + q*TPX
            } else {
                if constexpr (q != 0) v = cmul_tw<DIR>(v, tw_lookup<L>(tw_l, sJ * q));
                lds[pad(sJ * 16 + j + q * J)] = v;
            }
        });
        J *= 16;
    });
    if constexpr (RL > 1) {
        // last stage of radix RL < 16: butterflies idx = t + b*TPX, inputs idx + m*L/RL, output q at
        // idx + q*L/RL = t + (b + q*16/RL)*TPX; s = 0, so no stage twiddle
        __syncthreads();
        static_for<0, 16 / RL>([&](auto b_) {
            constexpr int bb = decltype(b_)::value;
            v2f x[RL];
            static_for<0, RL>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                x[m] = lds[pad(t + bb * TPX + m * (L / RL))];
            });
            fft_reg<RL, DIR>(x);
            static_for<0, RL>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                emit(std::integral_constant<int, bb + q * (16 / RL)>{}, x[brev<RL>(q)]);
            });
        });
    }
}


template <int LGL, int CW, int DIR, int MODE, bool BUF, int ROLE>
__global__ __launch_bounds__(((1 << LGL) / 16) * CW) void k_tile(TileArgs a)
{
    constexpr int AOUT = (ROLE == ROLE_FIRST || ROLE == ROLE_MIDDLE) ? AUX_SC1 : (ROLE == ROLE_LAST ? AUX_NT : AUX_DEFAULT);
    constexpr int AIN = (ROLE == ROLE_FIRST) ? AUX_NT : AUX_DEFAULT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware block -> tile mapping: each XCD gets a contiguous run of tiles (see kernels_1m.hip xcd_block)
    const uint32_t bid = a.xcd_swizzle ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t tile = bid % a.tile_count;
    const uint32_t rest = bid / a.tile_count;
    const uint32_t d1 = rest % a.d1_count;
    const uint64_t b = rest / a.d1_count;
    const v2f *in = a.in + b * a.in_sb + d1 * a.in_s1 + tile * a.in_st;
    v2f *out = a.out + b * a.out_sb + d1 * a.out_s1 + tile * a.out_st;
    tile_body<LGL, CW, DIR, MODE, BUF, AIN, AOUT>(in, out, tile * CW, a.tw, a.tw_lo, a.tw_hi, a.pitch, a.out_stride, a.scale,
                                                   reinterpret_cast<v2f *>(smem), threadIdx.x);
}

// ---------------------------------------------------------------------------
// k_team: both passes of a two-pass transform (n = N1*N2 <= 2^18) in ONE persistent launch, the intermediate kept in
// the L2 of one XCD.
//
// Why: in the per-pass launches the intermediate crosses the L2<->fabric boundary twice (ring write + ring read), and
// that traffic costs about as much as the HBM traffic itself (probes: HBM read + cache-resident write together sustain
// 7.4 TB/s; with the written region small enough to stay in L2 the same streams run at 11-12 TB/s,
// profiles/round2/probe_fabric_ring_size.txt).  The intermediate of one transform is n*8 bytes <= 2 MiB, an XCD's L2
// is 4 MiB: if every workgroup that touches a transform's intermediate sits on ONE XCD, the intermediate never has to
// leave that L2.
//
// Structure: workgroups group themselves at run time into TEAMS of N2/CWA workgroups that report the same
// HW_REG_XCC_ID (= share one L2).  A team owns one slab (n elements).  Per transform every member runs pass-A tile
// `member` (column FFTs, user buffer -> slab, plain write-back stores), the team meets at a barrier, every member runs
// pass-C tile `member` (row FFTs, slab -> user buffer, transposed store; slab loads are `sc1`, i.e. L1-bypassing and
// served by the L2).  Teams pull transforms from one global counter.
// Team barrier: no atomics.  Every member publishes {epoch, payload} with ONE plain 8-byte store (the line stays in
// the XCD's L2); one wave polls all members' granules with one `sc1` load instruction, lane i reading member i.  The
// leader's payload carries the next transform index.  (L2-local visibility of plain stores to sc1 loads inside an XCD
// is a property of gfx950's cache hierarchy, not of the HIP memory model: the kernel checks co-location by XCC id and
// the tests compare every output bit with the per-pass path.)
// Progress: a workgroup joins a team only once it is running, so team members are co-resident by construction;
// workgroups that cannot complete a team (or find no slab) leave as soon as all transforms have been claimed.  Spins
// are bounded (2 s) and set ctl[1].
// ---------------------------------------------------------------------------
constexpr int TEAM_CTL_GRANULE_WORD = 64;  // ctl[0] next transform, ctl[1] error, ctl[2] workgroups started, ctl[16+x] workgroups started on XCD x

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// returns the leader's payload of this epoch; *ok = false after a timeout
template <int TEAM>
__device__ __forceinline__ uint32_t team_barrier(unsigned long long *g, uint32_t member, uint32_t epoch, uint32_t payload,
                                                 uint32_t tid, uint32_t *s_word, uint32_t *err)
{
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(&g[member], ((unsigned long long)payload << 32) | epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tid < 64) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(g, 0, TEAM * 8, 0x00020000);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t lead = 0;
        for (;;) {
            v2u v = v2u{epoch, 0u};
            if (tid < TEAM) v = __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, 0, AUX_SC1);
            if (__all((int)(v.x - epoch) >= 0)) { lead = __builtin_amdgcn_readfirstlane(v.y); break; }
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lead = 0xFFFFFFFFu;
                break;
            }
        }
        if (tid == 0) *s_word = lead;
    }
    __syncthreads();
    return *s_word;
}

struct TeamArgs {
    const v2f *src;
    v2f *dst;
    v2f *slabs;                            // [xcd 8][team max_teams][n]
    const v2f *tw_a, *tw_lo, *tw_hi, *tw_c;  // W_N1, four-step lo/hi of domain n, W_N2
    uint32_t *ctl;
    uint32_t batch, max_teams;
    float scale;
};

template <int LGA, int CWA, int LGC, int CWC, int DIR>
__global__ __launch_bounds__(((1 << LGA) / 16) * CWA) void k_team(TeamArgs a)
{
    constexpr uint32_t N1 = 1u << LGA, N2 = 1u << LGC, N = N1 * N2;
    constexpr int THREADS = (N1 / 16) * CWA;
    static_assert(THREADS == (int)(N2 / 16) * CWC, "both passes use every thread");
    constexpr int TEAM = N2 / CWA;
    static_assert(TEAM == (int)(N1 / CWC) && TEAM <= 64, "one tile per member in both passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *lds_all = reinterpret_cast<v2f *>(smem);
    __shared__ uint32_t s_info[4];
    uint32_t *err = a.ctl + 1;

    if (threadIdx.x == 0) {
        const uint32_t x = xcc_id() & 7u;
        const uint32_t slot = __hip_atomic_fetch_add(&a.ctl[16 + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workgroups started, all XCDs
        const uint32_t team = slot / TEAM;
        uint32_t ok = team < a.max_teams;
        if (ok) {
            // Wait for the team to fill.  A member may only give up when the team can never fill, and every member must
            // reach the same verdict: that is the case once ALL workgroups of the grid have started (the per-XCD count
            // is final then).  Full teams never wait for anybody else, so the grid drains and late workgroups do start.
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const uint32_t started = __hip_atomic_load(&a.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t here = __hip_atomic_load(&a.ctl[16 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (here >= (team + 1) * TEAM) break;
                if (started == gridDim.x) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(16);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                    __hip_atomic_fetch_or(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        s_info[0] = x; s_info[1] = slot; s_info[2] = ok;
    }
    __syncthreads();
    if (!s_info[2]) return;
    const uint32_t xcc = s_info[0], team = s_info[1] / TEAM, member = s_info[1] % TEAM;
    const uint32_t tslot = xcc * a.max_teams + team;
    unsigned long long *g = reinterpret_cast<unsigned long long *>(a.ctl + TEAM_CTL_GRANULE_WORD) + (size_t)tslot * TEAM;
    v2f *slab = a.slabs + (size_t)tslot * N;

    uint32_t epoch = 0, t_next = 0;
    if (member == 0 && threadIdx.x == 0) t_next = __hip_atomic_fetch_add(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // keeps LICM from hoisting (and spilling) every lane-constant address of the loop
        // barrier 1: the slab is free (every member finished the previous transform) and everybody learns t
        const uint32_t t = team_barrier<TEAM>(g, member, ++epoch, t_next, tid, &s_info[3], err);
        if (t >= a.batch) break;  // includes the time-out value
        if (member == 0 && tid == 0) t_next = __hip_atomic_fetch_add(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // pass A: FFT over n1 (stride N2) of CWA adjacent columns, user buffer -> slab, twiddle W_n^{col*k1}
        tile_body<LGA, CWA, DIR, TILE_COLS, true, AUX_NT, AUX_DEFAULT>(a.src + (size_t)t * N + member * CWA, slab + member * CWA,
                                                                        member * CWA, a.tw_a, a.tw_lo, a.tw_hi, N2, 0, 1.0f,
                                                                        lds_all, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's slab stores have reached the L2
        // barrier 2: the slab is complete.  The leader republishes the CURRENT index: a member still polling barrier 1
        // may already see the leader's barrier-2 granule (never a later one: nobody passes barrier 2 before every
        // member has arrived at it), so both granules must carry the same payload.
        const uint32_t chk = team_barrier<TEAM>(g, member, ++epoch, t, tid, &s_info[3], err);
        if (chk == 0xFFFFFFFFu) break;
        // pass C: FFT over the contiguous axis of CWC adjacent rows k1, slab -> user buffer at X[k1 + N1*k2]
        tile_body<LGC, CWC, DIR, TILE_ROWS_T, true, AUX_SC1, AUX_NT>(slab + (size_t)member * CWC * N2, a.dst + (size_t)t * N + member * CWC,
                                                                     member * CWC, a.tw_c, nullptr, nullptr, N2, N1, a.scale,
                                                                     lds_all, tid);
    }
}

bool tile_supported(uint32_t lg_l, uint32_t cw)
{
    return (cw == 16 && lg_l >= 6 && lg_l <= 10) || (cw == 32 && lg_l >= 6 && lg_l <= 9);
}

template <int CW, int DIR, int MODE, bool BUF, int ROLE>
static const void *tile_kernel_p(uint32_t lg_l)
{
    switch (lg_l) {
        case 6: return reinterpret_cast<const void *>(&k_tile<6, CW, DIR, MODE, BUF, ROLE>);
        case 7: return reinterpret_cast<const void *>(&k_tile<7, CW, DIR, MODE, BUF, ROLE>);
        case 8: return reinterpret_cast<const void *>(&k_tile<8, CW, DIR, MODE, BUF, ROLE>);
        case 9: return reinterpret_cast<const void *>(&k_tile<9, CW, DIR, MODE, BUF, ROLE>);
        case 10:
            if constexpr (CW == 16) return reinterpret_cast<const void *>(&k_tile<10, CW, DIR, MODE, BUF, ROLE>);
            else return nullptr;
        default: return nullptr;
    }
}
// COLS passes come as first or middle pass, ROWS_T is always the last; the 64-bit-pointer form (BUF = false,
// only above 4-GiB tiles) has no policy bits.
template <int CW, int DIR, int MODE>
static const void *tile_kernel_m(uint32_t lg_l, bool buf, int role)
{
    if (!buf) return tile_kernel_p<CW, DIR, MODE, false, 0>(lg_l);
    if constexpr (MODE == TILE_COLS) {
        if (role == ROLE_MIDDLE) return tile_kernel_p<CW, DIR, MODE, true, ROLE_MIDDLE>(lg_l);
        return tile_kernel_p<CW, DIR, MODE, true, ROLE_FIRST>(lg_l);
    } else {
        return tile_kernel_p<CW, DIR, MODE, true, ROLE_LAST>(lg_l);
    }
}
static const void *tile_kernel(int dir, int mode, uint32_t cw, uint32_t lg_l, bool buf, int role)
{
    if (!tile_supported(lg_l, cw)) return nullptr;
#define FWA_TK(CWV)                                                                                          \
    (dir == FWD ? (mode == TILE_COLS ? tile_kernel_m<CWV, FWD, TILE_COLS>(lg_l, buf, role)                   \
                                     : tile_kernel_m<CWV, FWD, TILE_ROWS_T>(lg_l, buf, role))                \
                : (mode == TILE_COLS ? tile_kernel_m<CWV, INV, TILE_COLS>(lg_l, buf, role)                   \
                                     : tile_kernel_m<CWV, INV, TILE_ROWS_T>(lg_l, buf, role)))
    return cw == 16 ? FWA_TK(16) : FWA_TK(32);
#undef FWA_TK
}
static size_t tile_lds(uint32_t lg_l, uint32_t cw) { return (size_t)cw * tile_pstr(1u << lg_l) * sizeof(v2f); }

// called at plan creation: raises the dynamic-LDS limit of the kernels a plan will launch
hipError_t prepare_tile(uint32_t lg_l, uint32_t cw)
{
    if (!tile_supported(lg_l, cw)) return hipErrorInvalidValue;
    const size_t lds = tile_lds(lg_l, cw);
    if (lds <= 65536) return hipSuccess;
    for (int dir : {FWD, INV})
        for (int mode : {TILE_COLS, TILE_ROWS_T})
            for (int buf = 0; buf < 2; ++buf)
                for (int role : {ROLE_FIRST, ROLE_MIDDLE}) {
                    const void *k = tile_kernel(dir, mode, cw, lg_l, buf != 0, role);
                    if (!k) return hipErrorInvalidValue;
                    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                }
    return hipSuccess;
}

hipError_t launch_tile(int dir, int mode, uint32_t lg_l, const TileArgs &a, uint64_t batch, hipStream_t st)
{
    const uint64_t blocks = batch * a.d1_count * a.tile_count;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one tile?  COLS: L rows of `pitch`; ROWS_T: cw rows of `pitch` in, L outputs of out_stride
    const uint64_t L = 1ull << lg_l, cw = a.cw;
    const uint64_t span_in = (cw * a.pitch + L) * 8, span_out = (L * a.out_stride + cw) * 8;
    const uint64_t span = (mode == TILE_COLS) ? L * a.pitch * 8 + cw * 8 : (span_in > span_out ? span_in : span_out);
    const void *k = tile_kernel(dir, mode, a.cw, lg_l, span < (1ull << 32), (int)a.role);
    if (!k) return hipErrorInvalidValue;
    TileArgs copy = a;
    if (blocks % 8) copy.xcd_swizzle = 0;
    void *args[] = {&copy};
    return hipLaunchKernel(k, dim3((uint32_t)blocks), dim3((uint32_t)((L / 16) * cw)), args, tile_lds(lg_l, a.cw), st);
}


// ---- k_team host side ----
struct TeamCfg { uint32_t lga, cwa, lgc, cwc; };
static bool team_cfg(uint32_t lg_n, TeamCfg *c)
{
    switch (lg_n) {
        case 16: *c = {8, 16, 8, 16}; return true;
        case 17: *c = {8, 32, 9, 16}; return true;
        case 18: *c = {9, 16, 9, 16}; return true;
        default: return false;
    }
}
template <int DIR>
static const void *team_kernel(uint32_t lg_n)
{
    switch (lg_n) {
        case 16: return reinterpret_cast<const void *>(&k_team<8, 16, 8, 16, DIR>);
        case 17: return reinterpret_cast<const void *>(&k_team<8, 32, 9, 16, DIR>);
        case 18: return reinterpret_cast<const void *>(&k_team<9, 16, 9, 16, DIR>);
        default: return nullptr;
    }
}
bool team_supported(uint32_t lg_n) { TeamCfg c; return team_cfg(lg_n, &c); }
void team_geometry(uint32_t lg_n, uint32_t *team_size, uint32_t *threads, size_t *lds_bytes)
{
    TeamCfg c{};
    if (!team_cfg(lg_n, &c)) { *team_size = *threads = 0; *lds_bytes = 0; return; }
    *team_size = (1u << c.lgc) / c.cwa;
    *threads = ((1u << c.lga) / 16) * c.cwa;
    const size_t la = tile_lds(c.lga, c.cwa), lc = tile_lds(c.lgc, c.cwc);
    *lds_bytes = la > lc ? la : lc;
}
size_t team_ctl_bytes(uint32_t lg_n, uint32_t max_teams)
{
    uint32_t ts, th; size_t lds;
    team_geometry(lg_n, &ts, &th, &lds);
    return sizeof(uint32_t) * TEAM_CTL_GRANULE_WORD + (size_t)8 * max_teams * ts * 8;
}
hipError_t prepare_team(uint32_t lg_n)
{
    uint32_t ts, th; size_t lds;
    team_geometry(lg_n, &ts, &th, &lds);
    if (!ts) return hipErrorInvalidValue;
    for (const void *k : {team_kernel<FWD>(lg_n), team_kernel<INV>(lg_n)}) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_team(int dir, uint32_t lg_n, const v2f *src, v2f *dst, v2f *slabs, const v2f *tw_a, const v2f *tw_lo,
                       const v2f *tw_hi, const v2f *tw_c, uint32_t *ctl, uint32_t batch, uint32_t max_teams,
                       uint32_t n_workgroups, float scale, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    uint32_t ts, th; size_t lds;
    team_geometry(lg_n, &ts, &th, &lds);
    if (!ts || max_teams == 0 || n_workgroups < ts) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ctl, 0, team_ctl_bytes(lg_n, max_teams), st);
    if (e != hipSuccess) return e;
    TeamArgs a{src, dst, slabs, tw_a, tw_lo, tw_hi, tw_c, ctl, batch, max_teams, scale};
    void *args[] = {&a};
    const void *k = dir == FWD ? team_kernel<FWD>(lg_n) : team_kernel<INV>(lg_n);
    return hipLaunchKernel(k, dim3(n_workgroups), dim3(th), args, lds, st);
}

}  // namespace fwa
