// kernels_lab_team.hip -- LABORATORY build only (libfft_wgpu_amd_lab.so): k_team, both passes of a two-pass transform in one
// persistent launch, the intermediate kept in the L2 of one XCD (path 8; measured no faster than the per-pass launches).
#include "tile_body.h"

namespace fwa {

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
    // A team forms from workgroups of ONE XCD; with at least 8 x team size workgroups some XCD receives a full team
    // however the blocks are dealt (pigeonhole), and one team alone drains the whole batch.  Fewer could leave every XCD
    // short of a team: every workgroup would give up and the launch would report success without transforming anything.
    if (!ts || max_teams == 0 || n_workgroups < 8 * ts) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ctl, 0, team_ctl_bytes(lg_n, max_teams), st);
    if (e != hipSuccess) return e;
    TeamArgs a{src, dst, slabs, tw_a, tw_lo, tw_hi, tw_c, ctl, batch, max_teams, scale};
    void *args[] = {&a};
    const void *k = dir == FWD ? team_kernel<FWD>(lg_n) : team_kernel<INV>(lg_n);
    return hipLaunchKernel(k, dim3(n_workgroups), dim3(th), args, lds, st);
}


}  // namespace fwa
