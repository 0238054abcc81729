// plan.cpp -- the four plan objects of the reference (src/processor.rs: Forward :7-159, Inverse :231-341,
// Normalize :409-505, Onlyinverse :566-670) behind fwa_plan_create / fwa_plan_exec / fwa_plan_destroy.
//
// None of the reference's wgpu plumbing survives: a plan owns (or shares through the context's plan cache) its twiddle
// tables and scratch, exec only enqueues kernels on the caller's stream and allocates nothing.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "internal.h"

using namespace fwa_int;

namespace fwa_int {

// Path and per-pass FFT lengths (log2) for a transform length; shared by fwa_plan_create and fwa_describe_path.
// `batch` separates two regimes (profiles/round2/sweep_small_batch_latency.jsonl):
//  * throughput (n * batch > 2^20 samples): few passes of fat tiles -- a 1024-point first pass (k_p1_gen / the 2^20
//    pipeline, 64 KiB tiles of 512 threads) and 32-point-per-thread rows;
//  * latency (at most 2^20 samples in flight, or a single 2^21 / 2^22 transform, or fewer than FEW_1M transforms of
//    2^20): fat tiles leave most of the 256 CUs idle (one 2^16 transform = FOUR 1024 x 16 tiles), so the plan uses the
//    smallest tiles instead -- balanced two passes up to 2^17, balanced three passes of 64/128-point tiles above
//    (2^16 x 1: 11.9 us against 16.2; 2^18 x 1: 12.7 against 18.4; 2^20 x 1: 19 against 24).
constexpr uint64_t FEW_1M = 4;
int64_t choose_path(uint32_t n, uint64_t batch, uint32_t lf[3], bool *colsw)
{
    lf[0] = lf[1] = lf[2] = 0;
    if (colsw) *colsw = false;
    const uint32_t lg = ilog2(n);
    if (n == 1) return PATH_IDENTITY;
    if (n <= 32768) { lf[0] = lg; return PATH_SMALL; }
    const bool few = (lg < 20 && batch <= ((1ull << 20) >> lg)) || (lg == 20 && batch < FEW_1M)
        || ((lg == 21 || lg == 22) && batch == 1);
    if (n == (1u << 20) && !few) { lf[0] = lf[1] = 10; return PATH_TWOPASS_1M; }
    if (n <= (1u << 30)) {
        // factors of 64..1024 each, 2048 for the rows of a two-pass plan (re-tunable: key "factors").  Throughput
        // regime: two passes up to 2^19 and at 2^21 .. 2^23 (2048 / 4096-point passes), three otherwise; a 1024-point
        // first pass (k_p1_gen) wherever the other factors stay >= 64, measured faster than a balanced split except
        // at 2^22 (level) (profiles/round2/p1gen_sweep.jsonl, factor_sweep.jsonl, sweep_rows32.jsonl). Round 3
        // (profiles/round3/sweep_colsw_32GiB.jsonl, sweep_factors_24_28_colsw.jsonl): short columns in wide tiles
        // (k_colsw: 256 x 64 / 512 x 32, 512- / 256-byte row segments) beat the 1024 x 16 tile of k_p1_gen as pass A
        // wherever the last pass keeps <= 1024-point rows: 2^16 .. 2^19 + 7-9 %, three-pass sizes 2^24 .. 2^28 + 2-16 %
        // 1024 x 4096: k_p1_gen + k_rows32 (8 rows of 4096 per workgroup)
        if (!few && lg == 22) { lf[0] = 10; lf[1] = 12; }
        else if (!few && lg == 23) { lf[0] = 11; lf[1] = 12; }  // 2048 x 4096: k_cols32 + k_rows32
        else if (few && lg <= 17) { lf[0] = lg / 2; lf[1] = lg - lf[0]; }
        else if (!few && lg <= 18) { lf[0] = 8; lf[1] = lg - 8; if (colsw) *colsw = true; }   // 256 x (256 .. 1024)
        else if (!few && lg == 19) { lf[0] = 9; lf[1] = 10; if (colsw) *colsw = true; }       // 512 x 1024
        else if (!few && lg == 21) { lf[0] = 10; lf[1] = lg - 10; }
        else if (!few && lg >= 24 && lg <= 28) {
            lf[0] = 9; lf[1] = (lg - 9) / 2; lf[2] = lg - 9 - lf[1];
            if (colsw) *colsw = true;
        }
        else if (!few && lg >= 29) { lf[0] = 10; lf[1] = (lg - 10) / 2; lf[2] = lg - 10 - lf[1]; }
        // 16.4 us against 18.2 for 64 x 128 x 128 (sweep_factor_permutations_batch1.jsonl)
        else if (few && lg == 20) { lf[0] = lf[1] = 6; lf[2] = 8; }
        else for (uint32_t i = 0; i < 3; ++i) lf[i] = lg / 3 + (i >= 3 - lg % 3 ? 1 : 0);
        return PATH_TILED;
    }
    return PATH_R2_GLOBAL;
}

// tile width (FFTs per workgroup) of pass i of the tiled path
uint32_t pass_cw(const fwa_plan *, uint32_t) { return 16u; }

// Everything a plan needs for its path: kernel attributes (once per context), twiddle tables (shared through the
// context's plan cache) and, on the pipelined paths, the ring + internal streams with the default geometry.
// Two chains (pass A of one group beside pass C of another) pay off once each chain has a few groups to run; with fewer
// than 4 groups in all, the launches of the two chains only compete (2^20 x 32: 8.0 us per transform on two chains, 6.1
// on one; x 64: 6.2 against 6.5; 2^18 x 128: 1.68 against 1.51; profiles/round2/sweep_mid_batch_chains.jsonl).
int64_t default_chains(uint64_t batch, int64_t group)
{
    const uint64_t n_groups = group > 0 ? (batch + (uint64_t)group - 1) / (uint64_t)group : 0;
    return n_groups >= 4 ? 2 : 1;
}

int32_t setup_path(fwa_plan *p)
{
    fwa_ctx *ctx = p->ctx;
    const uint32_t fft_len = p->n;
    if (p->path == PATH_SMALL && fft_len > 4096 && !ctx->setup_small_done) {
        hipError_t e = fwa::setup_small_kernels();
        if (e != hipSuccess) return fail_hip(ctx, e, "hipFuncSetAttribute(max dynamic LDS)");
        ctx->setup_small_done = true;
    }
    if ((p->path == PATH_TWOPASS_1M || p->path == PATH_RING_1M || (p->path == PATH_TILED && p->lf[0] == 10))
        && !ctx->setup_1m_done) {
        hipError_t e = fwa::setup_1m_kernels();
#ifdef FWA_LAB
        if (e == hipSuccess) e = fwa::setup_lab_1m_kernels();
#endif
        if (e != hipSuccess) return fail_hip(ctx, e, "hipFuncSetAttribute(max dynamic LDS)");
        ctx->setup_1m_done = true;
    }
    if (p->path == PATH_TILED) {
        const uint32_t nf = p->lf[2] ? 3 : 2;
        for (uint32_t i = 0; i < nf; ++i)
        {
            if (nf == 2 && i == 1 && fwa::rows32_supported(p->lf[1])) {
                hipError_t re = fwa::prepare_rows32(p->lf[1]);
                if (re != hipSuccess) return fail_hip(ctx, re, "hipFuncSetAttribute(max dynamic LDS)");
            }
            if (i == 0 && fwa::cols32_supported(p->lf[0])) {
                hipError_t ce = fwa::prepare_cols32(p->lf[0]);
                if (ce != hipSuccess) return fail_hip(ctx, ce, "hipFuncSetAttribute(max dynamic LDS)");
            }
            if (i == 0 && fwa::colsw_supported(p->lf[0])) {
                hipError_t ce = fwa::prepare_colsw(p->lf[0]);
                if (ce != hipSuccess) return fail_hip(ctx, ce, "hipFuncSetAttribute(max dynamic LDS)");
            }
            if (p->lf[i] > 10) continue;  // 2048 / 4096-point passes: k_cols32 / k_rows32 only
            hipError_t pe = fwa::prepare_tile(p->lf[i], 16);
            if (pe != hipSuccess) return fail_hip(ctx, pe, "hipFuncSetAttribute(max dynamic LDS)");
        }
    }
    // tables: shared by every plan of this (length, path, factorisation) on the context
    const uint32_t sig = p->lf[0] | (p->lf[1] << 8) | (p->lf[2] << 16);
    const auto key = std::make_tuple(fft_len, p->path == PATH_RING_1M ? (int64_t)PATH_TWOPASS_1M : p->path, sig);
    // The tables of the NEW path / factorisation are held locally and handed to the plan only once its pipeline has
    // been built: a failed re-tune (e.g. no memory for the new ring) leaves the plan with the tables of the factors it
    // keeps (the callers restore path and factors).
    std::shared_ptr<Tables> tb;
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        tb = it->second;
        ++ctx->n_table_hits;
    } else {
        tb = std::make_shared<Tables>();
        int32_t st = build_tables(ctx, fft_len, std::get<1>(key), p->lf, tb.get());
        if (st) return st;
        ++ctx->n_table_builds;
        ctx->tables.emplace(key, tb);
    }
    int32_t st = FWA_OK;
    if (p->path == PATH_TILED) {
        // the intermediate of a group of transforms lives in a ring slab of 128 MiB per chain (two chains = the
        // 256-MiB Infinity Cache; group sweep in profiles/round1/h_tiled_group_sweep.jsonl)
        const uint64_t per = (uint64_t)fft_len * sizeof(v2f);
        int64_t g = (int64_t)((128ull << 20) / per);
        if (g < 1) g = 1;
        st = build_pipeline(p, g, default_chains(p->batch, g));
    } else if (p->path == PATH_TWOPASS_1M) {
        st = build_pipeline(p, 16, default_chains(p->batch, 16));  // 16 transforms = 1024 tiles per launch
    } else if (p->path == PATH_RING_1M) {
        st = build_pipeline(p, 0, 0);
    }
    if (st) return st;
    p->tb = tb;
    return FWA_OK;
}

size_t ctl_bytes(const fwa_plan *p)
{
#ifdef FWA_LAB
    return fwa::ring_ctl_bytes(p->batch);
#else
    (void)p;
    return 0;
#endif
}

static fwa_buf *result_buffer(fwa_plan *p)
{
    // processor.rs:153-157, :335-339, :664-668
    return (p->lg % 2 == 0) ? p->src : p->second;
}

// Block -> tile map of the tiled plans' kernels (xcd_map, device_common.h) when the caller has not set "xcd_swizzle":
// measured per size at the 32-GiB footprint, three interleaved runs (profiles/round4/sweep_tiled_block_maps.jsonl):
// the k_colsw plans gain 2-4 % from XCD-contiguous runs (2^17 .. 2^19: bit 0; 2^16 and 1024 x 2048: with the CU
// pairs, bits 0 + 2); 2^22 and up lose 1-10 %.
uint32_t tiled_swizzle_default(const fwa_plan *p)
{
    if (p->lf[2]) return 0u;
    if (p->colsw && p->lg == 16) return 5u;
    if (p->colsw && p->lg >= 17 && p->lg <= 19) return 1u;
    if (p->lg == 21 && p->lf[0] == 10 && p->lf[1] == 11) return 5u;
    return 0u;
}

// Run `body(group index, stream, chain index)` for every group, alternating over the plan's internal streams,
// forked from and joined back to the caller's stream with events.
template <class Body>
static int32_t run_groups(fwa_plan *plan, hipStream_t st, Body body)
{
    fwa_ctx *ctx = plan->ctx;
    const uint64_t G = (uint64_t)plan->group, n_groups = (plan->batch + G - 1) / G;
    const size_t ns = plan->istreams.size();
    if (plan->batch && !plan->ring)
        return fail(ctx, FWA_ERR_INVALID_ARG, "plan has no scratch ring (a failed re-tune?)");
    if (ns) {
        HIP_TRY(ctx, hipEventRecord(plan->ev_fork, st));
        for (size_t i = 0; i < ns; ++i) HIP_TRY(ctx, hipStreamWaitEvent(plan->istreams[i], plan->ev_fork, 0));
    }
    hipError_t e = hipSuccess;
    for (uint64_t g = 0; g < n_groups && e == hipSuccess; ++g) {
        const size_t c = ns ? (size_t)(g % ns) : 0;
        const uint64_t cnt = (plan->batch - g * G < G) ? plan->batch - g * G : G;
#ifdef FWA_LAB
        if (plan->inject_fail_group == (int64_t)g) { plan->inject_fail_group = -1; e = hipErrorLaunchFailure; break; }
#endif
        e = body(g, cnt, ns ? plan->istreams[c] : st, c);
    }
    // Join the chains back to the caller's stream ALSO when a launch failed: the groups enqueued before the failure
    // keep running on the chains, and whatever the caller enqueues next on `st` (a copy of the partial result, the
    // free of the buffer) must be ordered behind them.
    hipError_t je = hipSuccess;
    for (size_t i = 0; i < ns; ++i) {
        hipError_t r = hipEventRecord(plan->idone[i], plan->istreams[i]);
        if (r == hipSuccess) r = hipStreamWaitEvent(st, plan->idone[i], 0);
        if (r != hipSuccess && je == hipSuccess) je = r;
    }
    plan->last_stream = st;
    plan->ran_on_stream = true;
    if (e != hipSuccess) return fail_hip(ctx, e, "kernel launch", FWA_ERR_LAUNCH);
    if (je != hipSuccess) return fail_hip(ctx, je, "hipEventRecord/hipStreamWaitEvent (join of the chain streams)");
    return FWA_OK;
}


// ---- the tiled path (PATH_TILED): n = N1*N2[*N3]; index n = (n1*N2 + n2)*N3 + n3, k = k1 + N1*(k2 + N2*k3) ----
//   pass A: FFT over n1 (columns, four-step twiddle W_n), user buffer -> ring slab
//   pass B: FFT over n2 per k1 (columns, twiddle W_{N2*N3}), in place in the slab            [three factors only]
//   pass C: FFT over the contiguous axis with the transposed store, slab -> result buffer
// What the passes of one exec share: factor sizes, the block -> tile map and the slab layout pass A leaves for pass C.
struct TiledShape {
    uint64_t N, N1, N2, N3;
    bool three;
    uint32_t swizzle;   // xcd_map bits handed to every kernel
    // k_colsw writes the slab tile-contiguously ([tile][k1][ring_cw]) when the last pass (k_rows32) can read that layout
    // back; 0 = matrix layout
    uint32_t ring_cw;
};

// One pass of a tiled exec: which kernel, and every launch argument except the group's pointers and transform count.
// Built once per fwa_plan_exec by pass_a / pass_b / pass_c, immutable afterwards, called once per group.
struct TiledPass {
    enum Kernel { NONE, COLSW, COLS32, P1_GEN, TILE_COLS_K, ROWS32, TILE_ROWS_K } kernel = NONE;
    int dir = fwa::FWD;
    uint32_t lg_l = 0;                 // log2 of this pass's FFT length
    const v2f *tw = nullptr, *tw_lo = nullptr, *tw_hi = nullptr;
    uint32_t pitch = 0;                // columns of the n1 x pitch matrix (COLSW, COLS32, P1_GEN)
    uint32_t n1 = 0;                   // ROWS32: rows of the matrix
    uint64_t sb = 0;                   // elements between transforms, in and out
    float scale = 1.0f;
    uint32_t swizzle = 0, ring_cw = 0;
    fwa::TileArgs ta{};                // TILE_COLS_K / TILE_ROWS_K: complete except in / out
    hipError_t operator()(const v2f *in, v2f *out, uint64_t cnt, hipStream_t s) const
    {
        switch (kernel) {
            case COLSW:
                return fwa::launch_colsw(dir, lg_l, true, ring_cw != 0, in, out, tw, tw_lo, tw_hi, pitch, sb, sb,
                                         (uint32_t)cnt, swizzle, s);
            case COLS32:
                return fwa::launch_cols32(dir, lg_l, true, in, out, tw, tw_lo, tw_hi, pitch, sb, sb, (uint32_t)cnt,
                                          swizzle, s);
            case P1_GEN:
                return fwa::launch_p1_gen(dir, true, in, out, tw, tw_lo, tw_hi, pitch, sb, sb, (uint32_t)cnt, swizzle, s);
            case ROWS32:
                return fwa::launch_rows32(dir, lg_l, in, out, tw, n1, sb, sb, (uint32_t)cnt, scale, swizzle, ring_cw, s);
            case TILE_COLS_K:
            case TILE_ROWS_K: {
                fwa::TileArgs t = ta;
                t.in = in; t.out = out;
                return fwa::launch_tile(dir, kernel == TILE_COLS_K ? fwa::TILE_COLS : fwa::TILE_ROWS_T, lg_l, t, cnt, s);
            }
            case NONE: break;
        }
        return hipSuccess;
    }
};

static bool tiled_uses_colsw(const fwa_plan *p)
{
    return p->colsw && fwa::colsw_supported(p->lf[0]) && p->lg <= 28;
}

static TiledShape tiled_shape(const fwa_plan *p)
{
    TiledShape sh{};
    sh.three = p->lf[2] != 0;
    sh.N = p->n;
    sh.N1 = 1ull << p->lf[0]; sh.N2 = 1ull << p->lf[1]; sh.N3 = sh.three ? (1ull << p->lf[2]) : 1;
    sh.swizzle = p->xcd_swizzle < 0 ? tiled_swizzle_default(p) : (uint32_t)p->xcd_swizzle;
    const bool ring = tiled_uses_colsw(p) && p->tile_ring && !sh.three
        && fwa::rows32_ring_supported(p->lf[1], fwa::colsw_width(p->lf[0]));
    sh.ring_cw = ring ? fwa::colsw_width(p->lf[0]) : 0u;
    return sh;
}

// pass A: columns of length N1 at pitch N / N1, user buffer -> slab
static TiledPass pass_a(const fwa_plan *p, const TiledShape &sh, const Tables &tb, int dir)
{
    TiledPass ps;
    ps.dir = dir; ps.lg_l = p->lf[0]; ps.sb = sh.N; ps.swizzle = sh.swizzle; ps.ring_cw = sh.ring_cw;
    ps.pitch = (uint32_t)(sh.N / sh.N1);
    ps.tw = tb.tw_l[0]; ps.tw_lo = tb.tw_lo1; ps.tw_hi = tb.tw_hi1;
    ps.kernel = tiled_uses_colsw(p)                             ? TiledPass::COLSW     // 256 x 64 / 512 x 32 column tiles
                : p->lf[0] > 10                                  ? TiledPass::COLS32    // 2048-point columns
                : (p->lf[0] == 10 && p->p1_gen && tb.tw_inner)   ? TiledPass::P1_GEN    // the 2^20 column kernel at run-time pitch
                                                                 : TiledPass::TILE_COLS_K;
    switch (ps.kernel) {
        case TiledPass::P1_GEN:
            ps.tw = tb.tw_inner;
            break;
        case TiledPass::TILE_COLS_K: {
            const uint32_t cw = pass_cw(p, 0);
            fwa::TileArgs &t = ps.ta;
            t.tw = tb.tw_l[0]; t.tw_lo = tb.tw_lo1; t.tw_hi = tb.tw_hi1;
            t.scale = 1.0f; t.cw = cw; t.role = fwa::ROLE_FIRST; t.xcd_swizzle = sh.swizzle;
            t.in_sb = t.out_sb = sh.N; t.in_s1 = t.out_s1 = 0; t.in_st = t.out_st = cw;
            t.pitch = sh.N / sh.N1; t.out_stride = 0; t.d1_count = 1; t.tile_count = (uint32_t)(sh.N / sh.N1 / cw);
            break;
        }
        default: break;
    }
    return ps;
}

// pass B (three factors): columns of length N2 at pitch N3 inside every k1-plane, in place in the slab
static TiledPass pass_b(const fwa_plan *p, const TiledShape &sh, const Tables &tb, int dir)
{
    TiledPass ps;
    if (!sh.three) return ps;   // kernel == NONE
    ps.kernel = TiledPass::TILE_COLS_K;
    ps.dir = dir; ps.lg_l = p->lf[1];
    const uint32_t cw = pass_cw(p, 1);
    fwa::TileArgs &t = ps.ta;
    t.tw = tb.tw_l[1]; t.tw_lo = tb.tw_lo_b; t.tw_hi = tb.tw_hi_b;
    t.scale = 1.0f; t.cw = cw; t.role = fwa::ROLE_MIDDLE; t.xcd_swizzle = sh.swizzle;
    t.in_sb = t.out_sb = sh.N; t.in_s1 = t.out_s1 = sh.N2 * sh.N3; t.in_st = t.out_st = cw;
    t.pitch = sh.N3; t.out_stride = 0; t.d1_count = (uint32_t)sh.N1; t.tile_count = (uint32_t)(sh.N3 / cw);
    return ps;
}

// pass C: rows of the last axis, adjacent k1 per tile, transposed store into the result buffer (+ the 1/n of Inverse)
static TiledPass pass_c(const fwa_plan *p, const TiledShape &sh, const Tables &tb, int dir, float scale)
{
    TiledPass ps;
    const uint32_t li = sh.three ? 2 : 1;
    ps.dir = dir; ps.lg_l = p->lf[li]; ps.sb = sh.N; ps.scale = scale; ps.swizzle = sh.swizzle; ps.ring_cw = sh.ring_cw;
    ps.tw = tb.tw_l[li]; ps.n1 = (uint32_t)sh.N1;
    const bool rows32 = !sh.three && p->lg <= 28 && fwa::rows32_supported(p->lf[1])
        && (p->rows32 || p->lf[1] > 10 || sh.ring_cw);   // 2048 / 4096-point rows and the tile ring exist only there
    ps.kernel = rows32 ? TiledPass::ROWS32 : TiledPass::TILE_ROWS_K;
    switch (ps.kernel) {
        case TiledPass::TILE_ROWS_K: {
            const uint32_t cw = pass_cw(p, li);
            fwa::TileArgs &t = ps.ta;
            t.tw = tb.tw_l[li]; t.tw_lo = nullptr; t.tw_hi = nullptr;
            t.scale = scale; t.cw = cw; t.role = fwa::ROLE_LAST; t.xcd_swizzle = sh.swizzle;
            t.in_sb = t.out_sb = sh.N;
            t.pitch = sh.N / sh.N1;  // distance between the rows k1 and k1 + 1
            t.in_st = cw * (sh.N / sh.N1); t.out_st = cw; t.tile_count = (uint32_t)(sh.N1 / cw);
            if (sh.three) { t.d1_count = (uint32_t)sh.N2; t.in_s1 = sh.N3; t.out_s1 = sh.N1; t.out_stride = sh.N1 * sh.N2; }
            else { t.d1_count = 1; t.in_s1 = t.out_s1 = 0; t.out_stride = sh.N1; }
            break;
        }
        default: break;
    }
    return ps;
}


}  // namespace fwa_int

extern "C" {

// ---- plans ----------------------------------------------------------------
int32_t fwa_plan_destroy(fwa_plan *plan)
{
    if (!plan) return FWA_OK;
    (void)hipSetDevice(plan->ctx->device);
    // work of this plan may still be in flight on the caller's stream; the pooled ring must not be handed to the next
    // plan before it has drained (hipFree would have synchronised implicitly).  Only THIS plan's last exec is waited
    // for, through a marker on the stream that exec was enqueued on -- other streams and contexts keep running (a
    // device-wide synchronise here stalled them all and is illegal while any stream captures a graph).  An exec that
    // was captured into a graph enqueued nothing real: the plan must outlive the graphs that replay it.
    if (plan->frozen && plan->ring) {
        hipEvent_t ev = nullptr;
        bool waited = false;
        // only a stream that is known to be alive can take the marker: the null stream, or a stream of this context's
        // own making that has not been destroyed (HIP does not validate stream handles); otherwise the device-wide
        // wait below
        const auto &us = plan->ctx->user_streams;
        const bool alive = plan->last_stream == nullptr
            || std::find(us.begin(), us.end(), plan->last_stream) != us.end();
        if (plan->ran_on_stream && alive && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            // everything this plan enqueued on its last caller stream precedes this marker
            if (hipEventRecord(ev, plan->last_stream) == hipSuccess) waited = hipEventSynchronize(ev) == hipSuccess;
            (void)hipEventDestroy(ev);
        }
        if (!waited) {
            (void)hipGetLastError();
            // the stream is gone or not ours (fwa_stream_wrap), or a laboratory persistent path
            (void)hipDeviceSynchronize();
        }
    }
    Pipeline pl = take_pipeline(plan);
    destroy_pipeline_objects(plan->ctx, pl, true);
    if (plan->ring_ctl) (void)hipFree(plan->ring_ctl);
    if (plan->tw_half_private) (void)hipFree(plan->tw_half_private);
    if (plan->second_owned && plan->own_second.p) (void)hipFree(plan->own_second.p);
    delete plan;
    return FWA_OK;
}

int32_t fwa_plan_create(fwa_ctx *ctx, int32_t kind, uint32_t fft_len, fwa_buf *src, fwa_buf *src2_or_null,
                        fwa_plan **out)
{
    if (!out) return fail(ctx, FWA_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx || !src) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/src is NULL");
    for (const fwa_buf *b : {(const fwa_buf *)src, (const fwa_buf *)src2_or_null}) {
        if (b && !b->ctx) return fail(ctx, FWA_ERR_INVALID_ARG, "a buffer belongs to a context that has been destroyed");
        if (b && b->ctx->device != ctx->device)
            return fail(ctx, FWA_ERR_INVALID_ARG, "a buffer lives on another device than the plan's context");
    }
    if (kind < FWA_FORWARD || kind > FWA_NORMALIZE) return fail(ctx, FWA_ERR_INVALID_ARG, "unknown plan kind");
    if (!is_pow2(fft_len)) return fail(ctx, FWA_ERR_INVALID_ARG, "fft_len must be a power of two >= 1");
    if (fft_len > (1u << 30)) return fail(ctx, FWA_ERR_UNSUPPORTED, "fft_len above 2^30 is not supported");
    const uint64_t tbytes = (uint64_t)fft_len * 8;
    if (src->bytes % tbytes != 0)
        return fail(ctx, FWA_ERR_INVALID_ARG, "buffer size is not a multiple of 8*fft_len bytes");
    const bool needs_src2 = (kind == FWA_INVERSE_UNSCALED || kind == FWA_NORMALIZE);
    if (needs_src2 && !src2_or_null)
        return fail(ctx, FWA_ERR_INVALID_ARG, "this plan kind needs a caller-supplied second buffer");
    if (!needs_src2 && src2_or_null)
        return fail(ctx, FWA_ERR_INVALID_ARG, "this plan kind owns its second buffer; pass NULL");
    if (src2_or_null && src2_or_null->bytes != src->bytes)
        return fail(ctx, FWA_ERR_INVALID_ARG, "second buffer must have the size of the first");
    if (src2_or_null && src2_or_null->p == src->p && src->bytes)
        return fail(ctx, FWA_ERR_INVALID_ARG, "the two buffers must be distinct");
    if (reinterpret_cast<uintptr_t>(src->p) & 15)
        return fail(ctx, FWA_ERR_INVALID_ARG, "buffer must be 16-byte aligned");

    const auto t_begin = std::chrono::steady_clock::now();
    USE_DEVICE(ctx);
    fwa_plan *p = new (std::nothrow) fwa_plan;
    if (!p) return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    p->ctx = ctx; p->kind = kind; p->n = fft_len; p->lg = ilog2(fft_len);
    p->batch = src->bytes / tbytes;
    p->src = src; p->second = src2_or_null;

    int32_t st = FWA_OK;
    auto bail = [&](int32_t s) { fwa_plan_destroy(p); return s; };
    auto done = [&]() {
        ctx->last_plan_create_us =
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count();
        *out = p;
        return FWA_OK;
    };

    if (kind == FWA_NORMALIZE) {
        p->path = PATH_NORMALIZE;
        return done();
    }

    {
        bool cw = false;
        p->path = choose_path(fft_len, p->batch, p->lf, &cw);
        p->colsw = cw;
    }

    // Forward/Inverse own their ping-pong partner (processor.rs:34-41,261-269).  It is only
    // materialised when the result must land there (odd log2 n) or the path ping-pongs.
    const bool odd = (p->lg & 1) != 0;
    const bool need_second = odd || p->path == PATH_R2_GLOBAL;
    if (!p->second && need_second && src->bytes) {
        hipError_t e = hipMalloc(&p->own_second.p, src->bytes);
        if (e != hipSuccess) return bail(fail_hip(ctx, e, "hipMalloc(second buffer)"));
        p->own_second.ctx = ctx; p->own_second.bytes = src->bytes; p->own_second.owned = false;
        p->second = &p->own_second;
        p->second_owned = true;
    } else if (!p->second) {
        // even log2 n on an in-place path: the reference would still own a buffer_b; we keep a
        // zero-sized handle so the result rule never dereferences NULL.
        p->own_second.ctx = ctx;
        p->second = &p->own_second;
    }

    st = setup_path(p);
    if (st) return bail(st);
    return done();
}

int32_t fwa_plan_exec(fwa_plan *plan, fwa_stream *stream, fwa_buf **result)
{
    if (!plan) return fail(nullptr, FWA_ERR_INVALID_ARG, "plan is NULL");
    fwa_ctx *ctx = plan->ctx;
    USE_DEVICE(ctx);
    hipStream_t st = raw(stream);
    plan->frozen = true;
    const uint64_t total = plan->batch * (uint64_t)plan->n;
    hipError_t e = hipSuccess;

    if (plan->kind == FWA_NORMALIZE) {
        // processor.rs:433-439: (a, b) = (buffer1, buffer2) if log2 n even else (buffer2, buffer1); returns b
        fwa_buf *a = (plan->lg % 2 == 0) ? plan->src : plan->second;
        fwa_buf *b = (plan->lg % 2 == 0) ? plan->second : plan->src;
        e = fwa::launch_scale(static_cast<const v2f *>(a->p), static_cast<v2f *>(b->p), total,
                              1.0f / (float)plan->n, st);
        if (e != hipSuccess) return fail_hip(ctx, e, "normalize launch", FWA_ERR_LAUNCH);
        if (result) *result = b;
        return FWA_OK;
    }

    const int dir = (plan->kind == FWA_FORWARD) ? fwa::FWD : fwa::INV;
    const float scale = (plan->kind == FWA_INVERSE_SCALED) ? 1.0f / (float)plan->n : 1.0f;  // ifft.wgsl:65-74
    fwa_buf *res = result_buffer(plan);
    if (result) *result = res;
    if (total == 0) return FWA_OK;
    v2f *a = static_cast<v2f *>(plan->src->p);
    v2f *b = static_cast<v2f *>(plan->second->p);
    v2f *out = static_cast<v2f *>(res->p);
    const Tables &tb = *plan->tb;
    const uint64_t N = plan->n, G = (uint64_t)plan->group;

    switch (plan->path) {
        case PATH_IDENTITY:
            if (scale != 1.0f) e = fwa::launch_scale(a, a, total, scale, st);
            break;
        case PATH_SMALL:
#ifdef FWA_LAB
            // laboratory kernels (A/B): the direct 16-point kernels (small_reg = 3, 16 <= n <= 4096), with the
            // wavefront-shuffle exchange at 32 / 64 / 128 (small_reg = 2, which leaves n >= 512 to k_small32)
            if (plan->small_reg != 1 && plan->n >= 16 && plan->n <= 4096 && (plan->small_reg == 3 || plan->n < 512)) {
                e = fwa::launch_small16(dir, a, out, tb.tw_half, plan->n, plan->batch, scale, plan->small_reg == 2, st);
                break;
            }
#endif
            if (plan->n <= 256)
                e = fwa::launch_chunk(dir, a, out, tb.tw_half, plan->n, plan->batch, scale, st);
            else
                e = fwa::launch_small32(dir, a, out, tb.tw_half, plan->n, plan->batch, scale, st);
            break;
        case PATH_R2_GLOBAL: {
            const v2f *tw = plan->tw_half_private ? plan->tw_half_private : tb.tw_half;
            for (uint32_t s = 0; s < plan->lg && e == hipSuccess; ++s) {
                const v2f *from = (s % 2 == 0) ? a : b;
                v2f *to = (s % 2 == 0) ? b : a;
                e = fwa::launch_r2_stage(dir, from, to, tw, plan->n, s, plan->batch, (s + 1 == plan->lg) ? scale : 1.0f,
                                         st);
            }
            break;
        }
        case PATH_TWOPASS_1M: {
            // in place at group granularity: 2^20 has even log2, the result buffer is src (processor.rs:153-157)
            const v2f *two = tb.tw_outer;
            // default: XCD-contiguous tiles + adjacent tiles on the two residents of a CU (bit 2: + 4-8 % for
            // launches that have the chip to themselves, + 0.6 % with two chains in flight: tile_1m.h xcd_block,
            // profiles/round4/sweep_pair_map_two_chains.jsonl)
            const uint32_t swz = plan->xcd_swizzle < 0 ? 5u : (uint32_t)plan->xcd_swizzle;
            return run_groups(plan, st, [&](uint64_t g, uint64_t cnt, hipStream_t s, size_t c) {
                // ring region of this chain: transform i -> slot i (ring_rotate > 1, laboratory: successive groups of a
                // chain walk through ring_rotate such regions)
                const uint64_t round = g / (plan->istreams.empty() ? 1 : plan->istreams.size());
                v2f *slab = plan->ring + (round % (uint64_t)plan->ring_rotate) *
                                             (uint64_t)plan->n_streams * G * N + (uint64_t)c * G * N;
                hipError_t le = fwa::launch_p1_1m(dir, a + g * G * N, slab, tb.tw_inner, two, (uint32_t)cnt, swz, s);
                if (le != hipSuccess) return le;
                return fwa::launch_p2_1m(dir, slab, out + g * G * N, tb.tw_inner, (uint32_t)cnt, scale, swz, s);
            });
        }
#ifdef FWA_LAB
        case PATH_RING_1M: {
            if (!plan->ring || !plan->ring_ctl)
                return fail(ctx, FWA_ERR_INVALID_ARG, "plan has no scratch ring (a failed re-tune?)");
            const uint64_t slots = (uint64_t)plan->ring_slots < plan->batch ? (uint64_t)plan->ring_slots : plan->batch;
            const uint64_t depth = (uint64_t)plan->depth < slots ? (uint64_t)plan->depth : (slots > 1 ? slots - 1 : 1);
            e = fwa::launch_ring_1m(dir, a, out, plan->ring, tb.tw_inner, tb.tw_outer, plan->ring_ctl,
                                    (uint32_t)plan->batch,
                                    (uint32_t)depth, (uint32_t)(slots > depth ? slots : depth + 1),
                                        (uint32_t)plan->wgs, scale, st);
            break;
        }
#endif
        case PATH_TILED: {
            // Per group of transforms: pass A, user buffer -> ring slab; [pass B, in place in the slab]; pass C, slab ->
            // result buffer (src for even log2 n -- in place at group granularity -- else second).  The three passes are
            // built once per exec (TiledPass above: kernel choice + launch arguments, nothing on the heap) and only
            // receive the group's pointers here.
            const TiledShape shape = tiled_shape(plan);
            const TiledPass pa = pass_a(plan, shape, tb, dir), pb = pass_b(plan, shape, tb, dir),
                            pc = pass_c(plan, shape, tb, dir, scale);
            return run_groups(plan, st, [&](uint64_t g, uint64_t cnt, hipStream_t s, size_t c) {
                v2f *slab = plan->ring + (uint64_t)c * G * N;
                hipError_t le = pa(a + g * G * N, slab, cnt, s);
                if (le == hipSuccess && pb.kernel != TiledPass::NONE) le = pb(slab, slab, cnt, s);
                if (le == hipSuccess) le = pc(slab, out + g * G * N, cnt, s);
                return le;
            });
        }
        default:
            return fail(ctx, FWA_ERR_UNSUPPORTED, "plan path not implemented");
    }
    if (e != hipSuccess) return fail_hip(ctx, e, "kernel launch", FWA_ERR_LAUNCH);
    return FWA_OK;
}

int32_t fwa_describe_path(uint32_t fft_len, int32_t *path, uint32_t log2_factors[3])
{
    if (!path || !log2_factors) return fail(nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    if (!is_pow2(fft_len)) return fail(nullptr, FWA_ERR_INVALID_ARG, "fft_len must be a power of two >= 1");
    if (fft_len > (1u << 30)) return fail(nullptr, FWA_ERR_UNSUPPORTED, "fft_len above 2^30 is not supported");
    *path = (int32_t)choose_path(fft_len, ~0ull, log2_factors);
    return FWA_OK;
}

}  // extern "C"
