// api.cpp -- host side of the C ABI declared in include/fft_wgpu_amd.h.
//
// Mirrors the plan objects of reference src/processor.rs (Forward :7-159,
// Inverse :231-341, Normalize :409-505, Onlyinverse :566-670) without any of
// its wgpu plumbing: a plan owns (or shares through the context's plan cache)
// its twiddle tables and scratch, exec only enqueues kernels on the caller's
// stream and allocates nothing.
#include "../../include/fft_wgpu_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "internal.h"
#include "kernels.h"

using fwa::v2f;

// FWA_LAB (libfft_wgpu_amd_lab.so, `make lab`): the same ABI plus the kernel families that measured slower than the
// shipped ones -- paths 5 and 8, tile_w = 32, small_reg != 1.  The product library rejects those settings.
#ifdef FWA_LAB
constexpr bool kLab = true;
#else
constexpr bool kLab = false;
#endif

enum fwa_path : int64_t {
    PATH_SMALL = 0,       // n <= 32768: one launch (k_tiny / k_small16 / k_lds_small)
    PATH_TWOPASS_1M = 1,  // n = 2^20: k_p1_1m + k_p2_1m per group of transforms
    PATH_R2_GLOBAL = 2,   // the reference recurrence literally, one launch per stage (forced only)
    PATH_NORMALIZE = 3,
    PATH_IDENTITY = 4,    // n = 1
    PATH_RING_1M = 5,     // n = 2^20: the same two passes as ONE persistent launch with a small ring (k_ring_1m)
    PATH_TEAM = 8,        // n = 2^16 .. 2^18: both passes in one persistent launch, intermediate in one XCD's L2 (k_team)
    PATH_TILED = 7,       // n = N1*N2[*N3], each 64..1024: 2-3 k_tile passes
};

namespace {

// Device tables of one transform length, shared by every plan of that length on a context (plan cache):
// tables hold forward twiddles only (the inverse conjugates on use), so all plan kinds share them.
struct Tables {
    v2f *tw_half = nullptr;   // n/2 entries, processor.rs:43-49 (small / literal paths)
    v2f *tw_inner = nullptr;  // 2^20 path: [k1][n'] = W_1024^{n' k1}
    v2f *tw_outer[2] = {nullptr, nullptr};  // 2^20 path, tile width 16 / 32: per tile A[32][W], B[32][W]
    v2f *tw_l[3] = {nullptr, nullptr, nullptr};  // tiled path: per-factor W_L tables
    v2f *tw_lo1 = nullptr, *tw_hi1 = nullptr;    // four-step tables of pass A (domain n)
    v2f *tw_lo_b = nullptr, *tw_hi_b = nullptr;  // four-step tables of pass B (domain N2*N3)
    ~Tables()
    {
        for (v2f *t : {tw_half, tw_inner, tw_outer[0], tw_outer[1], tw_l[0], tw_l[1], tw_l[2], tw_lo1, tw_hi1, tw_lo_b,
                       tw_hi_b})
            if (t) (void)hipFree(t);
    }
};

}  // namespace

struct fwa_ctx {
    int device = -1;
    hipDeviceProp_t prop{};
    mutable std::string err;
    bool setup_1m_done = false;
    bool setup_small_done = false;
    // plan cache: (fft_len, path, factor signature) -> tables; ring allocations of destroyed plans by size
    std::map<std::tuple<uint32_t, int64_t, uint32_t>, std::shared_ptr<Tables>> tables;
    std::vector<std::pair<uint64_t, void *>> free_rings;  // rings of destroyed plans, oldest first
    uint64_t free_ring_bytes = 0;
    int64_t n_table_builds = 0, n_table_hits = 0, n_ring_allocs = 0, n_ring_reuses = 0, last_plan_create_us = 0;
    // Internal chain streams of the pipelined paths: created once per context, shared by every plan, and checked at
    // creation to run kernels side by side (chain_streams() below).
    std::vector<hipStream_t> chains;
    std::vector<hipStream_t> user_streams;  // alive streams made by fwa_stream_create, oldest first
    std::vector<fwa_stream *> live_streams; // every alive fwa_stream handle of this context (created or wrapped)
    int64_t n_chain_checks = 0, n_chain_rejects = 0, chain_pair_us = 0, chain_single_us = 0;
    int64_t chain_check = 1;                // fwa_ctx_set_i64("chain_check", 0): new streams are taken as the runtime hands them out
    std::vector<int> peers_enabled;         // device ordinals this context's device has peer access to (enabled once)
};
struct fwa_stream {
    fwa_ctx *ctx = nullptr;   // nullptr once the context has been destroyed (the handle can still be destroyed)
    hipStream_t s = nullptr;
    bool owned = false;
    int device = -1;
};
struct fwa_buf {
    fwa_ctx *ctx = nullptr;
    void *p = nullptr;
    uint64_t bytes = 0;
    bool owned = false;
    int device = -1;          // for fwa_buf_free after the context is gone
};
struct fwa_event {
    fwa_ctx *ctx = nullptr;
    hipEvent_t e = nullptr;
};

struct fwa_plan {
    fwa_ctx *ctx = nullptr;
    int32_t kind = 0;
    uint32_t n = 0;
    uint32_t lg = 0;
    uint64_t batch = 0;
    fwa_buf *src = nullptr;        // buffer_a (processor.rs:12,237,575) / buffer1 for Normalize
    fwa_buf *second = nullptr;     // buffer_b: plan-owned (Forward/Inverse) or caller's src2
    fwa_buf own_second;            // storage when plan-owned
    bool second_owned = false;
    int64_t path = PATH_R2_GLOBAL;
    bool frozen = false;           // first exec done -> tunables locked
    std::shared_ptr<Tables> tb;    // shared through ctx->tables
    v2f *tw_half_private = nullptr;  // forced literal path on a size whose cached tables have no n/2 table
    uint32_t lf[3] = {0, 0, 0};    // tiled path: log2 of the factors (lf[2] = 0 for two factors)
    // pipeline state (two-pass 2^20 and tiled paths): groups of transforms alternate over internal streams
    v2f *ring = nullptr;
    uint64_t ring_bytes = 0;
    int64_t group = 16;            // transforms per launch
    int64_t n_streams = 2;         // internal streams (chains) the groups alternate over
    int64_t tile_w = 16;           // 2^20 path: columns per tile (16: 512-thread workgroups, 32: 1024-thread)
    int64_t xcd_swizzle = -1;      // XCD-aware block -> tile mapping: -1 = per-path default (on for the 2^20 two-pass path:
                                   // +2 %; off for the tiled path: 1-5 % faster without, profiles/round2/sweep_xcd_swizzle.jsonl)
    int64_t rows32 = 1;            // two-pass tiled plans with a 512..2048-point second factor: 1 = k_rows32 as last pass
    int64_t p1_gen = 1;            // tiled plans with first factor 1024: 1 = k_p1_gen as pass A, 0 = k_tile
    int64_t colsw = 0;             // tiled plans with first factor 256 / 512: 1 = k_colsw (64 / 32-column tiles) as pass A, 0 = k_tile
    int64_t tile_ring = 1;         // k_colsw + k_rows32: 1 = tile-contiguous ring slab, 0 = matrix layout
    int64_t ring_rotate = 1;       // laboratory: the ring is this many times larger and the groups rotate through it (same
                                   // launches, larger cache footprint: prices what the Infinity Cache gives the ring)
    int64_t small_reg = 1;         // n <= 32768: 1 = k_chunk / k_small32, 3 = direct 16-point kernels, 2 = + wave shuffles, 0 = LDS radix-2
    std::vector<hipStream_t> istreams;
    std::vector<hipEvent_t> idone;
    hipEvent_t ev_fork = nullptr;
    hipStream_t last_stream = nullptr;  // the caller's stream of the last exec that used the ring: fwa_plan_destroy waits for the
    bool ran_on_stream = false;         // work enqueued there (an event per exec would cost 4-5 us on the 1-3-launch latency shapes)
    // persistent 2^20 pipeline (PATH_RING_1M)
    uint32_t *ring_ctl = nullptr;  // ticket, error word, per-transform hand-off counters
    int64_t depth = 8;             // pass-2 tiles of transform t run beside pass-1 tiles of transform t + depth
    int64_t ring_slots = 12;       // transforms of intermediate kept (>= depth + 1)
    int64_t wgs = 512;             // persistent workgroups (2 per CU)
    // L2-resident team pipeline (PATH_TEAM)
    int64_t max_teams = 0;         // teams (= slabs) per XCD; 0 = as many as fit 3 MiB of an XCD's 4-MiB L2
    int64_t inject_fail_group = -1;  // laboratory: the launch of this group fails once (error path of run_groups under test)
};

namespace {

thread_local std::string g_err;  // ctx-less failures

int32_t fail(const fwa_ctx *ctx, int32_t st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    else g_err = msg;
    return st;
}
int32_t fail_hip(const fwa_ctx *ctx, hipError_t e, const char *what, int32_t st = FWA_ERR_HIP)
{
    // HIP keeps the last error until somebody reads it; the launch wrappers read it after every launch, so an error that
    // has been reported here (e.g. an out-of-memory hipMalloc) must not surface again as a bogus launch failure later
    (void)hipGetLastError();
    std::string m = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    if (e == hipErrorOutOfMemory) st = FWA_ERR_OUT_OF_MEMORY;
    return fail(ctx, st, m);
}

#define HIP_TRY(ctx, call)                                       \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #call); \
    } while (0)

// Every entry point that touches the device makes the context's device current first: with one context per
// device in one process work must not land on whichever device was used last.
#define USE_DEVICE(ctx)                                                             \
    do {                                                                            \
        int cur_ = -1;                                                              \
        if (hipGetDevice(&cur_) != hipSuccess || cur_ != (ctx)->device) HIP_TRY((ctx), hipSetDevice((ctx)->device)); \
    } while (0)

bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }
uint32_t ilog2(uint32_t n)
{
    uint32_t l = 0;
    while ((1u << l) < n) ++l;
    return l;
}

// Path and per-pass FFT lengths (log2) for a transform length; shared by fwa_plan_create and fwa_describe_path.
// `batch` separates two regimes (profiles/round2/sweep_small_batch_latency.jsonl):
//  * throughput (n * batch > 2^20 samples): few passes of fat tiles -- a 1024-point first pass (k_p1_gen / the 2^20
//    pipeline, 64 KiB tiles of 512 threads) and 32-point-per-thread rows;
//  * latency (at most 2^20 samples in flight, or a single 2^21 / 2^22 transform, or fewer than FEW_1M transforms of 2^20):
//    fat tiles leave most of the 256 CUs idle (one 2^16 transform = FOUR 1024 x 16 tiles), so the plan uses the
//    smallest tiles instead -- balanced two passes up to 2^17, balanced three passes of 64/128-point tiles above
//    (2^16 x 1: 11.9 us against 16.2; 2^18 x 1: 12.7 against 18.4; 2^20 x 1: 19 against 24).
constexpr uint64_t FEW_1M = 4;
int64_t choose_path(uint32_t n, uint64_t batch, uint32_t lf[3], bool *colsw = nullptr)
{
    lf[0] = lf[1] = lf[2] = 0;
    if (colsw) *colsw = false;
    const uint32_t lg = ilog2(n);
    if (n == 1) return PATH_IDENTITY;
    if (n <= 32768) { lf[0] = lg; return PATH_SMALL; }
    const bool few = (lg < 20 && batch <= ((1ull << 20) >> lg)) || (lg == 20 && batch < FEW_1M) || ((lg == 21 || lg == 22) && batch == 1);
    if (n == (1u << 20) && !few) { lf[0] = lf[1] = 10; return PATH_TWOPASS_1M; }
    if (n <= (1u << 30)) {
        // factors of 64..1024 each, 2048 for the rows of a two-pass plan (re-tunable: key "factors").  Throughput
        // regime: two passes up to 2^19 and at 2^21 .. 2^23 (2048 / 4096-point passes), three otherwise; a 1024-point first pass (k_p1_gen) wherever the
        // other factors stay >= 64, measured faster than a balanced split except at 2^22 (level)
        // (profiles/round2/p1gen_sweep.jsonl, factor_sweep.jsonl, sweep_rows32.jsonl).
        // Round 3 (profiles/round3/sweep_colsw_32GiB.jsonl, sweep_factors_24_28_colsw.jsonl): short columns in wide tiles
        // (k_colsw: 256 x 64 / 512 x 32, 512- / 256-byte row segments) beat the 1024 x 16 tile of k_p1_gen as pass A
        // wherever the last pass keeps <= 1024-point rows: 2^16 .. 2^19 +7-9 %, three-pass sizes 2^24 .. 2^28 +2-16 %.
        if (!few && lg == 22) { lf[0] = 10; lf[1] = 12; }       // 1024 x 4096: k_p1_gen + k_rows32 (8 rows of 4096 per workgroup)
        else if (!few && lg == 23) { lf[0] = 11; lf[1] = 12; }  // 2048 x 4096: k_cols32 + k_rows32
        else if (few && lg <= 17) { lf[0] = lg / 2; lf[1] = lg - lf[0]; }
        else if (!few && lg <= 18) { lf[0] = 8; lf[1] = lg - 8; if (colsw) *colsw = true; }   // 256 x (256 .. 1024)
        else if (!few && lg == 19) { lf[0] = 9; lf[1] = 10; if (colsw) *colsw = true; }       // 512 x 1024
        else if (!few && lg == 21) { lf[0] = 10; lf[1] = lg - 10; }
        else if (!few && lg >= 24 && lg <= 28) { lf[0] = 9; lf[1] = (lg - 9) / 2; lf[2] = lg - 9 - lf[1]; if (colsw) *colsw = true; }
        else if (!few && lg >= 29) { lf[0] = 10; lf[1] = (lg - 10) / 2; lf[2] = lg - 10 - lf[1]; }
        else if (few && lg == 20) { lf[0] = lf[1] = 6; lf[2] = 8; }  // 16.4 us against 18.2 for 64 x 128 x 128 (sweep_factor_permutations_batch1.jsonl)
        else for (uint32_t i = 0; i < 3; ++i) lf[i] = lg / 3 + (i >= 3 - lg % 3 ? 1 : 0);
        return PATH_TILED;
    }
    return PATH_R2_GLOBAL;
}

// reference twiddle rule, processor.rs:43-49: f64 math, rounded to f32.
v2f tw_f64(uint64_t k, uint64_t n)
{
    const double PI = 3.14159265358979323846;
    const double theta = -2.0 * PI * (double)k / (double)n;
    return v2f{(float)std::cos(theta), (float)std::sin(theta)};
}

int32_t upload_table(fwa_ctx *ctx, const std::vector<v2f> &h, v2f **d)
{
    *d = nullptr;
    if (h.empty()) return FWA_OK;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(d), h.size() * sizeof(v2f)));
    HIP_TRY(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(v2f), hipMemcpyHostToDevice));
    return FWA_OK;
}

int32_t upload_half_table(fwa_ctx *ctx, uint32_t n, v2f **d)
{
    std::vector<v2f> h(n / 2);
    for (uint32_t k = 0; k < n / 2; ++k) h[k] = tw_f64(k, n);
    return upload_table(ctx, h, d);
}

// W_cur^e = hi[e >> 10] * lo[e & 1023]
int32_t upload_level(fwa_ctx *ctx, uint64_t cur, v2f **lo, v2f **hi)
{
    const uint64_t nlo = cur < 1024 ? cur : 1024, nhi = cur < 1024 ? 1 : cur / 1024;
    std::vector<v2f> l(nlo), h(nhi);
    for (uint64_t j = 0; j < nlo; ++j) l[j] = tw_f64(j, cur);
    for (uint64_t j = 0; j < nhi; ++j) h[j] = tw_f64(1024 * j, cur);
    int32_t s = upload_table(ctx, l, lo);
    return s ? s : upload_table(ctx, h, hi);
}

int32_t build_tables(fwa_ctx *ctx, uint32_t n, int64_t path, const uint32_t lf[3], Tables *t)
{
    int32_t st = FWA_OK;
    if (path == PATH_SMALL) return n >= 2 ? upload_half_table(ctx, n, &t->tw_half) : FWA_OK;
    if (path == PATH_TWOPASS_1M) {
        std::vector<v2f> inner(1024);
        for (uint32_t k1 = 0; k1 < 32; ++k1)
            for (uint32_t q = 0; q < 32; ++q) inner[k1 * 32 + q] = tw_f64((uint64_t)k1 * q, 1024);
        st = upload_table(ctx, inner, &t->tw_inner);
        const uint64_t N = 1ull << 20;
        for (int wi = 0; wi < (kLab ? 2 : 1) && !st; ++wi) {
            const uint32_t W = wi ? 32 : 16, tiles = 1024 / W;
            std::vector<v2f> outer((size_t)tiles * 64 * W);
            for (uint32_t tile = 0; tile < tiles; ++tile)
                for (uint32_t k = 0; k < 32; ++k)
                    for (uint32_t c = 0; c < W; ++c) {
                        const uint64_t n2 = (uint64_t)W * tile + c;
                        outer[(size_t)tile * 64 * W + k * W + c] = tw_f64(n2 * k, N);                // A[k1][c]
                        outer[(size_t)tile * 64 * W + 32 * W + k * W + c] = tw_f64(32 * n2 * k, N);  // B[k2][c]
                    }
            st = upload_table(ctx, outer, &t->tw_outer[wi]);
        }
        return st;
    }
    if (path == PATH_TILED) {
        const uint32_t nf = lf[2] ? 3 : 2;
        for (uint32_t i = 0; i < nf && !st; ++i) st = upload_half_table(ctx, 1u << lf[i], &t->tw_l[i]);
        if (!st) st = upload_level(ctx, n, &t->tw_lo1, &t->tw_hi1);
        if (!st && lf[0] == 10) {  // k_p1_gen's first-stage table [k1][n'] = W_1024^{n' k1}
            std::vector<v2f> inner(1024);
            for (uint32_t k1 = 0; k1 < 32; ++k1)
                for (uint32_t q = 0; q < 32; ++q) inner[k1 * 32 + q] = tw_f64((uint64_t)k1 * q, 1024);
            st = upload_table(ctx, inner, &t->tw_inner);
        }
        if (!st && nf == 3) st = upload_level(ctx, (uint64_t)n >> lf[0], &t->tw_lo_b, &t->tw_hi_b);
        return st;
    }
    return FWA_OK;
}

hipStream_t raw(fwa_stream *s) { return s ? s->s : nullptr; }

struct Pipeline {
    v2f *ring = nullptr;
    uint64_t ring_bytes = 0;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    hipEvent_t fork = nullptr;
};

// newest pooled ring of exactly this size, or nullptr
void *pool_take(fwa_ctx *ctx, uint64_t bytes)
{
    for (size_t i = ctx->free_rings.size(); i-- > 0;)
        if (ctx->free_rings[i].first == bytes) {
            void *p = ctx->free_rings[i].second;
            ctx->free_ring_bytes -= bytes;
            ctx->free_rings.erase(ctx->free_rings.begin() + (std::ptrdiff_t)i);
            ++ctx->n_ring_reuses;
            return p;
        }
    return nullptr;
}

void destroy_pipeline_objects(fwa_ctx *ctx, Pipeline &pl, bool pool_ring)
{
    // pl.streams are borrowed from the context (ctx->chains)
    for (auto e : pl.done) (void)hipEventDestroy(e);
    pl.streams.clear();
    pl.done.clear();
    if (pl.fork) { (void)hipEventDestroy(pl.fork); pl.fork = nullptr; }
    if (pl.ring) {
        // keep up to 1 GiB of ring allocations of destroyed plans for the next plan of the same shape; the oldest
        // entries make room for newer ones
        if (pool_ring && ctx && pl.ring_bytes <= (1ull << 30)) {
            ctx->free_rings.emplace_back(pl.ring_bytes, pl.ring);
            ctx->free_ring_bytes += pl.ring_bytes;
            while (ctx->free_ring_bytes > (1ull << 30)) {
                (void)hipFree(ctx->free_rings.front().second);
                ctx->free_ring_bytes -= ctx->free_rings.front().first;
                ctx->free_rings.erase(ctx->free_rings.begin());
            }
        } else {
            (void)hipFree(pl.ring);
        }
        pl.ring = nullptr;
        pl.ring_bytes = 0;
    }
}

// Streams that overlap.  Two HIP streams do not always run side by side on this stack: which hardware queue a new
// stream lands on depends on what the process created and destroyed before, and a pair that shares one runs strictly one
// after the other -- a pipelined plan whose two chains shared a queue took the single-chain time on every exec (+15 %:
// profiles/round3/probe_plan_instance_modes.txt), a host pipeline whose transfer streams shared one moved 21 GB/s each way
// instead of 44.  So a stream created by this library is accepted only if a memory-free spin kernel on it overlaps the same
// kernel on its `peers` (time on all of them at once < single + half a spin); a rejected candidate stays alive until the
// search ends so that the runtime cannot hand the same queue back.  Best effort: after 6 rejections the last candidate is
// kept (a process with more streams than the runtime has hardware queues cannot overlap them all).
//
// Side effects, and how a caller controls them (include/fft_wgpu_amd.h, "Threading"): the check launches ~40-us spin
// kernels on the candidate, on the peers and on a private base stream of its own -- never on the null stream -- and runs
// only when there are peers to overlap with (a plan with one chain, or a context's first stream, costs nothing).  It is
// refused with FWA_ERR_UNSUPPORTED while a stream of this context is capturing a graph (its timing would be meaningless and
// the peers may be the capturing streams), and fwa_ctx_set_i64(ctx, "chain_check", 0) turns it off: streams are then taken
// as the runtime hands them out.
bool any_stream_capturing(const fwa_ctx *ctx)
{
    for (const fwa_stream *s : ctx->live_streams) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (s->s && hipStreamIsCapturing(s->s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return true;
    }
    (void)hipGetLastError();
    return false;
}

int32_t overlapping_stream(fwa_ctx *ctx, const std::vector<hipStream_t> &peers, hipStream_t *out)
{
    *out = nullptr;
    if (peers.empty() || !ctx->chain_check) {
        hipError_t ce = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        if (ce != hipSuccess) { *out = nullptr; return fail_hip(ctx, ce, "hipStreamCreateWithFlags"); }
        return FWA_OK;
    }
    if (any_stream_capturing(ctx))
        return fail(ctx, FWA_ERR_UNSUPPORTED,
                    "a stream of this context is capturing a graph: create plans and streams before the capture begins, or "
                    "turn the stream-overlap check off with fwa_ctx_set_i64(ctx, \"chain_check\", 0)");
    constexpr uint32_t TICKS = 4000, BLOCKS = 256;  // 40 us, one wave per CU
    hipEvent_t e0 = nullptr, e1 = nullptr, fork = nullptr;
    hipStream_t base = nullptr;  // the check's own fork / join stream: nothing here touches the null stream
    std::vector<hipEvent_t> done;
    std::vector<hipStream_t> rejected;
    hipError_t e = hipStreamCreateWithFlags(&base, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    auto timed = [&](const std::vector<hipStream_t> &set, float *us) {  // spin on every stream of `set`, forked from / joined to `base`
        float best = 1e30f;
        for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {
            while (done.size() < set.size() && e == hipSuccess) {
                hipEvent_t d;
                e = hipEventCreateWithFlags(&d, hipEventDisableTiming);
                if (e == hipSuccess) done.push_back(d);
            }
            if (e == hipSuccess) e = hipEventRecord(e0, base);
            if (e == hipSuccess) e = hipEventRecord(fork, base);
            for (size_t i = 0; i < set.size() && e == hipSuccess; ++i) {
                e = hipStreamWaitEvent(set[i], fork, 0);
                if (e == hipSuccess) e = fwa::launch_spin(TICKS, BLOCKS, set[i]);
                if (e == hipSuccess) e = hipEventRecord(done[i], set[i]);
                if (e == hipSuccess) e = hipStreamWaitEvent(base, done[i], 0);
            }
            if (e == hipSuccess) e = hipEventRecord(e1, base);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (ms * 1e3f < best) best = ms * 1e3f;
        }
        *us = best;
    };
    while (!*out && e == hipSuccess) {
        hipStream_t cand = nullptr;
        e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) break;
        std::vector<hipStream_t> set = peers;
        set.push_back(cand);
        float single = 0, all = 0;
        timed({cand}, &single);
        timed(set, &all);
        ++ctx->n_chain_checks;
        const bool overlaps = all < single + 0.5f * (TICKS * 0.01f);
        if (e == hipSuccess && (overlaps || rejected.size() >= 6)) {
            *out = cand;
            ctx->chain_single_us = (int64_t)single;
            ctx->chain_pair_us = (int64_t)all;
        } else {
            rejected.push_back(cand);  // destroyed below, also when a HIP call of the check failed
            if (e == hipSuccess) ++ctx->n_chain_rejects;
        }
    }
    for (auto s : rejected) (void)hipStreamDestroy(s);
    for (auto d : done) (void)hipEventDestroy(d);
    for (auto ev : {e0, e1, fork}) if (ev) (void)hipEventDestroy(ev);
    if (base) (void)hipStreamDestroy(base);
    if (e != hipSuccess) return fail_hip(ctx, e, "stream setup");
    return FWA_OK;
}

// The chain streams of the pipelined paths: created once per context, shared by every plan, each checked against the
// chains accepted before it.
int32_t chain_streams(fwa_ctx *ctx, size_t n)
{
    while (ctx->chains.size() < n) {
        hipStream_t s = nullptr;
        int32_t st = overlapping_stream(ctx, ctx->chains, &s);
        if (st) return st;
        ctx->chains.push_back(s);
    }
    return FWA_OK;
}

Pipeline take_pipeline(fwa_plan *p)
{
    Pipeline pl;
    pl.ring = p->ring; pl.ring_bytes = p->ring_bytes; pl.streams.swap(p->istreams); pl.done.swap(p->idone);
    pl.fork = p->ev_fork;
    p->ring = nullptr; p->ring_bytes = 0; p->ev_fork = nullptr;
    return pl;
}

// Allocate the scratch ring and the internal streams of the pipelined paths.  The new objects are built first
// and swapped in only on success, so a failed re-tune (e.g. a group too large for the free memory) leaves the
// plan exactly as it was.
int32_t build_pipeline(fwa_plan *p, int64_t group, int64_t n_streams)
{
    fwa_ctx *ctx = p->ctx;
#ifdef FWA_LAB
    if (p->path == PATH_RING_1M) {
        // one launch, no internal streams: ring of min(ring_slots, batch) transforms + the control words
        Pipeline pl;
        const uint64_t slots = (uint64_t)p->ring_slots < p->batch ? (uint64_t)p->ring_slots : p->batch;
        pl.ring_bytes = slots * (sizeof(v2f) << 20);
        uint32_t *ctl = nullptr;
        if (pl.ring_bytes) {
            if (void *pooled = pool_take(ctx, pl.ring_bytes)) {
                pl.ring = static_cast<v2f *>(pooled);
            } else {
                hipError_t e = hipMalloc(reinterpret_cast<void **>(&pl.ring), pl.ring_bytes);
                if (e != hipSuccess) return fail_hip(ctx, e, "hipMalloc(ring)");
                ++ctx->n_ring_allocs;
            }
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctl), fwa::ring_ctl_bytes(p->batch));
            if (e != hipSuccess) { destroy_pipeline_objects(ctx, pl, false); return fail_hip(ctx, e, "hipMalloc(ring control)"); }
        }
        Pipeline old = take_pipeline(p);
        destroy_pipeline_objects(ctx, old, true);
        if (p->ring_ctl) (void)hipFree(p->ring_ctl);
        p->ring_ctl = ctl;
        p->ring = pl.ring; p->ring_bytes = pl.ring_bytes;
        return FWA_OK;
    }
    if (p->path == PATH_TEAM) {
        uint32_t ts = 0, th = 0;
        size_t lds = 0;
        fwa::team_geometry(p->lg, &ts, &th, &lds);
        const uint64_t slab = (uint64_t)p->n * sizeof(v2f);
        if (p->max_teams <= 0) p->max_teams = (int64_t)((3ull << 20) / slab ? (3ull << 20) / slab : 1);
        const uint64_t need_teams = (p->batch + 7) / 8;  // more teams than transforms per XCD are useless
        if ((uint64_t)p->max_teams > need_teams && need_teams) p->max_teams = (int64_t)need_teams;
        p->wgs = 8 * p->max_teams * (int64_t)ts;
        Pipeline pl;
        pl.ring_bytes = p->batch ? 8ull * (uint64_t)p->max_teams * slab : 0;
        uint32_t *ctl = nullptr;
        if (pl.ring_bytes) {
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&pl.ring), pl.ring_bytes);
            if (e != hipSuccess) return fail_hip(ctx, e, "hipMalloc(team slabs)");
            ++ctx->n_ring_allocs;
            e = hipMalloc(reinterpret_cast<void **>(&ctl), fwa::team_ctl_bytes(p->lg, (uint32_t)p->max_teams));
            if (e != hipSuccess) { destroy_pipeline_objects(ctx, pl, false); return fail_hip(ctx, e, "hipMalloc(team control)"); }
        }
        Pipeline old = take_pipeline(p);
        destroy_pipeline_objects(ctx, old, true);
        if (p->ring_ctl) (void)hipFree(p->ring_ctl);
        p->ring_ctl = ctl;
        p->ring = pl.ring; p->ring_bytes = pl.ring_bytes;
        return FWA_OK;
    }
#endif
    if (p->path != PATH_TWOPASS_1M && p->path != PATH_TILED) return FWA_OK;
    if (group < 1) group = 1;
    if ((uint64_t)group > p->batch && p->batch) group = (int64_t)p->batch;
    const uint64_t n_groups = p->batch ? (p->batch + group - 1) / group : 0;
    if (n_streams < 1) n_streams = 1;
    if ((uint64_t)n_streams > n_groups && n_groups) n_streams = (int64_t)n_groups;
    Pipeline pl;
    pl.ring_bytes = p->batch ? (uint64_t)group * (uint64_t)n_streams * (uint64_t)p->n * sizeof(v2f) * (uint64_t)p->ring_rotate : 0;
    auto bail = [&](int32_t st) { destroy_pipeline_objects(ctx, pl, false); return st; };
    if (pl.ring_bytes) {
        if (void *pooled = pool_take(ctx, pl.ring_bytes)) {
            pl.ring = static_cast<v2f *>(pooled);
        } else {
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&pl.ring), pl.ring_bytes);
            if (e != hipSuccess) { pl.ring = nullptr; return bail(fail_hip(ctx, e, "hipMalloc(ring)")); }
            ++ctx->n_ring_allocs;
        }
        if (n_streams > 1) {
            int32_t cs = chain_streams(ctx, (size_t)n_streams);
            if (cs) return bail(cs);
            hipError_t e = hipEventCreateWithFlags(&pl.fork, hipEventDisableTiming);
            if (e != hipSuccess) { pl.fork = nullptr; return bail(fail_hip(ctx, e, "hipEventCreate")); }
            for (int64_t i = 0; i < n_streams; ++i) {
                hipEvent_t ev;
                pl.streams.push_back(ctx->chains[(size_t)i]);
                e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e != hipSuccess) return bail(fail_hip(ctx, e, "hipEventCreate"));
                pl.done.push_back(ev);
            }
        }
    }
    Pipeline old = take_pipeline(p);
    destroy_pipeline_objects(ctx, old, true);
    p->ring = pl.ring; p->ring_bytes = pl.ring_bytes; p->istreams.swap(pl.streams); p->idone.swap(pl.done);
    p->ev_fork = pl.fork;
    p->group = group;
    p->n_streams = n_streams;
    return FWA_OK;
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
    if ((p->path == PATH_TWOPASS_1M || p->path == PATH_RING_1M || (p->path == PATH_TILED && p->lf[0] == 10)) && !ctx->setup_1m_done) {
        hipError_t e = fwa::setup_1m_kernels();
#ifdef FWA_LAB
        if (e == hipSuccess) e = fwa::setup_lab_1m_kernels();
#endif
        if (e != hipSuccess) return fail_hip(ctx, e, "hipFuncSetAttribute(max dynamic LDS)");
        ctx->setup_1m_done = true;
    }
#ifdef FWA_LAB
    if (p->path == PATH_TEAM) {
        hipError_t pe = fwa::prepare_team(p->lg);
        if (pe != hipSuccess) return fail_hip(ctx, pe, "hipFuncSetAttribute(max dynamic LDS)");
    }
#endif
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
    const auto key = std::make_tuple(fft_len, p->path == PATH_RING_1M ? (int64_t)PATH_TWOPASS_1M
                                              : (p->path == PATH_TEAM ? (int64_t)PATH_TILED : p->path), sig);
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
    } else if (p->path == PATH_RING_1M || p->path == PATH_TEAM) {
        st = build_pipeline(p, 0, 0);
    }
    if (st) return st;
    p->tb = tb;
    return FWA_OK;
}

static size_t ctl_bytes(const fwa_plan *p)
{
#ifdef FWA_LAB
    if (p->path == PATH_TEAM) return fwa::team_ctl_bytes(p->lg, (uint32_t)p->max_teams);
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

// Block -> tile map of the tiled plans' kernels (xcd_map, device_common.h) when the caller has not set "xcd_swizzle": measured per
// size at the 32-GiB footprint, three interleaved runs (profiles/round4/sweep_tiled_block_maps.jsonl): the k_colsw plans gain 2-4 %
// from XCD-contiguous runs (2^17 .. 2^19: bit 0; 2^16 and 1024 x 2048: with the CU pairs, bits 0 + 2); 2^22 and up lose 1-10 %.
static uint32_t tiled_swizzle_default(const fwa_plan *p)
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
    if (plan->batch && !plan->ring) return fail(ctx, FWA_ERR_INVALID_ARG, "plan has no scratch ring (a failed re-tune?)");
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
    // Join the chains back to the caller's stream ALSO when a launch failed: the groups enqueued before the failure keep
    // running on the chains, and whatever the caller enqueues next on `st` (a copy of the partial result, the free of the
    // buffer) must be ordered behind them.
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

}  // namespace

extern "C" {

int32_t fwa_abi_version(void) { return FWA_ABI_VERSION; }

const char *fwa_last_error_string(const fwa_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

const char *fwa_status_string(int32_t status)
{
    switch (status) {
        case FWA_OK: return "ok";
        case FWA_ERR_INVALID_ARG: return "invalid argument";
        case FWA_ERR_OUT_OF_MEMORY: return "out of device memory";
        case FWA_ERR_HIP: return "HIP runtime error";
        case FWA_ERR_LAUNCH: return "kernel launch failed";
        case FWA_ERR_NO_DEVICE: return "no usable device";
        case FWA_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

int32_t fwa_device_count(int32_t *count)
{
    if (!count) return fail(nullptr, FWA_ERR_INVALID_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(nullptr, e, "hipGetDeviceCount", FWA_ERR_NO_DEVICE);
    }
    *count = n;
    return FWA_OK;
}

int32_t fwa_device_info(int32_t device_ordinal, char *name, size_t name_cap, int32_t *compute_units, uint64_t *hbm_bytes,
                        int32_t *usable)
{
    if (name && name_cap) name[0] = 0;
    if (compute_units) *compute_units = 0;
    if (hbm_bytes) *hbm_bytes = 0;
    if (usable) *usable = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, FWA_ERR_NO_DEVICE, "no HIP device visible");
    }
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, FWA_ERR_INVALID_ARG, "device ordinal out of range");
    hipDeviceProp_t prop{};
    e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e != hipSuccess) return fail_hip(nullptr, e, "hipGetDeviceProperties");
    if (name && name_cap) {
        std::strncpy(name, prop.gcnArchName, name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (usable) *usable = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    return FWA_OK;
}

// The slab rule of SURVEY.md 8(e): contiguous runs of whole transforms, sizes differing by at most one.  Pure host logic;
// fft_wgpu_amd/sharding.py::slab and fft_wgpu::slab (include/fft_wgpu.hpp) are this function.
int32_t fwa_slab(uint64_t batch, int32_t rank, int32_t world, uint64_t *first, uint64_t *count)
{
    if (!first || !count) return fail(nullptr, FWA_ERR_INVALID_ARG, "first/count is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(nullptr, FWA_ERR_INVALID_ARG, "bad rank / world size");
    const uint64_t base = batch / (uint64_t)world, extra = batch % (uint64_t)world, r = (uint64_t)rank;
    *first = r * base + (r < extra ? r : extra);
    *count = base + (r < extra ? 1 : 0);
    return FWA_OK;
}

int32_t fwa_ctx_create(int32_t device_ordinal, fwa_ctx **out)
{
    if (!out) return fail(nullptr, FWA_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, FWA_ERR_NO_DEVICE,
                    std::string("no HIP device visible: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0"));
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, FWA_ERR_INVALID_ARG, "device ordinal out of range");
    fwa_ctx *ctx = new (std::nothrow) fwa_ctx;
    if (!ctx) return fail(nullptr, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device_ordinal;
    e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_ordinal);
    if (e != hipSuccess) {
        int32_t st = fail_hip(nullptr, e, "hipSetDevice/hipGetDeviceProperties", FWA_ERR_NO_DEVICE);
        delete ctx;
        return st;
    }
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        int32_t st = fail(nullptr, FWA_ERR_NO_DEVICE,
                          std::string("device is ") + ctx->prop.gcnArchName + ", this library is built for gfx950 only");
        delete ctx;
        return st;
    }
    *out = ctx;
    return FWA_OK;
}

int32_t fwa_ctx_destroy(fwa_ctx *ctx)
{
    if (!ctx) return FWA_OK;
    (void)hipSetDevice(ctx->device);
    for (auto &kv : ctx->free_rings) (void)hipFree(kv.second);
    for (auto s : ctx->chains) (void)hipStreamDestroy(s);
    // stream handles that outlive their context (a garbage-collected host language frees in any order) stay destroyable
    for (fwa_stream *st : ctx->live_streams) st->ctx = nullptr;
    ctx->tables.clear();
    delete ctx;
    return FWA_OK;
}

int32_t fwa_ctx_synchronize(fwa_ctx *ctx)
{
    if (!ctx) return fail(nullptr, FWA_ERR_INVALID_ARG, "ctx is NULL");
    USE_DEVICE(ctx);
    HIP_TRY(ctx, hipDeviceSynchronize());
    return FWA_OK;
}

int32_t fwa_ctx_get_i64(const fwa_ctx *ctx, const char *key, int64_t *value)
{
    if (!ctx || !key || !value) return fail(ctx, FWA_ERR_INVALID_ARG, "NULL argument");
    const std::string k(key);
    if (k == "device") *value = ctx->device;
    else if (k == "table_builds") *value = ctx->n_table_builds;
    else if (k == "table_cache_hits") *value = ctx->n_table_hits;
    else if (k == "ring_allocs") *value = ctx->n_ring_allocs;
    else if (k == "ring_reuses") *value = ctx->n_ring_reuses;
    else if (k == "last_plan_create_us") *value = ctx->last_plan_create_us;
    else if (k == "pooled_ring_bytes") *value = (int64_t)ctx->free_ring_bytes;
    else if (k == "chain_streams") *value = (int64_t)ctx->chains.size();
    else if (k == "chain_checks") *value = ctx->n_chain_checks;
    else if (k == "chain_rejects") *value = ctx->n_chain_rejects;
    else if (k == "chain_pair_us") *value = ctx->chain_pair_us;
    else if (k == "chain_single_us") *value = ctx->chain_single_us;
    else if (k == "chain_check") *value = ctx->chain_check;
    else if (k == "live_streams") *value = (int64_t)ctx->live_streams.size();
    else if (k == "mem_free_bytes" || k == "mem_total_bytes") {
        int cur = -1;
        size_t fr = 0, tot = 0;
        if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) (void)hipSetDevice(ctx->device);
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) return fail(ctx, FWA_ERR_HIP, "hipMemGetInfo");
        *value = (int64_t)(k == "mem_free_bytes" ? fr : tot);
    }
    else return fail(ctx, FWA_ERR_INVALID_ARG, "unknown key: " + k);
    return FWA_OK;
}

int32_t fwa_ctx_set_i64(fwa_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return fail(ctx, FWA_ERR_INVALID_ARG, "NULL argument");
    const std::string k(key);
    if (k == "chain_check") {
        // 0: streams this library creates (the chain streams of pipelined plans, fwa_stream_create) are no longer tested
        // for overlap with spin kernels (overlapping_stream above)
        if (value != 0 && value != 1) return fail(ctx, FWA_ERR_INVALID_ARG, "chain_check is 0 or 1");
        ctx->chain_check = value;
        return FWA_OK;
    }
    return fail(ctx, FWA_ERR_INVALID_ARG, "unknown or read-only key: " + k);
}

int32_t fwa_ctx_device_info(const fwa_ctx *ctx, char *name, size_t name_cap, int32_t *compute_units,
                            uint64_t *hbm_bytes)
{
    if (!ctx) return fail(nullptr, FWA_ERR_INVALID_ARG, "ctx is NULL");
    if (name && name_cap) {
        std::strncpy(name, ctx->prop.gcnArchName, name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return FWA_OK;
}

// ---- streams -------------------------------------------------------------
int32_t fwa_stream_create(fwa_ctx *ctx, fwa_stream **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    USE_DEVICE(ctx);
    // checked to overlap the (up to two) streams this context created most recently: a caller that makes a transfer
    // stream and a compute stream back to back gets two that really run side by side (overlapping_stream above)
    hipStream_t s;
    std::vector<hipStream_t> peers(ctx->user_streams.end() - (std::ptrdiff_t)std::min<size_t>(2, ctx->user_streams.size()),
                                   ctx->user_streams.end());
    int32_t rc = overlapping_stream(ctx, peers, &s);
    if (rc) return rc;
    fwa_stream *st = new (std::nothrow) fwa_stream;
    if (!st) { (void)hipStreamDestroy(s); return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    st->ctx = ctx; st->s = s; st->owned = true; st->device = ctx->device;
    ctx->user_streams.push_back(s);
    ctx->live_streams.push_back(st);
    *out = st;
    return FWA_OK;
}

int32_t fwa_stream_wrap(fwa_ctx *ctx, void *hip_stream, fwa_stream **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    fwa_stream *st = new (std::nothrow) fwa_stream;
    if (!st) return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    st->ctx = ctx; st->s = reinterpret_cast<hipStream_t>(hip_stream); st->owned = false; st->device = ctx->device;
    ctx->live_streams.push_back(st);
    *out = st;
    return FWA_OK;
}

int32_t fwa_stream_synchronize(fwa_stream *stream)
{
    if (!stream) return fail(nullptr, FWA_ERR_INVALID_ARG, "stream is NULL");
    if (!stream->ctx) return fail(nullptr, FWA_ERR_INVALID_ARG, "the stream's context has been destroyed");
    USE_DEVICE(stream->ctx);
    HIP_TRY(stream->ctx, hipStreamSynchronize(stream->s));
    return FWA_OK;
}

int32_t fwa_stream_destroy(fwa_stream *stream)
{
    if (!stream) return FWA_OK;
    if (stream->ctx) {
        auto &ls = stream->ctx->live_streams;
        ls.erase(std::remove(ls.begin(), ls.end(), stream), ls.end());
        auto &us = stream->ctx->user_streams;
        us.erase(std::remove(us.begin(), us.end(), stream->s), us.end());
    }
    if (stream->owned) {
        (void)hipSetDevice(stream->device);
        (void)hipStreamDestroy(stream->s);
    }
    delete stream;
    return FWA_OK;
}

// ---- buffers -------------------------------------------------------------
int32_t fwa_buf_alloc(fwa_ctx *ctx, uint64_t bytes, fwa_buf **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    USE_DEVICE(ctx);
    void *p = nullptr;
    if (bytes) {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail_hip(ctx, e, "hipMalloc");
    }
    fwa_buf *b = new (std::nothrow) fwa_buf;
    if (!b) { (void)hipFree(p); return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    b->ctx = ctx; b->p = p; b->bytes = bytes; b->owned = true; b->device = ctx->device;
    *out = b;
    return FWA_OK;
}

int32_t fwa_buf_wrap(fwa_ctx *ctx, void *device_ptr, uint64_t bytes, fwa_buf **out)
{
    if (!ctx || !out || (!device_ptr && bytes)) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out/device_ptr is NULL");
    if (reinterpret_cast<uintptr_t>(device_ptr) & 15)
        return fail(ctx, FWA_ERR_INVALID_ARG, "device pointer must be 16-byte aligned");
    fwa_buf *b = new (std::nothrow) fwa_buf;
    if (!b) return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->ctx = ctx; b->p = device_ptr; b->bytes = bytes; b->owned = false; b->device = ctx->device;
    *out = b;
    return FWA_OK;
}

int32_t fwa_buf_free(fwa_buf *buf)
{
    if (!buf) return FWA_OK;
    if (buf->owned && buf->p) { (void)hipSetDevice(buf->device); (void)hipFree(buf->p); }
    delete buf;
    return FWA_OK;
}

int32_t fwa_buf_upload(fwa_buf *dst, uint64_t dst_offset, const void *host, uint64_t bytes, fwa_stream *stream)
{
    if (!dst || (!host && bytes)) return fail(dst ? dst->ctx : nullptr, FWA_ERR_INVALID_ARG, "dst/host is NULL");
    if (dst_offset > dst->bytes || bytes > dst->bytes - dst_offset)
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "upload range exceeds buffer");
    if (!bytes) return FWA_OK;
    USE_DEVICE(dst->ctx);
    HIP_TRY(dst->ctx, hipMemcpyAsync(static_cast<char *>(dst->p) + dst_offset, host, bytes, hipMemcpyHostToDevice,
                                     raw(stream)));
    return FWA_OK;
}

int32_t fwa_buf_download(void *host, const fwa_buf *src, uint64_t src_offset, uint64_t bytes, fwa_stream *stream)
{
    if (!src || (!host && bytes)) return fail(src ? src->ctx : nullptr, FWA_ERR_INVALID_ARG, "src/host is NULL");
    if (src_offset > src->bytes || bytes > src->bytes - src_offset)
        return fail(src->ctx, FWA_ERR_INVALID_ARG, "download range exceeds buffer");
    if (!bytes) return FWA_OK;
    USE_DEVICE(src->ctx);
    HIP_TRY(src->ctx, hipMemcpyAsync(host, static_cast<const char *>(src->p) + src_offset, bytes,
                                     hipMemcpyDeviceToHost, raw(stream)));
    // map_async + poll(wait) in the reference (examples/basic.rs:105-106): the data is on the host on return
    HIP_TRY(src->ctx, hipStreamSynchronize(raw(stream)));
    return FWA_OK;
}

// Peer reachability of two contexts' devices.  kind: 0 = none (stage through the host or use fwa_comm_*), 1 = the same
// device, 2 = peer access (xGMI or PCIe peer-to-peer), enabled on first use in both directions.
static int32_t peer_kind(fwa_ctx *a, fwa_ctx *b, int32_t *kind)
{
    *kind = 0;
    if (a->device == b->device) { *kind = 1; return FWA_OK; }
    int ab = 0, ba = 0;
    HIP_TRY(a, hipDeviceCanAccessPeer(&ab, a->device, b->device));
    HIP_TRY(a, hipDeviceCanAccessPeer(&ba, b->device, a->device));
    if (!ab || !ba) return FWA_OK;
    for (fwa_ctx *c : {a, b}) {
        fwa_ctx *o = c == a ? b : a;
        if (std::find(c->peers_enabled.begin(), c->peers_enabled.end(), o->device) != c->peers_enabled.end()) continue;
        HIP_TRY(c, hipSetDevice(c->device));
        hipError_t e = hipDeviceEnablePeerAccess(o->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail_hip(c, e, "hipDeviceEnablePeerAccess");
        (void)hipGetLastError();
        c->peers_enabled.push_back(o->device);
    }
    *kind = 2;
    return FWA_OK;
}

int32_t fwa_ctx_peer_access(fwa_ctx *ctx, fwa_ctx *peer, int32_t *kind)
{
    if (!ctx || !peer || !kind) return fail(ctx, FWA_ERR_INVALID_ARG, "NULL argument");
    return peer_kind(ctx, peer, kind);
}

int32_t fwa_buf_copy(fwa_buf *dst, uint64_t dst_offset, const fwa_buf *src, uint64_t src_offset, uint64_t bytes,
                     fwa_stream *stream)
{
    if (!dst || !src) return fail(nullptr, FWA_ERR_INVALID_ARG, "dst/src is NULL");
    if (dst_offset > dst->bytes || bytes > dst->bytes - dst_offset || src_offset > src->bytes ||
        bytes > src->bytes - src_offset)
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "copy range exceeds buffer");
    if (stream && stream->ctx != dst->ctx && stream->ctx != src->ctx)
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "the stream belongs to neither buffer's context");
    if (!bytes) return FWA_OK;
    char *d = static_cast<char *>(dst->p) + dst_offset;
    const char *s = static_cast<const char *>(src->p) + src_offset;
    fwa_ctx *on = stream ? stream->ctx : dst->ctx;  // the copy is enqueued on a stream of this context's device
    if (dst->ctx->device == src->ctx->device) {
        USE_DEVICE(on);
        HIP_TRY(on, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, raw(stream)));
        return FWA_OK;
    }
    // two devices (one process driving several contexts: SURVEY.md 8(e)): an explicit peer copy, or a status code -- never a
    // pointer the current device cannot reach handed to a plain device-to-device copy
    int32_t kind = 0;
    int32_t st = peer_kind(dst->ctx, const_cast<fwa_ctx *>(src->ctx), &kind);
    if (st) return st;
    if (kind != 2)
        return fail(dst->ctx, FWA_ERR_UNSUPPORTED,
                    "devices " + std::to_string(src->ctx->device) + " and " + std::to_string(dst->ctx->device) +
                        " have no peer access: stage through the host (fwa_buf_download / fwa_buf_upload) or move the slab with fwa_comm_*");
    USE_DEVICE(on);
    HIP_TRY(on, hipMemcpyPeerAsync(d, dst->ctx->device, s, src->ctx->device, bytes, raw(stream)));
    return FWA_OK;
}

int32_t fwa_host_alloc(fwa_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    if (!bytes) return FWA_OK;
    USE_DEVICE(ctx);
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail_hip(ctx, e, "hipHostMalloc", FWA_ERR_OUT_OF_MEMORY);
    return FWA_OK;
}

int32_t fwa_host_free(fwa_ctx *ctx, void *ptr)
{
    if (!ptr) return FWA_OK;
    HIP_TRY(ctx, hipHostFree(ptr));
    return FWA_OK;
}

int32_t fwa_buf_download_async(void *host, const fwa_buf *src, uint64_t src_offset, uint64_t bytes, fwa_stream *stream)
{
    if (!src || (!host && bytes)) return fail(src ? src->ctx : nullptr, FWA_ERR_INVALID_ARG, "src/host is NULL");
    if (src_offset > src->bytes || bytes > src->bytes - src_offset)
        return fail(src->ctx, FWA_ERR_INVALID_ARG, "download range exceeds buffer");
    if (!bytes) return FWA_OK;
    USE_DEVICE(src->ctx);
    HIP_TRY(src->ctx, hipMemcpyAsync(host, static_cast<const char *>(src->p) + src_offset, bytes,
                                     hipMemcpyDeviceToHost, raw(stream)));
    return FWA_OK;
}

int32_t fwa_stream_wait_stream(fwa_stream *stream, fwa_stream *other)
{
    if (!stream || !other || !stream->ctx) return fail(nullptr, FWA_ERR_INVALID_ARG, "stream is NULL or its context has been destroyed");
    USE_DEVICE(stream->ctx);
    hipEvent_t ev;
    HIP_TRY(stream->ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, other->s);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream->s, ev, 0);
    (void)hipEventDestroy(ev);  // destruction is deferred by the runtime until the event has completed
    if (e != hipSuccess) return fail_hip(stream->ctx, e, "hipEventRecord/hipStreamWaitEvent");
    return FWA_OK;
}

void *fwa_buf_device_ptr(const fwa_buf *buf) { return buf ? buf->p : nullptr; }
uint64_t fwa_buf_size(const fwa_buf *buf) { return buf ? buf->bytes : 0; }

// ---- plans ----------------------------------------------------------------
int32_t fwa_plan_destroy(fwa_plan *plan)
{
    if (!plan) return FWA_OK;
    (void)hipSetDevice(plan->ctx->device);
    // work of this plan may still be in flight on the caller's stream; the pooled ring must not be handed to the
    // next plan before it has drained (hipFree would have synchronised implicitly).  Only THIS plan's last exec is
    // waited for, through a marker on the stream that exec was enqueued on -- other streams and contexts keep running (a
    // device-wide synchronise here stalled them all and is illegal while any stream captures a graph).  An exec that was
    // captured into a graph enqueued nothing real: the plan must outlive the graphs that replay it.
    if (plan->frozen && plan->ring) {
        hipEvent_t ev = nullptr;
        bool waited = false;
        // only a stream that is known to be alive can take the marker: the null stream, or a stream of this context's own
        // making that has not been destroyed (HIP does not validate stream handles); otherwise the device-wide wait below
        const auto &us = plan->ctx->user_streams;
        const bool alive = plan->last_stream == nullptr || std::find(us.begin(), us.end(), plan->last_stream) != us.end();
        if (plan->ran_on_stream && alive && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            // everything this plan enqueued on its last caller stream precedes this marker
            if (hipEventRecord(ev, plan->last_stream) == hipSuccess) waited = hipEventSynchronize(ev) == hipSuccess;
            (void)hipEventDestroy(ev);
        }
        if (!waited) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();  // the stream is gone or not ours (fwa_stream_wrap), or a laboratory persistent path
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
    if (reinterpret_cast<uintptr_t>(src->p) & 15) return fail(ctx, FWA_ERR_INVALID_ARG, "buffer must be 16-byte aligned");

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
            if (plan->small_reg != 1) {  // laboratory kernels (A/B): direct 16-point kernels, shuffle exchange, LDS radix 2
                if (plan->small_reg && plan->n < 16)
                    e = fwa::launch_tiny(dir, a, out, plan->n, plan->batch, scale, st);
                else if (plan->small_reg && plan->n >= 512 && (plan->small_reg != 3 || plan->n > 4096))
                    e = fwa::launch_small32(dir, a, out, tb.tw_half, plan->n, plan->batch, scale, st);
                else if (plan->small_reg)
                    e = fwa::launch_small16(dir, a, out, tb.tw_half, plan->n, plan->batch, scale, plan->small_reg == 2, st);
                else
                    e = fwa::launch_lds_small(dir, a, out, tb.tw_half, plan->n, plan->batch, scale, st);
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
            const int w = (int)plan->tile_w;
            const v2f *two = tb.tw_outer[w == 32 ? 1 : 0];
            // default: XCD-contiguous tiles + adjacent tiles on the two residents of a CU (bit 2: + 4-8 % for launches that have the
            // chip to themselves, + 0.6 % with two chains in flight: tile_1m.h xcd_block, profiles/round4/sweep_pair_map_two_chains.jsonl)
            const uint32_t swz = plan->xcd_swizzle < 0 ? 5u : (uint32_t)plan->xcd_swizzle;
            return run_groups(plan, st, [&](uint64_t g, uint64_t cnt, hipStream_t s, size_t c) {
                // ring region of this chain: transform i -> slot i (ring_rotate > 1, laboratory: successive groups of a
                // chain walk through ring_rotate such regions)
                v2f *slab = plan->ring + ((g / (plan->istreams.empty() ? 1 : plan->istreams.size())) % (uint64_t)plan->ring_rotate) *
                                             (uint64_t)plan->n_streams * G * N + (uint64_t)c * G * N;
                hipError_t le = fwa::launch_p1_1m(dir, w, a + g * G * N, slab, tb.tw_inner, two, (uint32_t)cnt, swz, s);
                if (le != hipSuccess) return le;
                return fwa::launch_p2_1m(dir, w, slab, out + g * G * N, tb.tw_inner, (uint32_t)cnt, scale, swz, s);
            });
        }
#ifdef FWA_LAB
        case PATH_RING_1M: {
            if (!plan->ring || !plan->ring_ctl) return fail(ctx, FWA_ERR_INVALID_ARG, "plan has no scratch ring (a failed re-tune?)");
            const uint64_t slots = (uint64_t)plan->ring_slots < plan->batch ? (uint64_t)plan->ring_slots : plan->batch;
            const uint64_t depth = (uint64_t)plan->depth < slots ? (uint64_t)plan->depth : (slots > 1 ? slots - 1 : 1);
            e = fwa::launch_ring_1m(dir, a, out, plan->ring, tb.tw_inner, tb.tw_outer[0], plan->ring_ctl, (uint32_t)plan->batch,
                                    (uint32_t)depth, (uint32_t)(slots > depth ? slots : depth + 1), (uint32_t)plan->wgs, scale, st);
            break;
        }
        case PATH_TEAM: {
            if (!plan->ring || !plan->ring_ctl) return fail(ctx, FWA_ERR_INVALID_ARG, "plan has no team slabs (a failed re-tune?)");
            e = fwa::launch_team(dir, plan->lg, a, out, plan->ring, tb.tw_l[0], tb.tw_lo1, tb.tw_hi1, tb.tw_l[1], plan->ring_ctl,
                                 (uint32_t)plan->batch, (uint32_t)plan->max_teams, (uint32_t)plan->wgs, scale, st);
            break;
        }
#endif
        case PATH_TILED: {
            // n = N1*N2[*N3]; index n = (n1*N2 + n2)*N3 + n3, k = k1 + N1*(k2 + N2*k3).  Per group of transforms:
            // pass A: FFT over n1 (cols, twiddle W_n), user buffer -> ring slab; [pass B: FFT over n2 per k1 (cols,
            // twiddle W_{N2*N3}), in place in the slab]; pass C: FFT over the contiguous axis with the transposed
            // store, slab -> result buffer (src for even log2 n -- in place at group granularity -- else second).
            const bool three = plan->lf[2] != 0;
            const uint64_t N1 = 1ull << plan->lf[0], N2 = 1ull << plan->lf[1], N3 = three ? (1ull << plan->lf[2]) : 1;
            return run_groups(plan, st, [&](uint64_t g, uint64_t cnt, hipStream_t s, size_t c) {
                v2f *slab = plan->ring + (uint64_t)c * G * N;
                fwa::TileArgs ta{};
                ta.xcd_swizzle = plan->xcd_swizzle < 0 ? tiled_swizzle_default(plan) : (uint32_t)plan->xcd_swizzle;
                // pass A
                uint32_t cw = pass_cw(plan, 0);
                ta.in = a + g * G * N; ta.out = slab; ta.tw = tb.tw_l[0]; ta.tw_lo = tb.tw_lo1; ta.tw_hi = tb.tw_hi1;
                ta.scale = 1.0f; ta.cw = cw; ta.role = fwa::ROLE_FIRST;
                ta.in_sb = ta.out_sb = N; ta.in_s1 = ta.out_s1 = 0; ta.in_st = ta.out_st = cw;
                ta.pitch = N / N1; ta.out_stride = 0; ta.d1_count = 1; ta.tile_count = (uint32_t)(N / N1 / cw);
                hipError_t le;
                // k_colsw writes the slab tile-contiguously when the last pass (k_rows32) can read that layout back
                const bool use_colsw = plan->colsw && fwa::colsw_supported(plan->lf[0]) && plan->lg <= 28;
                const uint32_t ring_cw = (use_colsw && plan->tile_ring && !three && fwa::rows32_ring_supported(plan->lf[1], fwa::colsw_width(plan->lf[0])))
                                             ? fwa::colsw_width(plan->lf[0]) : 0u;
                if (use_colsw)
                    le = fwa::launch_colsw(dir, plan->lf[0], true, ring_cw != 0, ta.in, slab, tb.tw_l[0], tb.tw_lo1, tb.tw_hi1, (uint32_t)(N / N1),
                                           N, N, (uint32_t)cnt, ta.xcd_swizzle, s);
                else if (plan->lf[0] > 10)
                    le = fwa::launch_cols32(dir, plan->lf[0], true, ta.in, slab, tb.tw_l[0], tb.tw_lo1, tb.tw_hi1, (uint32_t)(N / N1), N,
                                            N, (uint32_t)cnt, ta.xcd_swizzle, s);
                else if (plan->lf[0] == 10 && plan->p1_gen && tb.tw_inner)
                    le = fwa::launch_p1_gen(dir, true, ta.in, slab, tb.tw_inner, tb.tw_lo1, tb.tw_hi1, (uint32_t)(N / N1), N, N,
                                            (uint32_t)cnt, ta.xcd_swizzle, s);
                else
                    le = fwa::launch_tile(dir, fwa::TILE_COLS, plan->lf[0], ta, cnt, s);
                if (le != hipSuccess) return le;
                if (three) {  // pass B, in place in the slab
                    cw = pass_cw(plan, 1);
                    if (N3 < cw) cw = 16;
                    ta.in = slab; ta.out = slab; ta.tw = tb.tw_l[1]; ta.tw_lo = tb.tw_lo_b; ta.tw_hi = tb.tw_hi_b;
                    ta.cw = cw; ta.role = fwa::ROLE_MIDDLE; ta.in_st = ta.out_st = cw;
                    ta.in_s1 = ta.out_s1 = N2 * N3; ta.pitch = N3; ta.d1_count = (uint32_t)N1;
                    ta.tile_count = (uint32_t)(N3 / cw);
                    le = fwa::launch_tile(dir, fwa::TILE_COLS, plan->lf[1], ta, cnt, s);
                    if (le != hipSuccess) return le;
                }
                // pass C: rows of the last axis, cw adjacent k1 per tile
                if (!three && plan->lg <= 28 && fwa::rows32_supported(plan->lf[1]) && (plan->rows32 || plan->lf[1] > 10 || ring_cw))
                    return fwa::launch_rows32(dir, plan->lf[1], slab, out + g * G * N, tb.tw_l[1], (uint32_t)N1, N, N, (uint32_t)cnt,
                                              scale, ta.xcd_swizzle, ring_cw, s);
                const uint32_t li = three ? 2 : 1;
                cw = pass_cw(plan, li);
                ta.in = slab; ta.out = out + g * G * N; ta.tw = tb.tw_l[li]; ta.tw_lo = nullptr; ta.tw_hi = nullptr;
                ta.scale = scale; ta.cw = cw; ta.role = fwa::ROLE_LAST;
                ta.in_sb = ta.out_sb = N;
                ta.pitch = N / N1;  // distance between the rows k1 and k1+1
                ta.in_st = cw * (N / N1); ta.out_st = cw; ta.tile_count = (uint32_t)(N1 / cw);
                if (three) { ta.d1_count = (uint32_t)N2; ta.in_s1 = N3; ta.out_s1 = N1; ta.out_stride = N1 * N2; }
                else { ta.d1_count = 1; ta.in_s1 = ta.out_s1 = 0; ta.out_stride = N1; }
                return fwa::launch_tile(dir, fwa::TILE_ROWS_T, plan->lf[li], ta, cnt, s);
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

int32_t fwa_plan_get_i64(const fwa_plan *plan, const char *key, int64_t *value)
{
    if (!plan || !key || !value) return fail(plan ? plan->ctx : nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    const std::string k(key);
    const int64_t ng = plan->group ? (int64_t)((plan->batch + plan->group - 1) / plan->group) : 0;
    if (k == "batch") *value = (int64_t)plan->batch;
    else if (k == "fft_len") *value = plan->n;
    else if (k == "path") *value = plan->path;
    else if (k == "group") *value = plan->group;
    else if (k == "streams") *value = plan->n_streams;
    else if (k == "tile_w") *value = plan->tile_w;
    else if (k == "xcd_swizzle") *value = plan->xcd_swizzle < 0 ? (plan->path == PATH_TWOPASS_1M ? 5 : (plan->path == PATH_TILED ? (int64_t)tiled_swizzle_default(plan) : 0)) : plan->xcd_swizzle;
    else if (k == "depth") *value = plan->depth;
    else if (k == "ring_slots") *value = plan->ring_slots;
    else if (k == "wgs") *value = plan->wgs;
    else if (k == "max_teams") *value = plan->max_teams;
    else if (k == "device_error") {
        // bounded-spin timeout flag of the persistent kernel (0 in every healthy run); synchronises the device
        *value = 0;
        if (plan->ring_ctl) {
            uint32_t w = 0;
            HIP_TRY(plan->ctx, hipDeviceSynchronize());
            HIP_TRY(plan->ctx, hipMemcpy(&w, plan->ring_ctl + 1, sizeof(w), hipMemcpyDeviceToHost));
            *value = w;
        }
    }
    else if (k == "small_reg") *value = plan->small_reg;
    else if (k == "p1_gen") *value = plan->p1_gen;
    else if (k == "rows32") *value = plan->rows32;
    else if (k == "colsw") *value = plan->colsw;
    else if (k == "tile_ring") *value = plan->tile_ring;
    else if (k == "ring_rotate") *value = plan->ring_rotate;
    else if (k == "factors") *value = plan->lf[0] | (plan->lf[1] << 8) | (plan->lf[2] << 16);
    else if (k == "tables_shared") *value = plan->tb ? (int64_t)plan->tb.use_count() - 1 : 0;  // other holders: cache + plans
    else if (k == "scratch_bytes")
        *value = (int64_t)plan->ring_bytes + (plan->second_owned ? (int64_t)plan->own_second.bytes : 0) +
                 (plan->ring_ctl ? (int64_t)ctl_bytes(plan) : 0);
    else if (k == "launches_per_exec") {
        switch (plan->path) {
            case PATH_TWOPASS_1M: *value = 2 * ng; break;
            case PATH_RING_1M: case PATH_TEAM: *value = 1; break;
            case PATH_TILED: *value = (plan->lf[2] ? 3 : 2) * ng; break;
            case PATH_R2_GLOBAL: *value = plan->lg; break;
            case PATH_IDENTITY: *value = (plan->kind == FWA_INVERSE_SCALED) ? 1 : 0; break;
            default: *value = 1;
        }
    } else return fail(plan->ctx, FWA_ERR_INVALID_ARG, "unknown key: " + k);
    return FWA_OK;
}

int32_t fwa_plan_set_i64(fwa_plan *plan, const char *key, int64_t value)
{
    if (!plan || !key) return fail(plan ? plan->ctx : nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    if (plan->frozen) return fail(plan->ctx, FWA_ERR_INVALID_ARG, "plan tunables are locked after the first exec");
    fwa_ctx *ctx = plan->ctx;
    USE_DEVICE(ctx);
    const std::string k(key);
    if (k == "group" || k == "streams") {
        if (plan->path != PATH_TWOPASS_1M && plan->path != PATH_TILED)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the pipelined paths (2^20 two-pass, tiled)");
        if (value < 1 || value > (k == "streams" ? 16 : 4096)) return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        return build_pipeline(plan, k == "group" ? value : plan->group, k == "streams" ? value : plan->n_streams);
    }
    if (k == "inject_launch_failure") {
        // laboratory: the launch of group `value` fails once (nothing is enqueued for it): the error path of run_groups
        if (!kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "inject_launch_failure is a laboratory knob (libfft_wgpu_amd_lab.so)");
        if (plan->path != PATH_TWOPASS_1M && plan->path != PATH_TILED) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the pipelined paths");
        plan->inject_fail_group = value;
        return FWA_OK;
    }
    if (k == "ring_rotate") {
        if (!kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "ring_rotate is a laboratory knob (libfft_wgpu_amd_lab.so)");
        if (plan->path != PATH_TWOPASS_1M) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the 2^20 two-pass path");
        if (value < 1 || value > 64) return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        const int64_t old = plan->ring_rotate;
        plan->ring_rotate = value;
        const int32_t st = build_pipeline(plan, plan->group, plan->n_streams);
        if (st) plan->ring_rotate = old;
        return st;
    }
    if (k == "tile_w") {
        if (plan->path != PATH_TWOPASS_1M) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the 2^20 two-pass path");
        if (value != 16 && value != 32) return fail(ctx, FWA_ERR_INVALID_ARG, "tile_w is 16 or 32");
        if (value == 32 && !kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "tile_w = 32 is a laboratory variant (libfft_wgpu_amd_lab.so)");
        plan->tile_w = value;
        return FWA_OK;
    }
#ifdef FWA_LAB
    if (k == "max_teams" || (k == "wgs" && plan->path == PATH_TEAM)) {
        if (plan->path != PATH_TEAM) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the team path");
        if (value < 1 || value > 65536) return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        if (k == "wgs") {
            // a team only forms from workgroups of ONE XCD and blocks are dealt round-robin over the 8 XCDs: fewer than
            // 8 x team size workgroups may leave every XCD short of a team and the launch would transform nothing
            uint32_t ts = 0, th = 0;
            size_t lds = 0;
            fwa::team_geometry(plan->lg, &ts, &th, &lds);
            if (value < 8 * (int64_t)ts) return fail(ctx, FWA_ERR_INVALID_ARG, "wgs must be at least 8 x the team size");
            plan->wgs = value;
            return FWA_OK;
        }
        plan->max_teams = value;
        return build_pipeline(plan, 0, 0);
    }
#endif
    if (k == "max_teams" || k == "depth" || k == "ring_slots" || k == "wgs") {
        if (!kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "key belongs to a laboratory path (libfft_wgpu_amd_lab.so)");
        if (plan->path != PATH_RING_1M) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the persistent 2^20 path");
        if (value < 1 || value > 65536) return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        if (k == "wgs") { plan->wgs = value; return FWA_OK; }
        const int64_t d = k == "depth" ? value : plan->depth, r = k == "ring_slots" ? value : plan->ring_slots;
        if (k == "depth") { plan->depth = d; if (r < d + 1) plan->ring_slots = d + 1; }
        else { if (r < plan->depth + 1) return fail(ctx, FWA_ERR_INVALID_ARG, "ring_slots must exceed depth"); plan->ring_slots = r; }
        return build_pipeline(plan, 0, 0);
    }
    if (k == "xcd_swizzle") {
        if (plan->path != PATH_TWOPASS_1M && plan->path != PATH_TILED)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the pipelined paths");
        plan->xcd_swizzle = value & 7;
        return FWA_OK;
    }
    if (k == "factors") {
        // re-factorise a multi-pass plan: value = log2(N1) | log2(N2) << 8 | log2(N3) << 16 (N3 = 0: two passes), every
        // factor 64..1024, product n.  A tuning knob: every factorisation computes the same transform.
        if (plan->path != PATH_TILED && plan->path != PATH_TWOPASS_1M)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to multi-pass plans");
        const uint32_t f[3] = {(uint32_t)value & 255u, (uint32_t)(value >> 8) & 255u, (uint32_t)(value >> 16) & 255u};
        const uint32_t nf = f[2] ? 3 : 2;
        uint32_t sum = 0;
        for (uint32_t i = 0; i < nf; ++i) {
            // 2048: as the first factor (k_cols32); 2048 / 4096: as the second of two (k_rows32); n <= 2^28
            const uint32_t top = plan->lg > 28 ? 10u : (i == 0 ? 11u : ((nf == 2 && i == 1) ? 12u : 10u));
            if (f[i] < 6 || f[i] > top) return fail(ctx, FWA_ERR_INVALID_ARG, "every factor must be 2^6..2^10 (2^11: first; 2^11, 2^12: second of two)");
            sum += f[i];
        }
        if (sum != plan->lg || (value >> 24)) return fail(ctx, FWA_ERR_INVALID_ARG, "factors do not multiply to fft_len");
        const int64_t old_path = plan->path;
        uint32_t old_lf[3] = {plan->lf[0], plan->lf[1], plan->lf[2]};
        plan->path = PATH_TILED;
        plan->lf[0] = f[0]; plan->lf[1] = f[1]; plan->lf[2] = f[2];
        const int32_t st = setup_path(plan);
        if (st) { plan->path = old_path; plan->lf[0] = old_lf[0]; plan->lf[1] = old_lf[1]; plan->lf[2] = old_lf[2]; }
        return st;
    }
    if (k == "p1_gen" || k == "rows32" || k == "colsw" || k == "tile_ring") {
        if (plan->path != PATH_TILED) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to tiled plans");
        (k == "p1_gen" ? plan->p1_gen : k == "rows32" ? plan->rows32 : k == "colsw" ? plan->colsw : plan->tile_ring) = value != 0;
        return FWA_OK;
    }
    if (k == "small_reg") {
        if (plan->path != PATH_SMALL) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to n <= 32768");
        if (value != 1 && !kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "small_reg != 1 selects laboratory kernels (libfft_wgpu_amd_lab.so)");
        if (!value && plan->n > 4096) return fail(ctx, FWA_ERR_UNSUPPORTED, "the LDS radix-2 kernel stops at n = 4096");
        // 1: k_chunk (4 .. 256) and k_small32 (from 512), the default; 3: the direct-addressing kernels k_tiny16 /
        // k_small16 up to 4096 (A/B); 2: as 3 with the wavefront-shuffle exchange at n = 32/64/128; 0: LDS radix-2 kernel
        plan->small_reg = (value >= 0 && value <= 3) ? value : 1;
        return FWA_OK;
    }
    if (k == "path") {
        if (plan->kind == FWA_NORMALIZE) return fail(ctx, FWA_ERR_UNSUPPORTED, "normalize has one path");
        if (value == plan->path) return FWA_OK;
        if ((value == PATH_RING_1M || value == PATH_TEAM) && !kLab)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "paths 5 and 8 are laboratory paths (libfft_wgpu_amd_lab.so)");
        if ((value == PATH_RING_1M || value == PATH_TWOPASS_1M) && (plan->path == PATH_RING_1M || plan->path == PATH_TWOPASS_1M)) {
            // the two forms of the 2^20 pipeline: per-group launches with a large ring, or one persistent launch
            const int64_t old = plan->path;
            plan->path = value;
            const int32_t st = setup_path(plan);
            if (st) plan->path = old;
            if (!st && value == PATH_TWOPASS_1M && plan->ring_ctl) { (void)hipFree(plan->ring_ctl); plan->ring_ctl = nullptr; }
            return st;
        }
        if ((value == PATH_TEAM || value == PATH_TILED) && (plan->path == PATH_TEAM || plan->path == PATH_TILED)) {
#ifdef FWA_LAB
            if (value == PATH_TEAM && !fwa::team_supported(plan->lg))
                return fail(ctx, FWA_ERR_UNSUPPORTED, "the team path covers n = 2^16 .. 2^18");
#endif
            const int64_t old = plan->path;
            uint32_t old_lf[3] = {plan->lf[0], plan->lf[1], plan->lf[2]};
            plan->path = value;
            if (value == PATH_TEAM) { plan->lf[0] = plan->lg / 2; plan->lf[1] = plan->lg - plan->lf[0]; plan->lf[2] = 0; }
            else { bool cw = false; (void)choose_path(plan->n, plan->batch, plan->lf, &cw); plan->colsw = cw; }
            const int32_t st = setup_path(plan);
            if (st) { plan->path = old; plan->lf[0] = old_lf[0]; plan->lf[1] = old_lf[1]; plan->lf[2] = old_lf[2]; }
            if (!st && value == PATH_TILED && plan->ring_ctl) { (void)hipFree(plan->ring_ctl); plan->ring_ctl = nullptr; }
            return st;
        }
        if (value == PATH_R2_GLOBAL && plan->n >= 2) {
            // force the literal reference recurrence (one launch per stage, kernel/fft.wgsl:27-62)
            if (!plan->tb->tw_half && !plan->tw_half_private) {
                int32_t st = upload_half_table(ctx, plan->n, &plan->tw_half_private);
                if (st) return st;
            }
            if (!plan->second->p && plan->src->bytes) {
                if (plan->second != &plan->own_second) return fail(ctx, FWA_ERR_INVALID_ARG, "second buffer missing");
                hipError_t e = hipMalloc(&plan->own_second.p, plan->src->bytes);
                if (e != hipSuccess) return fail_hip(ctx, e, "hipMalloc(second buffer)");
                plan->own_second.bytes = plan->src->bytes;
                plan->second_owned = true;
            }
            Pipeline pl = take_pipeline(plan);
            destroy_pipeline_objects(ctx, pl, true);
            if (plan->ring_ctl) { (void)hipFree(plan->ring_ctl); plan->ring_ctl = nullptr; }
            plan->path = PATH_R2_GLOBAL;
            return FWA_OK;
        }
        return fail(ctx, FWA_ERR_UNSUPPORTED, "only path = 2 (the literal radix-2 recurrence) or, at n = 2^20, 1 / 5 can be set");
    }
    return fail(ctx, FWA_ERR_INVALID_ARG, "unknown key: " + k);
}

// ---- events ----------------------------------------------------------------
int32_t fwa_event_create(fwa_ctx *ctx, fwa_event **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    USE_DEVICE(ctx);
    hipEvent_t e;
    HIP_TRY(ctx, hipEventCreate(&e));
    fwa_event *ev = new (std::nothrow) fwa_event;
    if (!ev) { (void)hipEventDestroy(e); return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    ev->ctx = ctx; ev->e = e;
    *out = ev;
    return FWA_OK;
}
int32_t fwa_event_record(fwa_event *ev, fwa_stream *stream)
{
    if (!ev) return fail(nullptr, FWA_ERR_INVALID_ARG, "event is NULL");
    USE_DEVICE(ev->ctx);
    HIP_TRY(ev->ctx, hipEventRecord(ev->e, raw(stream)));
    return FWA_OK;
}
int32_t fwa_event_synchronize(fwa_event *ev)
{
    if (!ev) return fail(nullptr, FWA_ERR_INVALID_ARG, "event is NULL");
    USE_DEVICE(ev->ctx);
    HIP_TRY(ev->ctx, hipEventSynchronize(ev->e));
    return FWA_OK;
}
int32_t fwa_stream_wait_event(fwa_stream *stream, fwa_event *ev)
{
    if (!ev) return fail(nullptr, FWA_ERR_INVALID_ARG, "event is NULL");
    USE_DEVICE(ev->ctx);
    HIP_TRY(ev->ctx, hipStreamWaitEvent(raw(stream), ev->e, 0));
    return FWA_OK;
}
int32_t fwa_event_elapsed_ms(fwa_event *start, fwa_event *end, float *ms)
{
    if (!start || !end || !ms) return fail(nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    USE_DEVICE(end->ctx);
    HIP_TRY(end->ctx, hipEventSynchronize(end->e));
    HIP_TRY(end->ctx, hipEventElapsedTime(ms, start->e, end->e));
    return FWA_OK;
}
int32_t fwa_event_destroy(fwa_event *ev)
{
    if (!ev) return FWA_OK;
    (void)hipEventDestroy(ev->e);
    delete ev;
    return FWA_OK;
}

// ---- synthetic data / calibration -------------------------------------------
int32_t fwa_fill_synthetic(fwa_buf *dst, uint64_t seed, uint64_t first_transform, uint32_t fft_len, float scale,
                           fwa_stream *stream)
{
    if (!dst || !fft_len) return fail(dst ? dst->ctx : nullptr, FWA_ERR_INVALID_ARG, "dst NULL or fft_len 0");
    USE_DEVICE(dst->ctx);
    hipError_t e = fwa::launch_fill(static_cast<v2f *>(dst->p), seed, first_transform * (uint64_t)fft_len,
                                    dst->bytes / 8, scale, raw(stream));
    if (e != hipSuccess) return fail_hip(dst->ctx, e, "fill launch", FWA_ERR_LAUNCH);
    return FWA_OK;
}

int32_t fwa_calib_copy(fwa_buf *dst, const fwa_buf *src, uint64_t bytes, fwa_stream *stream)
{
    if (!dst || !src) return fail(nullptr, FWA_ERR_INVALID_ARG, "dst/src is NULL");
    if (bytes > dst->bytes || bytes > src->bytes || (bytes & 15))
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "copy size exceeds a buffer or is not a multiple of 16");
    USE_DEVICE(dst->ctx);
    // dst == src: an in-place streaming pass (every line read, then written back) -- the normalize kernel with scale 1:
    // 64-KiB chunk per workgroup, every wave walks 16 KiB with 32 nt loads in flight: the fastest streaming shape on this part
    hipError_t e = (dst->p == src->p) ? fwa::launch_scale(static_cast<const v2f *>(src->p), static_cast<v2f *>(dst->p), bytes / 8, 1.0f, raw(stream))
                                      : fwa::launch_copy(src->p, dst->p, bytes, raw(stream));
    if (e != hipSuccess) return fail_hip(dst->ctx, e, "copy launch", FWA_ERR_LAUNCH);
    return FWA_OK;
}

}  // extern "C"

namespace fwa_int {
int32_t fail(const fwa_ctx *ctx, int32_t status, const std::string &msg) { return ::fail(ctx, status, msg); }
int32_t fail_hip(const fwa_ctx *ctx, hipError_t e, const char *what) { return ::fail_hip(ctx, e, what); }
int32_t use_device(fwa_ctx *ctx)
{
    USE_DEVICE(ctx);
    return FWA_OK;
}
int ctx_device(const fwa_ctx *ctx) { return ctx->device; }
fwa_ctx *buf_ctx(const fwa_buf *b) { return b->ctx; }
hipStream_t stream_raw(fwa_stream *s) { return s ? s->s : nullptr; }
fwa_ctx *stream_ctx(fwa_stream *s) { return s ? s->ctx : nullptr; }
}  // namespace fwa_int
