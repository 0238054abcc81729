// internal.h -- handles, shared helpers and the cross-file interface of the library's host side (not installed).
//
// The host side of the C ABI (include/fft_wgpu_amd.h) is split by concern:
//   ctx_streams.cpp  contexts, error strings, device enumeration, the slab rule, streams (+ the overlap check), events
//   buffers.cpp      device / pinned memory, uploads, downloads, copies (peer copies across contexts), synthetic data
//   tables.cpp       twiddle tables (reference twiddle rule src/processor.rs:43-49), the ring pool, pipeline objects
//   plan.cpp         path choice, plan create / destroy / exec (the four plan objects of src/processor.rs)
//   tuning.cpp       fwa_plan_get_i64 / fwa_plan_set_i64
//   comm.cpp         fwa_comm_*: slab movement over RCCL
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_set>
#include <vector>

#include "../../include/fft_wgpu_amd.h"
#include "kernels.h"

using fwa::v2f;

// FWA_LAB (libfft_wgpu_amd_lab.so, `make lab`): the same ABI plus the kernel families that measured slower than the
// shipped ones -- path 5 (persistent 2^20 ring), small_reg = 2 / 3 (direct 16-point kernels, wavefront-shuffle exchange) --
// and two test knobs (ring_rotate, inject_launch_failure).  The product library rejects those settings.
#ifdef FWA_LAB
constexpr bool kLab = true;
#else
constexpr bool kLab = false;
#endif

enum fwa_path : int64_t {
    PATH_SMALL = 0,       // n <= 32768: one launch (k_chunk / k_small32)
    PATH_TWOPASS_1M = 1,  // n = 2^20: k_p1_1m + k_p2_1m per group of transforms
    PATH_R2_GLOBAL = 2,   // the reference recurrence literally, one launch per stage (forced only)
    PATH_NORMALIZE = 3,
    PATH_IDENTITY = 4,    // n = 1
    PATH_RING_1M = 5,     // n = 2^20: the same two passes as ONE persistent launch with a small ring (k_ring_1m)
    // 8 was the L2-resident team path (k_team, rounds 2-5; removed in round 6: profiles/round6/lab_pruned_families.patch)
    PATH_TILED = 7,       // n = N1*N2[*N3], each 64..1024: 2-3 k_tile passes
};

namespace fwa_int {

// Device tables of one transform length, shared by every plan of that length on a context (plan cache):
// tables hold forward twiddles only (the inverse conjugates on use), so all plan kinds share them.
struct Tables {
    v2f *tw_half = nullptr;   // n/2 entries, processor.rs:43-49 (small / literal paths)
    v2f *tw_inner = nullptr;  // 2^20 path: [k1][n'] = W_1024^{n' k1}
    v2f *tw_outer = nullptr;  // 2^20 path: per 16-column tile A[32][16], B[32][16]
    v2f *tw_l[3] = {nullptr, nullptr, nullptr};  // tiled path: per-factor W_L tables
    v2f *tw_lo1 = nullptr, *tw_hi1 = nullptr;    // four-step tables of pass A (domain n)
    v2f *tw_lo_b = nullptr, *tw_hi_b = nullptr;  // four-step tables of pass B (domain N2*N3)
    ~Tables()
    {
        for (v2f *t : {tw_half, tw_inner, tw_outer, tw_l[0], tw_l[1], tw_l[2], tw_lo1, tw_hi1, tw_lo_b,
                       tw_hi_b})
            if (t) (void)hipFree(t);
    }
};

}  // namespace fwa_int

struct fwa_ctx {
    int device = -1;
    hipDeviceProp_t prop{};
    mutable std::string err;
    bool setup_1m_done = false;
    bool setup_small_done = false;
    // plan cache: (fft_len, path, factor signature) -> tables; ring allocations of destroyed plans by size
    std::map<std::tuple<uint32_t, int64_t, uint32_t>, std::shared_ptr<fwa_int::Tables>> tables;
    std::vector<std::pair<uint64_t, void *>> free_rings;  // rings of destroyed plans, oldest first
    uint64_t free_ring_bytes = 0;
    int64_t n_table_builds = 0, n_table_hits = 0, n_ring_allocs = 0, n_ring_reuses = 0, last_plan_create_us = 0;
    // Internal chain streams of the pipelined paths: created once per context, shared by every plan, and checked at
    // creation to run kernels side by side (chain_streams() below).
    std::vector<hipStream_t> chains;
    std::vector<hipStream_t> user_streams;  // alive streams made by fwa_stream_create, oldest first
    std::vector<fwa_stream *> live_streams; // every alive fwa_stream handle of this context (created or wrapped)
    int64_t n_chain_checks = 0, n_chain_rejects = 0, chain_pair_us = 0, chain_single_us = 0;
    // fwa_ctx_set_i64("chain_check", 0): new streams are taken as the runtime hands them out
    int64_t chain_check = 1;
    std::vector<int> peers_enabled;         // device ordinals this context's device has peer access to (enabled once)
    // Every alive fwa_buf / fwa_event handle of this context (allocated or wrapped).  fwa_ctx_destroy clears their `ctx`,
    // as it does for streams, so that a handle which outlives its context (garbage-collected hosts free in any order)
    // answers FWA_ERR_INVALID_ARG instead of dereferencing freed memory.  `live_mu` guards the two sets and the stream list
    // below only: buffers may be allocated and freed from several threads of one context.
    std::mutex live_mu;
    std::unordered_set<fwa_buf *> live_bufs;
    std::unordered_set<fwa_event *> live_events;
};
struct fwa_stream {
    fwa_ctx *ctx = nullptr;   // nullptr once the context has been destroyed (the handle can still be destroyed)
    hipStream_t s = nullptr;
    bool owned = false;
    int device = -1;
};
struct fwa_buf {
    fwa_ctx *ctx = nullptr;   // nullptr once the context has been destroyed: every entry point then refuses the handle
    void *p = nullptr;
    uint64_t bytes = 0;
    bool owned = false;
    int device = -1;          // for fwa_buf_free after the context is gone
};
struct fwa_event {
    fwa_ctx *ctx = nullptr;   // nullptr once the context has been destroyed
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
    std::shared_ptr<fwa_int::Tables> tb;    // shared through ctx->tables
    v2f *tw_half_private = nullptr;  // forced literal path on a size whose cached tables have no n/2 table
    uint32_t lf[3] = {0, 0, 0};    // tiled path: log2 of the factors (lf[2] = 0 for two factors)
    // pipeline state (two-pass 2^20 and tiled paths): groups of transforms alternate over internal streams
    v2f *ring = nullptr;
    uint64_t ring_bytes = 0;
    int64_t group = 16;            // transforms per launch
    int64_t n_streams = 2;         // internal streams (chains) the groups alternate over
    // XCD-aware block -> tile mapping (xcd_map bits): -1 = per-path, per-size default (5 on the 2^20 two-pass path;
    // tiled plans: tiled_swizzle_default, plan.cpp)
    int64_t xcd_swizzle = -1;
    int64_t rows32 = 1;            // two-pass tiled plans with a 512..4096-point second factor: 1 = k_rows32 last
    int64_t p1_gen = 1;            // tiled plans with first factor 1024: 1 = k_p1_gen as pass A, 0 = k_tile
    int64_t colsw = 0;             // tiled plans with first factor 256 / 512: 1 = k_colsw (64 / 32-column tiles) first
    int64_t tile_ring = 1;         // k_colsw + k_rows32: 1 = tile-contiguous ring slab, 0 = matrix layout
    // laboratory: the ring is this many times larger and the groups rotate through it (same launches, larger cache
    // footprint: prices what the Infinity Cache gives the ring)
    int64_t ring_rotate = 1;
    // n <= 32768: 1 = k_chunk / k_small32; laboratory: 3 = direct 16-point kernels (16 .. 4096), 2 = + wave shuffles
    int64_t small_reg = 1;
    std::vector<hipStream_t> istreams;
    std::vector<hipEvent_t> idone;
    hipEvent_t ev_fork = nullptr;
    // the caller's stream of the last exec that used the ring: fwa_plan_destroy waits for the work enqueued there (an
    // event per exec would cost 4-5 us on the 1-3-launch latency shapes)
    hipStream_t last_stream = nullptr;
    bool ran_on_stream = false;
    // persistent 2^20 pipeline (PATH_RING_1M)
    uint32_t *ring_ctl = nullptr;  // ticket, error word, per-transform hand-off counters
    int64_t depth = 8;             // pass-2 tiles of transform t run beside pass-1 tiles of transform t + depth
    int64_t ring_slots = 12;       // transforms of intermediate kept (>= depth + 1)
    int64_t wgs = 512;             // persistent workgroups (2 per CU)
    // laboratory: the launch of this group fails once (error path of run_groups under test)
    int64_t inject_fail_group = -1;
};

namespace fwa_int {

// ---- errors (ctx_streams.cpp): record the message on the context (or thread-locally without one), return the status
int32_t fail(const fwa_ctx *ctx, int32_t status, const std::string &msg);
int32_t fail_hip(const fwa_ctx *ctx, hipError_t e, const char *what, int32_t status = FWA_ERR_HIP);
const char *thread_error_string();

#define HIP_TRY(ctx, call)                                                \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fwa_int::fail_hip((ctx), e_, #call); \
    } while (0)

// Every entry point that touches the device makes the context's device current first: with one context per
// device in one process work must not land on whichever device was used last.
#define USE_DEVICE(ctx)                                                                                   \
    do {                                                                                                  \
        int cur_ = -1;                                                                                    \
        if (hipGetDevice(&cur_) != hipSuccess || cur_ != (ctx)->device) HIP_TRY((ctx), hipSetDevice((ctx)->device)); \
    } while (0)

inline bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }
inline uint32_t ilog2(uint32_t n)
{
    uint32_t l = 0;
    while ((1u << l) < n) ++l;
    return l;
}
inline hipStream_t raw(fwa_stream *s) { return s ? s->s : nullptr; }

// A buffer / event handle whose context has been destroyed may still be freed, never used (include/fft_wgpu_amd.h,
// "Lifetimes"): every entry point that takes one starts with this check.
#define LIVE_HANDLE(h, what)                                                                              \
    do {                                                                                                  \
        if (!(h)->ctx)                                                                                    \
            return fwa_int::fail(nullptr, FWA_ERR_INVALID_ARG, what " belongs to a context that has been destroyed"); \
    } while (0)

// ---- ctx_streams.cpp ----
int32_t overlapping_stream(fwa_ctx *ctx, const std::vector<hipStream_t> &peers, hipStream_t *out);
int32_t chain_streams(fwa_ctx *ctx, size_t n);  // the context's chain streams, created (and checked) on demand

// ---- tables.cpp ----
struct Pipeline {
    v2f *ring = nullptr;
    uint64_t ring_bytes = 0;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    hipEvent_t fork = nullptr;
};
int32_t upload_half_table(fwa_ctx *ctx, uint32_t n, v2f **d);
int32_t build_tables(fwa_ctx *ctx, uint32_t n, int64_t path, const uint32_t lf[3], Tables *t);
void *pool_take(fwa_ctx *ctx, uint64_t bytes);
void destroy_pipeline_objects(fwa_ctx *ctx, Pipeline &pl, bool pool_ring);
Pipeline take_pipeline(fwa_plan *p);
int32_t build_pipeline(fwa_plan *p, int64_t group, int64_t n_streams);

// ---- plan.cpp ----
int64_t choose_path(uint32_t n, uint64_t batch, uint32_t lf[3], bool *colsw = nullptr);
int32_t setup_path(fwa_plan *p);
size_t ctl_bytes(const fwa_plan *p);
uint32_t tiled_swizzle_default(const fwa_plan *p);

// ---- accessors comm.cpp was written against ----
inline int32_t use_device(fwa_ctx *ctx)
{
    USE_DEVICE(ctx);
    return FWA_OK;
}
inline int ctx_device(const fwa_ctx *ctx) { return ctx->device; }
inline fwa_ctx *buf_ctx(const fwa_buf *b) { return b->ctx; }
inline hipStream_t stream_raw(fwa_stream *s) { return s ? s->s : nullptr; }
inline fwa_ctx *stream_ctx(fwa_stream *s) { return s ? s->ctx : nullptr; }

}  // namespace fwa_int
