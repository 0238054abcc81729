// ctx_streams.cpp -- contexts, error reporting, device enumeration, the slab rule, streams and events of the C ABI.
//
// Replaces the Instance / Adapter / Device / Queue plumbing of the reference (src/lib.rs:29-62) and the submission
// ordering its callers rely on (src/examples/basic.rs:73-122): a stream is a command encoder's queue.
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

thread_local std::string g_err;  // ctx-less failures

int32_t fail(const fwa_ctx *ctx, int32_t st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    else g_err = msg;
    return st;
}
int32_t fail_hip(const fwa_ctx *ctx, hipError_t e, const char *what, int32_t st)
{
    // HIP keeps the last error until somebody reads it; the launch wrappers read it after every launch, so an error
    // that has been reported here (e.g. an out-of-memory hipMalloc) must not surface again as a bogus launch failure
    (void)hipGetLastError();
    std::string m = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    if (e == hipErrorOutOfMemory) st = FWA_ERR_OUT_OF_MEMORY;
    return fail(ctx, st, m);
}
const char *thread_error_string() { return g_err.c_str(); }

// Streams that overlap.  Two HIP streams do not always run side by side on this stack: which hardware queue a new
// stream lands on depends on what the process created and destroyed before, and a pair that shares one runs strictly
// one after the other -- a pipelined plan whose two chains shared a queue took the single-chain time on every exec
// (+15 %: profiles/round3/probe_plan_instance_modes.txt), a host pipeline whose transfer streams shared one moved 21
// GB/s each way instead of 44.  So a stream created by this library is accepted only if a memory-free spin kernel on
// it overlaps the same kernel on its `peers` (time on all of them at once < single + half a spin); a rejected
// candidate stays alive until the search ends so that the runtime cannot hand the same queue back.  Best effort:
// after 6 rejections the last candidate is kept (a process with more streams than the runtime has hardware queues
// cannot overlap them all).
//
// Side effects, and how a caller controls them (include/fft_wgpu_amd.h, "Threading"): the check launches ~40-us spin
// kernels on the candidate, on the peers and on a private base stream of its own -- never on the null stream -- and
// runs only when there are peers to overlap with (a plan with one chain, or a context's first stream, costs nothing).
// It is refused with FWA_ERR_UNSUPPORTED while a stream of this context is capturing a graph (its timing would be
// meaningless and the peers may be the capturing streams), and fwa_ctx_set_i64(ctx, "chain_check", 0) turns it off:
// streams are then taken as the runtime hands them out.
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
                    "a stream of this context is capturing a graph: create plans and streams before the capture "
                    "begins, or turn the stream-overlap check off with fwa_ctx_set_i64(ctx, \"chain_check\", 0)");
    constexpr uint32_t TICKS = 4000, BLOCKS = 256;  // 40 us, one wave per CU
    hipEvent_t e0 = nullptr, e1 = nullptr, fork = nullptr;
    hipStream_t base = nullptr;  // the check's own fork / join stream: nothing here touches the null stream
    std::vector<hipEvent_t> done;
    std::vector<hipStream_t> rejected;
    hipError_t e = hipStreamCreateWithFlags(&base, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    // spin on every stream of `set`, forked from / joined to `base`
    auto timed = [&](const std::vector<hipStream_t> &set, float *us) {
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

}  // namespace fwa_int

extern "C" {

int32_t fwa_abi_version(void) { return FWA_ABI_VERSION; }

const char *fwa_last_error_string(const fwa_ctx *ctx) { return ctx ? ctx->err.c_str() : thread_error_string(); }

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

int32_t fwa_device_info(int32_t device_ordinal, char *name, size_t name_cap, int32_t *compute_units,
                        uint64_t *hbm_bytes,
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
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, FWA_ERR_INVALID_ARG, "device ordinal out of range");
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

// The slab rule of SURVEY.md 8(e): contiguous runs of whole transforms, sizes differing by at most one.  Pure host
// logic; fft_wgpu_amd/sharding.py::slab and fft_wgpu::slab (include/fft_wgpu.hpp) are this function.
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
                          std::string("device is ") + ctx->prop.gcnArchName +
                              ", this library is built for gfx950 only");
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
    // ... and so do buffer and event handles: freed memory is never dereferenced through them, using one afterwards is
    // FWA_ERR_INVALID_ARG (the device memory of an owned buffer is released by its own fwa_buf_free)
    for (fwa_buf *b : ctx->live_bufs) b->ctx = nullptr;
    for (fwa_event *ev : ctx->live_events) ev->ctx = nullptr;
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
    else if (k == "live_buffers") {
        std::lock_guard<std::mutex> lk(const_cast<fwa_ctx *>(ctx)->live_mu);
        *value = (int64_t)ctx->live_bufs.size();
    }
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
        // 0: streams this library creates (the chain streams of pipelined plans, fwa_stream_create) are no longer
        // tested for overlap with spin kernels (overlapping_stream above)
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
    std::vector<hipStream_t> peers(ctx->user_streams.end() - (std::ptrdiff_t)std::min<size_t>(2,
                                                                                              ctx->user_streams.size()),
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
        // a wrapper (fwa_stream_wrap) of a handle fwa_stream_create made must not take the still-alive owned stream
        // out of the overlap-check peer list
        auto &us = stream->ctx->user_streams;
        if (stream->owned) us.erase(std::remove(us.begin(), us.end(), stream->s), us.end());
    }
    if (stream->owned) {
        (void)hipSetDevice(stream->device);
        (void)hipStreamDestroy(stream->s);
    }
    delete stream;
    return FWA_OK;
}

int32_t fwa_stream_wait_stream(fwa_stream *stream, fwa_stream *other)
{
    if (!stream || !other || !stream->ctx)
        return fail(nullptr, FWA_ERR_INVALID_ARG, "stream is NULL or its context has been destroyed");
    USE_DEVICE(stream->ctx);
    hipEvent_t ev;
    HIP_TRY(stream->ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, other->s);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream->s, ev, 0);
    (void)hipEventDestroy(ev);  // destruction is deferred by the runtime until the event has completed
    if (e != hipSuccess) return fail_hip(stream->ctx, e, "hipEventRecord/hipStreamWaitEvent");
    return FWA_OK;
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
    {
        std::lock_guard<std::mutex> lk(ctx->live_mu);
        ctx->live_events.insert(ev);
    }
    *out = ev;
    return FWA_OK;
}
int32_t fwa_event_record(fwa_event *ev, fwa_stream *stream)
{
    if (!ev) return fail(nullptr, FWA_ERR_INVALID_ARG, "event is NULL");
    LIVE_HANDLE(ev, "the event");
    USE_DEVICE(ev->ctx);
    HIP_TRY(ev->ctx, hipEventRecord(ev->e, raw(stream)));
    return FWA_OK;
}
int32_t fwa_event_synchronize(fwa_event *ev)
{
    if (!ev) return fail(nullptr, FWA_ERR_INVALID_ARG, "event is NULL");
    LIVE_HANDLE(ev, "the event");
    USE_DEVICE(ev->ctx);
    HIP_TRY(ev->ctx, hipEventSynchronize(ev->e));
    return FWA_OK;
}
int32_t fwa_stream_wait_event(fwa_stream *stream, fwa_event *ev)
{
    if (!ev) return fail(nullptr, FWA_ERR_INVALID_ARG, "event is NULL");
    LIVE_HANDLE(ev, "the event");
    USE_DEVICE(ev->ctx);
    HIP_TRY(ev->ctx, hipStreamWaitEvent(raw(stream), ev->e, 0));
    return FWA_OK;
}
int32_t fwa_event_elapsed_ms(fwa_event *start, fwa_event *end, float *ms)
{
    if (!start || !end || !ms) return fail(nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    LIVE_HANDLE(start, "the event");
    LIVE_HANDLE(end, "the event");
    USE_DEVICE(end->ctx);
    HIP_TRY(end->ctx, hipEventSynchronize(end->e));
    HIP_TRY(end->ctx, hipEventElapsedTime(ms, start->e, end->e));
    return FWA_OK;
}
int32_t fwa_event_destroy(fwa_event *ev)
{
    if (!ev) return FWA_OK;
    if (ev->ctx) {
        std::lock_guard<std::mutex> lk(ev->ctx->live_mu);
        ev->ctx->live_events.erase(ev);
    }
    (void)hipEventDestroy(ev->e);
    delete ev;
    return FWA_OK;
}

}  // extern "C"
