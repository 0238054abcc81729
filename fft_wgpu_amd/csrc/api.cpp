// api.cpp -- host side of the C ABI declared in include/fft_wgpu_amd.h.
//
// Mirrors the plan objects of reference src/processor.rs (Forward :7-159,
// Inverse :231-341, Normalize :409-505, Onlyinverse :566-670) without any of
// its wgpu plumbing: a plan owns its twiddle tables and scratch, exec only
// enqueues kernels on the caller's stream and allocates nothing.
#include "../../include/fft_wgpu_amd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"

using fwa::v2f;

struct fwa_ctx {
    int device = -1;
    hipDeviceProp_t prop{};
    mutable std::string err;
    bool setup_1m_done = false;
    bool setup_small_done = false;
};
struct fwa_stream {
    fwa_ctx *ctx = nullptr;
    hipStream_t s = nullptr;
    bool owned = false;
};
struct fwa_buf {
    fwa_ctx *ctx = nullptr;
    void *p = nullptr;
    uint64_t bytes = 0;
    bool owned = false;
};
struct fwa_event {
    fwa_ctx *ctx = nullptr;
    hipEvent_t e = nullptr;
};

enum fwa_path : int64_t {
    PATH_LDS_SMALL = 0,
    PATH_TWOPASS_1M = 1,
    PATH_R2_GLOBAL = 2,
    PATH_NORMALIZE = 3,
    PATH_IDENTITY = 4,
    PATH_FUSED_1M = 5,  // in-place persistent pipeline, one launch per exec
    PATH_TILED = 7,     // n = N1*N2[*N3], each 64..1024: 2-3 k_tile16 passes (default for 2^15..2^19, 2^21..2^30)
    PATH_SPLIT = 6,     // n = R1*R2*M: strided register-radix passes + fast sub-transforms (M = 2^20 or 4096) + permute
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
    // tables (device)
    v2f *tw_half = nullptr;        // n/2 entries, processor.rs:43-49
    v2f *tw_inner = nullptr;       // 2^20 path: [k1][n'] = W_1024^{n' k1}
    v2f *tw_outer = nullptr;       // 2^20 path: per tile A[32][16], B[32][16]
    // tiled path (PATH_TILED): log2 of the factors (lf[2] = 0 for two factors), per-factor W_L tables,
    // four-step tables of pass A (domain n) and pass B (domain N2*N3)
    uint32_t lf[3] = {0, 0, 0};
    v2f *tw_l[3] = {nullptr, nullptr, nullptr};
    v2f *tw_lo_b = nullptr, *tw_hi_b = nullptr;
    // split path (PATH_SPLIT)
    uint32_t r1 = 1, r2 = 1, leaf = 0;  // n = r1 * r2 * leaf
    v2f *tw_lo1 = nullptr, *tw_hi1 = nullptr, *tw_lo2 = nullptr, *tw_hi2 = nullptr;
    uint64_t leaf_batch = 0;            // sub-transforms the 2^20 pipeline / LDS kernel runs per exec
    // 2^20 pipeline
    v2f *ring = nullptr;
    int64_t group = 16;            // transforms per launch pair (measured best with unmixed launches, two chains)
    int64_t n_streams = 2;         // internal streams the groups alternate over
    uint64_t ring_slots = 0;
    uint64_t slot_bytes = 0;
    // fused 2^20 pipeline
    uint32_t *fused_ctl = nullptr;
    int64_t depth = 4;             // pass-2 tiles of transform t-depth interleave with pass-1 tiles of t
    int64_t wgs = 0;               // persistent workgroups (0 = 2 per CU)
    int64_t small_reg = 1;         // n <= 16384: 1 = register kernels, 0 = LDS radix-2 kernel, 2 = register + wave shuffles
    int64_t mix = 0;               // 1: each launch carries pass-1 tiles of group g and pass-2 tiles of group g-1
                                   // (same-run A/B, profiles/round1/f_mixed_vs_unmixed.txt: unmixed 16x2 21.4 ms, mixed 8x2 22.6 ms)
    int64_t policy = 1;            // cache-policy variant of the 2^20 kernels (kernels_1m.hip)
    int64_t dbg = 0;               // timing-only ablation switches of k_fused_1m (results wrong when != 0)
    std::vector<hipStream_t> istreams;
    std::vector<hipEvent_t> idone;
    hipEvent_t ev_fork = nullptr;
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
    std::string m = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    if (e == hipErrorOutOfMemory) st = FWA_ERR_OUT_OF_MEMORY;
    return fail(ctx, st, m);
}

#define HIP_TRY(ctx, call)                                       \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #call); \
    } while (0)

bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }
uint32_t ilog2(uint32_t n)
{
    uint32_t l = 0;
    while ((1u << l) < n) ++l;
    return l;
}

// Path and per-pass FFT lengths (log2) for a transform length; shared by fwa_plan_create and fwa_describe_path.
int64_t choose_path(uint32_t n, uint32_t lf[3])
{
    lf[0] = lf[1] = lf[2] = 0;
    const uint32_t lg = ilog2(n);
    if (n == 1) return PATH_IDENTITY;
    if (n <= 16384) { lf[0] = lg; return PATH_LDS_SMALL; }
    if (n == (1u << 20)) { lf[0] = lf[1] = 10; return PATH_TWOPASS_1M; }
    if (n <= (1u << 30)) {
        // factors of 64..1024 each.  Tiles of 512/1024-point FFTs leave one or two workgroups per CU and run
        // slower per pass than three passes of <= 256-point tiles (measured: 2^18 as 512x512 3.0 ms vs
        // 64x64x64 2.3 ms per 2^28 samples), so two factors only while both stay <= 512
        const uint32_t nf = lg <= 17 ? 2 : 3;
        for (uint32_t i = 0; i < nf; ++i) lf[i] = lg / nf + (i >= nf - lg % nf ? 1 : 0);
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

hipStream_t raw(fwa_stream *s) { return s ? s->s : nullptr; }

void release_pipeline(fwa_plan *p)
{
    for (auto s : p->istreams) (void)hipStreamDestroy(s);
    for (auto e : p->idone) (void)hipEventDestroy(e);
    p->istreams.clear();
    p->idone.clear();
    if (p->ev_fork) { (void)hipEventDestroy(p->ev_fork); p->ev_fork = nullptr; }
    if (p->ring) { (void)hipFree(p->ring); p->ring = nullptr; }
    if (p->fused_ctl) { (void)hipFree(p->fused_ctl); p->fused_ctl = nullptr; }
}

// Allocate the scratch ring and the internal streams of the 2^20 pipeline (first exec or plan creation).
int32_t build_pipeline(fwa_plan *p)
{
    fwa_ctx *ctx = p->ctx;
    release_pipeline(p);
    const uint64_t pbatch = p->leaf_batch;
    if (p->path == PATH_FUSED_1M) {
        if (pbatch == 0) return FWA_OK;
        if (p->wgs <= 0) p->wgs = 2 * (int64_t)ctx->prop.multiProcessorCount;
        if (p->wgs < 64) p->wgs = 64;  // progress guarantee of k_fused_1m needs >= 64 resident workgroups
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&p->fused_ctl), fwa::fused_ctl_bytes(pbatch)));
        return FWA_OK;
    }
    if (p->group < 1) p->group = 1;
    if ((uint64_t)p->group > pbatch && pbatch) p->group = (int64_t)pbatch;
    const uint64_t n_groups = pbatch ? (pbatch + p->group - 1) / p->group : 0;
    if (p->n_streams < 1) p->n_streams = 1;
    if ((uint64_t)p->n_streams > n_groups && n_groups) p->n_streams = (int64_t)n_groups;
    p->ring_slots = (uint64_t)p->group * (uint64_t)p->n_streams * ((p->mix && p->path != PATH_TILED) ? 2 : 1);
    if (p->ring_slots == 0) return FWA_OK;
    // one slot = one (sub-)transform of the pipeline: 2^20 samples on the two-pass paths, n on the tiled path
    p->slot_bytes = (p->path == PATH_TILED) ? (uint64_t)p->n * sizeof(v2f) : (sizeof(v2f) << 20);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&p->ring), p->ring_slots * p->slot_bytes));
    if (p->n_streams > 1) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
        for (int64_t i = 0; i < p->n_streams; ++i) {
            hipStream_t s;
            hipEvent_t e;
            HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            p->istreams.push_back(s);
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            p->idone.push_back(e);
        }
    }
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
    delete ctx;
    return FWA_OK;
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    fwa_stream *st = new (std::nothrow) fwa_stream;
    if (!st) { (void)hipStreamDestroy(s); return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    st->ctx = ctx; st->s = s; st->owned = true;
    *out = st;
    return FWA_OK;
}

int32_t fwa_stream_wrap(fwa_ctx *ctx, void *hip_stream, fwa_stream **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    fwa_stream *st = new (std::nothrow) fwa_stream;
    if (!st) return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    st->ctx = ctx; st->s = reinterpret_cast<hipStream_t>(hip_stream); st->owned = false;
    *out = st;
    return FWA_OK;
}

int32_t fwa_stream_synchronize(fwa_stream *stream)
{
    if (!stream) return fail(nullptr, FWA_ERR_INVALID_ARG, "stream is NULL");
    HIP_TRY(stream->ctx, hipStreamSynchronize(stream->s));
    return FWA_OK;
}

int32_t fwa_stream_destroy(fwa_stream *stream)
{
    if (!stream) return FWA_OK;
    if (stream->owned) (void)hipStreamDestroy(stream->s);
    delete stream;
    return FWA_OK;
}

// ---- buffers -------------------------------------------------------------
int32_t fwa_buf_alloc(fwa_ctx *ctx, uint64_t bytes, fwa_buf **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    if (bytes) {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail_hip(ctx, e, "hipMalloc");
    }
    fwa_buf *b = new (std::nothrow) fwa_buf;
    if (!b) { (void)hipFree(p); return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    b->ctx = ctx; b->p = p; b->bytes = bytes; b->owned = true;
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
    b->ctx = ctx; b->p = device_ptr; b->bytes = bytes; b->owned = false;
    *out = b;
    return FWA_OK;
}

int32_t fwa_buf_free(fwa_buf *buf)
{
    if (!buf) return FWA_OK;
    if (buf->owned && buf->p) (void)hipFree(buf->p);
    delete buf;
    return FWA_OK;
}

int32_t fwa_buf_upload(fwa_buf *dst, uint64_t dst_offset, const void *host, uint64_t bytes, fwa_stream *stream)
{
    if (!dst || (!host && bytes)) return fail(dst ? dst->ctx : nullptr, FWA_ERR_INVALID_ARG, "dst/host is NULL");
    if (dst_offset > dst->bytes || bytes > dst->bytes - dst_offset)
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "upload range exceeds buffer");
    if (!bytes) return FWA_OK;
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
    HIP_TRY(src->ctx, hipMemcpyAsync(host, static_cast<const char *>(src->p) + src_offset, bytes,
                                     hipMemcpyDeviceToHost, raw(stream)));
    // map_async + poll(wait) in the reference (examples/basic.rs:105-106): the data is on the host on return
    HIP_TRY(src->ctx, hipStreamSynchronize(raw(stream)));
    return FWA_OK;
}

int32_t fwa_buf_copy(fwa_buf *dst, uint64_t dst_offset, const fwa_buf *src, uint64_t src_offset, uint64_t bytes,
                     fwa_stream *stream)
{
    if (!dst || !src) return fail(nullptr, FWA_ERR_INVALID_ARG, "dst/src is NULL");
    if (dst_offset > dst->bytes || bytes > dst->bytes - dst_offset || src_offset > src->bytes ||
        bytes > src->bytes - src_offset)
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "copy range exceeds buffer");
    if (!bytes) return FWA_OK;
    HIP_TRY(dst->ctx, hipMemcpyAsync(static_cast<char *>(dst->p) + dst_offset,
                                     static_cast<const char *>(src->p) + src_offset, bytes, hipMemcpyDeviceToDevice,
                                     raw(stream)));
    return FWA_OK;
}

int32_t fwa_host_alloc(fwa_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    if (!bytes) return FWA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(src->ctx, hipMemcpyAsync(host, static_cast<const char *>(src->p) + src_offset, bytes,
                                     hipMemcpyDeviceToHost, raw(stream)));
    return FWA_OK;
}

int32_t fwa_stream_wait_stream(fwa_stream *stream, fwa_stream *other)
{
    if (!stream || !other) return fail(nullptr, FWA_ERR_INVALID_ARG, "stream is NULL");
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
    release_pipeline(plan);
    if (plan->tw_half) (void)hipFree(plan->tw_half);
    if (plan->tw_inner) (void)hipFree(plan->tw_inner);
    if (plan->tw_outer) (void)hipFree(plan->tw_outer);
    for (v2f *t : {plan->tw_lo1, plan->tw_hi1, plan->tw_lo2, plan->tw_hi2, plan->tw_l[0], plan->tw_l[1], plan->tw_l[2],
                   plan->tw_lo_b, plan->tw_hi_b})
        if (t) (void)hipFree(t);
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

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fwa_plan *p = new (std::nothrow) fwa_plan;
    if (!p) return fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    p->ctx = ctx; p->kind = kind; p->n = fft_len; p->lg = ilog2(fft_len);
    p->batch = src->bytes / tbytes;
    p->src = src; p->second = src2_or_null;

    int32_t st = FWA_OK;
    auto bail = [&](int32_t s) { fwa_plan_destroy(p); return s; };

    if (kind == FWA_NORMALIZE) {
        p->path = PATH_NORMALIZE;
        *out = p;
        return FWA_OK;
    }

    p->path = choose_path(fft_len, p->lf);  // PATH_FUSED_1M is opt-in (experimental)
    if (p->path == PATH_TILED && std::getenv("FWA_FORCE_SPLIT")) p->path = PATH_SPLIT;  // cross-check path
    p->leaf_batch = p->batch;
    if (p->path == PATH_SPLIT) {
        p->leaf = (fft_len > (1u << 20)) ? (1u << 20) : 4096u;
        const uint32_t rt = fft_len / p->leaf;
        p->r1 = rt <= 32 ? rt : 32;
        p->r2 = rt / p->r1;
        p->leaf_batch = p->batch * rt;
    }

    // Forward/Inverse own their ping-pong partner (processor.rs:34-41,261-269).  It is only
    // materialised when the result must land there (odd log2 n) or the path ping-pongs.
    const bool odd = (p->lg & 1) != 0;
    const bool need_second = odd || p->path == PATH_R2_GLOBAL || p->path == PATH_SPLIT;
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

    if (p->path == PATH_LDS_SMALL && fft_len > 4096 && !ctx->setup_small_done) {
        hipError_t e = fwa::setup_small_kernels();
        if (e != hipSuccess) return bail(fail_hip(ctx, e, "hipFuncSetAttribute(max dynamic LDS)"));
        ctx->setup_small_done = true;
    }
    auto level = [&](uint64_t cur, v2f **lo, v2f **hi) -> int32_t {
        const uint64_t nlo = cur < 1024 ? cur : 1024, nhi = cur < 1024 ? 1 : cur / 1024;
        std::vector<v2f> l(nlo), h(nhi);
        for (uint64_t j = 0; j < nlo; ++j) l[j] = tw_f64(j, cur);
        for (uint64_t j = 0; j < nhi; ++j) h[j] = tw_f64(1024 * j, cur);
        int32_t s = upload_table(ctx, l, lo);
        return s ? s : upload_table(ctx, h, hi);
    };
    if (p->path == PATH_TILED) {
        const uint32_t nf = p->lf[2] ? 3 : 2;
        for (uint32_t i = 0; i < nf; ++i) {
            const uint32_t L = 1u << p->lf[i];
            std::vector<v2f> h(L / 2);
            for (uint32_t k = 0; k < L / 2; ++k) h[k] = tw_f64(k, L);
            st = upload_table(ctx, h, &p->tw_l[i]);
            if (st) return bail(st);
            hipError_t pe = fwa::prepare_tile16(p->lf[i]);
            if (pe != hipSuccess) return bail(fail_hip(ctx, pe, "hipFuncSetAttribute(max dynamic LDS)"));
        }
        st = level(fft_len, &p->tw_lo1, &p->tw_hi1);
        if (st) return bail(st);
        if (nf == 3) {
            st = level((uint64_t)fft_len >> p->lf[0], &p->tw_lo_b, &p->tw_hi_b);
            if (st) return bail(st);
        }
        // intermediate of a group of transforms lives in a ring slab of 128 MiB per chain (two chains = the
        // 256-MiB Infinity Cache; group sweep in profiles/round1/h_tiled_group_sweep.jsonl)
        const uint64_t per = (uint64_t)fft_len * sizeof(v2f);
        p->group = (int64_t)((128ull << 20) / per);
        if (p->group < 1) p->group = 1;
        st = build_pipeline(p);
        if (st) return bail(st);
    }
    if (p->path == PATH_SPLIT) {
        st = level(fft_len, &p->tw_lo1, &p->tw_hi1);
        if (st) return bail(st);
        if (p->r2 > 1) {
            st = level(fft_len / p->r1, &p->tw_lo2, &p->tw_hi2);
            if (st) return bail(st);
        }
    }
    const uint32_t leaf_n = (p->path == PATH_SPLIT) ? p->leaf : fft_len;
    if (p->path == PATH_LDS_SMALL || p->path == PATH_R2_GLOBAL || (p->path == PATH_SPLIT && leaf_n == 4096)) {
        std::vector<v2f> h(leaf_n / 2);
        for (uint32_t k = 0; k < leaf_n / 2; ++k) h[k] = tw_f64(k, leaf_n);
        st = upload_table(ctx, h, &p->tw_half);
        if (st) return bail(st);
    } else if (p->path == PATH_TWOPASS_1M || p->path == PATH_FUSED_1M || p->path == PATH_SPLIT) {
        if (!ctx->setup_1m_done) {
            hipError_t e = fwa::setup_1m_kernels();
            if (e != hipSuccess) return bail(fail_hip(ctx, e, "hipFuncSetAttribute(max dynamic LDS)"));
            ctx->setup_1m_done = true;
        }
        std::vector<v2f> inner(1024), outer((size_t)64 * 1024);
        for (uint32_t k1 = 0; k1 < 32; ++k1)
            for (uint32_t q = 0; q < 32; ++q) inner[k1 * 32 + q] = tw_f64((uint64_t)k1 * q, 1024);
        const uint64_t N = 1ull << 20;
        for (uint32_t tile = 0; tile < 64; ++tile)
            for (uint32_t k = 0; k < 32; ++k)
                for (uint32_t c = 0; c < 16; ++c) {
                    const uint64_t n2 = 16 * tile + c;
                    outer[(size_t)tile * 1024 + k * 16 + c] = tw_f64(n2 * k, N);             // A[k1][c]
                    outer[(size_t)tile * 1024 + 512 + k * 16 + c] = tw_f64(32 * n2 * k, N);  // B[k2][c]
                }
        st = upload_table(ctx, inner, &p->tw_inner);
        if (st) return bail(st);
        st = upload_table(ctx, outer, &p->tw_outer);
        if (st) return bail(st);
        st = build_pipeline(p);
        if (st) return bail(st);
    }
    *out = p;
    return FWA_OK;
}

static fwa_buf *result_buffer(fwa_plan *p)
{
    // processor.rs:153-157, :335-339, :664-668
    return (p->lg % 2 == 0) ? p->src : p->second;
}

// The 2^20 two-pass pipeline on `nb` transforms starting at `a` (results to `out`, may alias `a`).
static int32_t exec_twopass(fwa_plan *plan, int dir, v2f *a, v2f *out, uint64_t nb, float scale, hipStream_t st)
{
    fwa_ctx *ctx = plan->ctx;
    hipError_t e = hipSuccess;
    const uint64_t G = (uint64_t)plan->group;
    const uint64_t n_groups = (nb + G - 1) / G;
    const size_t ns = plan->istreams.size();
    const size_t chains = ns ? ns : 1;
    const uint64_t N = 1ull << 20;
    if (ns) {
        HIP_TRY(ctx, hipEventRecord(plan->ev_fork, st));
        for (size_t i = 0; i < ns; ++i) HIP_TRY(ctx, hipStreamWaitEvent(plan->istreams[i], plan->ev_fork, 0));
    }
    auto count = [&](uint64_t g) { return (uint32_t)((nb - g * G < G) ? nb - g * G : G); };
    if (plan->mix) {
        // chain c owns groups c, c+chains, ...; launch i of a chain = pass 1 of its i-th group next to
        // pass 2 of its (i-1)-th group; two ring slabs per chain, used alternately.
        const uint64_t rounds = (n_groups + chains - 1) / chains;
        for (uint64_t i = 0; i <= rounds && e == hipSuccess; ++i) {
            for (size_t c = 0; c < chains && e == hipSuccess; ++c) {
                const uint64_t g = i * chains + c, gp = g - chains;  // gp valid when i > 0
                const bool has1 = (i < rounds) && g < n_groups;
                const bool has2 = (i > 0) && gp < n_groups;
                if (!has1 && !has2) continue;
                v2f *slab_w = plan->ring + ((uint64_t)c * 2 + (i & 1)) * G * N;
                v2f *slab_r = plan->ring + ((uint64_t)c * 2 + ((i + 1) & 1)) * G * N;
                e = fwa::launch_mix_1m(dir, (int)plan->policy, has1 ? a + g * G * N : a, slab_w, has1 ? count(g) : 0,
                                       slab_r, has2 ? out + gp * G * N : out, has2 ? count(gp) : 0, plan->tw_inner,
                                       plan->tw_outer, scale, (uint32_t)plan->dbg, ns ? plan->istreams[c] : st);
            }
        }
    } else {
        for (uint64_t g = 0; g < n_groups && e == hipSuccess; ++g) {
            const uint64_t t0 = g * G;
            const uint32_t cnt = count(g);
            const size_t si = ns ? (size_t)(g % ns) : 0;
            hipStream_t s = ns ? plan->istreams[si] : st;
            // ring region of this stream: G slots; inside a group transform t uses slot (t - t0)
            v2f *ring = plan->ring + (uint64_t)si * G * N;
            e = fwa::launch_p1_1m(dir, (int)plan->policy, a + t0 * N, ring, plan->tw_inner, plan->tw_outer, (uint32_t)G,
                                  0, cnt, s);
            if (e != hipSuccess) break;
            e = fwa::launch_p2_1m(dir, (int)plan->policy, ring, out + t0 * N, plan->tw_inner, (uint32_t)G, 0, cnt, scale,
                                  s);
        }
    }
    if (e != hipSuccess) return fail_hip(ctx, e, "kernel launch", FWA_ERR_LAUNCH);
    if (ns) {
        for (size_t i = 0; i < ns; ++i) {
            HIP_TRY(ctx, hipEventRecord(plan->idone[i], plan->istreams[i]));
            HIP_TRY(ctx, hipStreamWaitEvent(st, plan->idone[i], 0));
        }
    }
    return FWA_OK;
}

int32_t fwa_plan_exec(fwa_plan *plan, fwa_stream *stream, fwa_buf **result)
{
    if (!plan) return fail(nullptr, FWA_ERR_INVALID_ARG, "plan is NULL");
    fwa_ctx *ctx = plan->ctx;
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

    switch (plan->path) {
        case PATH_IDENTITY:
            if (scale != 1.0f) e = fwa::launch_scale(a, a, total, scale, st);
            break;
        case PATH_LDS_SMALL:
            if (plan->small_reg && plan->n < 16)
                e = fwa::launch_tiny(dir, a, out, plan->n, plan->batch, scale, st);
            else if (plan->small_reg)
                e = fwa::launch_small16(dir, a, out, plan->tw_half, plan->n, plan->batch, scale, plan->small_reg == 2, st);
            else
                e = fwa::launch_lds_small(dir, a, out, plan->tw_half, plan->n, plan->batch, scale, st);
            break;
        case PATH_R2_GLOBAL:
            for (uint32_t s = 0; s < plan->lg && e == hipSuccess; ++s) {
                const v2f *from = (s % 2 == 0) ? a : b;
                v2f *to = (s % 2 == 0) ? b : a;
                e = fwa::launch_r2_stage(dir, from, to, plan->tw_half, plan->n, s, plan->batch,
                                         (s + 1 == plan->lg) ? scale : 1.0f, st);
            }
            break;
        case PATH_FUSED_1M:
            // in place: 2^20 has even log2, the result buffer is src (processor.rs:153-157)
            e = fwa::launch_fused_1m(dir, (int)plan->policy, a, plan->tw_inner, plan->tw_outer, plan->fused_ctl, (uint32_t)plan->batch,
                                     (uint32_t)plan->depth, (uint32_t)plan->wgs, scale, (uint32_t)plan->dbg, st);
            break;
        case PATH_TWOPASS_1M: {
            const int32_t s2 = exec_twopass(plan, dir, a, out, plan->batch, scale, st);
            if (s2) return s2;
            break;
        }
        case PATH_TILED: {
            // n = N1*N2[*N3]; index n = (n1*N2 + n2)*N3 + n3, k = k1 + N1*(k2 + N2*k3).  Per group of transforms:
            // pass A: FFT over n1 (cols, twiddle W_n), user buffer -> ring slab; [pass B: FFT over n2 per k1 (cols,
            // twiddle W_{N2*N3}), in place in the slab]; pass C: FFT over the contiguous axis with the transposed
            // store, slab -> result buffer (src for even log2 n -- in place at group granularity -- else second).
            const bool three = plan->lf[2] != 0;
            const uint64_t N = plan->n, N1 = 1ull << plan->lf[0], N2 = 1ull << plan->lf[1],
                           N3 = three ? (1ull << plan->lf[2]) : 1;
            const uint64_t G = (uint64_t)plan->group, n_groups = (plan->batch + G - 1) / G;
            const size_t ns = plan->istreams.size();
            if (ns) {
                HIP_TRY(ctx, hipEventRecord(plan->ev_fork, st));
                for (size_t i = 0; i < ns; ++i) HIP_TRY(ctx, hipStreamWaitEvent(plan->istreams[i], plan->ev_fork, 0));
            }
            for (uint64_t g = 0; g < n_groups && e == hipSuccess; ++g) {
                const uint64_t cnt = (plan->batch - g * G < G) ? plan->batch - g * G : G;
                const size_t c = ns ? (size_t)(g % ns) : 0;
                hipStream_t s = ns ? plan->istreams[c] : st;
                v2f *slab = plan->ring + (uint64_t)c * G * N;
                fwa::TileArgs ta{};
                ta.scale = 1.0f;
                ta.flags = (uint32_t)plan->dbg;
                // pass A
                ta.in = a + g * G * N; ta.out = slab; ta.tw = plan->tw_l[0]; ta.tw_lo = plan->tw_lo1; ta.tw_hi = plan->tw_hi1;
                ta.in_sb = ta.out_sb = N; ta.in_s1 = ta.out_s1 = 0; ta.in_st = ta.out_st = 16;
                ta.pitch = N / N1; ta.out_stride = 0; ta.d1_count = 1; ta.tile_count = (uint32_t)(N / N1 / 16);
                ta.flags = (uint32_t)plan->dbg | (plan->policy ? (1u << 8) : 0);  // first pass: user buffer -> ring
                e = fwa::launch_tile16(dir, fwa::TILE_COLS, plan->lf[0], ta, cnt, s);
                if (e != hipSuccess) break;
                if (three) {  // pass B, in place in the slab
                    ta.in = slab; ta.out = slab; ta.tw = plan->tw_l[1]; ta.tw_lo = plan->tw_lo_b; ta.tw_hi = plan->tw_hi_b;
                    ta.in_s1 = ta.out_s1 = N2 * N3; ta.pitch = N3; ta.d1_count = (uint32_t)N1;
                    ta.tile_count = (uint32_t)(N3 / 16);
                    ta.flags = (uint32_t)plan->dbg | (plan->policy ? (2u << 8) : 0);  // middle pass: ring -> ring
                    e = fwa::launch_tile16(dir, fwa::TILE_COLS, plan->lf[1], ta, cnt, s);
                    if (e != hipSuccess) break;
                }
                // pass C: rows of the last axis, 16 adjacent k1 per tile
                const uint32_t li = three ? 2 : 1;
                ta.in = slab; ta.out = out + g * G * N; ta.tw = plan->tw_l[li]; ta.tw_lo = nullptr; ta.tw_hi = nullptr;
                ta.scale = scale;
                ta.in_sb = ta.out_sb = N;
                ta.pitch = N / N1;  // distance between the rows k1 and k1+1
                ta.in_st = 16 * (N / N1); ta.out_st = 16; ta.tile_count = (uint32_t)(N1 / 16);
                if (three) { ta.d1_count = (uint32_t)N2; ta.in_s1 = N3; ta.out_s1 = N1; ta.out_stride = N1 * N2; }
                else { ta.d1_count = 1; ta.in_s1 = ta.out_s1 = 0; ta.out_stride = N1; }
                ta.flags = (uint32_t)plan->dbg | (plan->policy ? (3u << 8) : 0);  // last pass: ring -> user buffer
                e = fwa::launch_tile16(dir, fwa::TILE_ROWS_T, plan->lf[li], ta, cnt, s);
            }
            if (e != hipSuccess) break;
            if (ns) {
                for (size_t i = 0; i < ns; ++i) {
                    HIP_TRY(ctx, hipEventRecord(plan->idone[i], plan->istreams[i]));
                    HIP_TRY(ctx, hipStreamWaitEvent(st, plan->idone[i], 0));
                }
            }
            break;
        }
        case PATH_SPLIT: {
            // even log2 n: src -> second (pass 1), work in second, permute back into src; odd: work in src,
            // permute into second -- the result lands where processor.rs:153-157 says.
            const uint32_t lg_m = ilog2(plan->leaf), lg_r1 = ilog2(plan->r1), lg_r2 = ilog2(plan->r2);
            v2f *work = (plan->lg % 2 == 0) ? b : a;
            e = fwa::launch_radix_pass(dir, (int)plan->r1, a, work, plan->tw_lo1, plan->tw_hi1, lg_m + lg_r2,
                                       plan->batch, st);
            if (e == hipSuccess && plan->r2 > 1)
                e = fwa::launch_radix_pass(dir, (int)plan->r2, work, work, plan->tw_lo2, plan->tw_hi2, lg_m,
                                           plan->batch * plan->r1, st);
            if (e != hipSuccess) break;
            if (plan->leaf == (1u << 20)) {
                const int32_t s2 = exec_twopass(plan, dir, work, work, plan->leaf_batch, 1.0f, st);
                if (s2) return s2;
            } else {
                e = plan->small_reg
                        ? fwa::launch_small16(dir, work, work, plan->tw_half, plan->leaf, plan->leaf_batch, 1.0f, false, st)
                        : fwa::launch_lds_small(dir, work, work, plan->tw_half, plan->leaf, plan->leaf_batch, 1.0f, st);
                if (e != hipSuccess) break;
            }
            e = fwa::launch_permute(work, out, lg_r1, lg_r2, lg_m, plan->batch, scale, st);
            break;
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
    *path = (int32_t)choose_path(fft_len, log2_factors);
    return FWA_OK;
}

int32_t fwa_plan_get_i64(const fwa_plan *plan, const char *key, int64_t *value)
{
    if (!plan || !key || !value) return fail(plan ? plan->ctx : nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    const std::string k(key);
    if (k == "batch") *value = (int64_t)plan->batch;
    else if (k == "fft_len") *value = plan->n;
    else if (k == "path") *value = plan->path;
    else if (k == "group") *value = plan->group;
    else if (k == "streams") *value = plan->n_streams;
    else if (k == "depth") *value = plan->depth;
    else if (k == "wgs") *value = plan->wgs;
    else if (k == "policy") *value = plan->policy;
    else if (k == "mix") *value = plan->mix;
    else if (k == "small_reg") *value = plan->small_reg;
    else if (k == "device_error") {
        // bounded-spin timeout flag of the fused kernel (0 in every healthy run); synchronises the device
        *value = 0;
        if (plan->fused_ctl) {
            uint32_t w = 0;
            HIP_TRY(plan->ctx, hipDeviceSynchronize());
            HIP_TRY(plan->ctx, hipMemcpy(&w, plan->fused_ctl + 1, sizeof(w), hipMemcpyDeviceToHost));
            *value = w;
        }
    }
    else if (k == "scratch_bytes")
        *value = (int64_t)(plan->ring_slots * plan->slot_bytes) +
                 (plan->second_owned ? (int64_t)plan->own_second.bytes : 0) +
                 (plan->fused_ctl ? (int64_t)fwa::fused_ctl_bytes(plan->batch) : 0);
    else if (k == "launches_per_exec") {
        switch (plan->path) {
            case PATH_TWOPASS_1M: {
                const int64_t ng = (int64_t)((plan->batch + plan->group - 1) / plan->group);
                const int64_t ch = plan->istreams.empty() ? 1 : (int64_t)plan->istreams.size();
                *value = plan->mix ? ng + (ng < ch ? ng : ch) : 2 * ng;
                break;
            }
            case PATH_FUSED_1M: *value = 1; break;
            case PATH_TILED:
                *value = (plan->lf[2] ? 3 : 2) * (int64_t)((plan->batch + plan->group - 1) / plan->group);
                break;
            case PATH_SPLIT: {
                int64_t leafl = 1;
                if (plan->leaf == (1u << 20)) {
                    const int64_t ng = (int64_t)((plan->leaf_batch + plan->group - 1) / plan->group);
                    const int64_t ch = plan->istreams.empty() ? 1 : (int64_t)plan->istreams.size();
                    leafl = plan->mix ? ng + (ng < ch ? ng : ch) : 2 * ng;
                }
                *value = 1 + (plan->r2 > 1 ? 1 : 0) + leafl + 1;
                break;
            }
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
    const std::string k(key);
    if (k == "group" || k == "streams") {
        if (plan->path != PATH_TWOPASS_1M && plan->path != PATH_TILED && !(plan->path == PATH_SPLIT && plan->leaf == (1u << 20)))
            return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "key only applies to the two-launch 2^20 pipeline");
        if (value < 1 || value > 4096) return fail(plan->ctx, FWA_ERR_INVALID_ARG, "value out of range");
        if (k == "group") plan->group = value; else plan->n_streams = value;
        return build_pipeline(plan);
    }
    if (k == "dbg") { plan->dbg = value; return FWA_OK; }
    if (k == "small_reg") {
        if (!value && plan->n > 4096) return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "the LDS radix-2 kernel stops at n = 4096");
        plan->small_reg = value == 2 ? 2 : (value ? 1 : 0);  // 2: wavefront-shuffle exchange at n = 32/64/128
        return FWA_OK;
    }
    if (k == "mix") {
        if (plan->path != PATH_TWOPASS_1M) return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "key only applies to the two-launch 2^20 path");
        plan->mix = value ? 1 : 0;
        return build_pipeline(plan);
    }
    if (k == "policy") {
        if (value < 0 || value > 7) return fail(plan->ctx, FWA_ERR_INVALID_ARG, "policy out of range");
        plan->policy = value;
        return FWA_OK;
    }
    if (k == "depth" || k == "wgs") {
        if (plan->path != PATH_FUSED_1M) return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "key only applies to the fused 2^20 path");
        if (value < 1 || value > 65536) return fail(plan->ctx, FWA_ERR_INVALID_ARG, "value out of range");
        if (k == "depth") plan->depth = value; else plan->wgs = value < 64 ? 64 : value;
        return FWA_OK;
    }
    if (k == "path") {
        if (plan->kind == FWA_NORMALIZE) return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "normalize has one path");
        if (value == PATH_FUSED_1M && plan->batch >= (1u << 24))
            return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "fused path needs batch < 2^24");
        if ((value == PATH_TWOPASS_1M || value == PATH_FUSED_1M) && plan->n == (1u << 20)) {
            plan->path = value;
            return build_pipeline(plan);
        }
        if (value == PATH_TILED || value == PATH_SPLIT)
            return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "tiled/split are chosen at plan creation (FWA_FORCE_SPLIT=1 selects split)");
        if (value == PATH_R2_GLOBAL && plan->n >= 2) {
            // force the literal reference recurrence (one launch per stage)
            if (!plan->tw_half) {
                std::vector<v2f> h(plan->n / 2);
                for (uint32_t i = 0; i < plan->n / 2; ++i) h[i] = tw_f64(i, plan->n);
                int32_t st = upload_table(plan->ctx, h, &plan->tw_half);
                if (st) return st;
            }
            if (!plan->second->p && plan->src->bytes) {
                if (plan->second != &plan->own_second)
                    return fail(plan->ctx, FWA_ERR_INVALID_ARG, "second buffer missing");
                hipError_t e = hipMalloc(&plan->own_second.p, plan->src->bytes);
                if (e != hipSuccess) return fail_hip(plan->ctx, e, "hipMalloc(second buffer)");
                plan->own_second.bytes = plan->src->bytes;
                plan->second_owned = true;
            }
            plan->path = PATH_R2_GLOBAL;
            return FWA_OK;
        }
        return fail(plan->ctx, FWA_ERR_UNSUPPORTED, "only path=2 (radix-2 global) or, at n=2^20, 1/5 can be forced");
    }
    return fail(plan->ctx, FWA_ERR_INVALID_ARG, "unknown key: " + k);
}

// ---- events ----------------------------------------------------------------
int32_t fwa_event_create(fwa_ctx *ctx, fwa_event **out)
{
    if (!ctx || !out) return fail(ctx, FWA_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
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
    HIP_TRY(ev->ctx, hipEventRecord(ev->e, raw(stream)));
    return FWA_OK;
}
int32_t fwa_event_elapsed_ms(fwa_event *start, fwa_event *end, float *ms)
{
    if (!start || !end || !ms) return fail(nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
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
    hipError_t e = fwa::launch_copy(src->p, dst->p, bytes, raw(stream));
    if (e != hipSuccess) return fail_hip(dst->ctx, e, "copy launch", FWA_ERR_LAUNCH);
    return FWA_OK;
}

}  // extern "C"
