// buffers.cpp -- device and pinned host memory of the C ABI: allocation, uploads, downloads, copies.
//
// Replaces wgpu's create_buffer / write_buffer / copy_buffer_to_buffer / map_async of the reference's callers
// (src/examples/basic.rs:50-64,73,92-122); copies between two contexts' devices are explicit peer copies.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "internal.h"

using namespace fwa_int;

extern "C" {

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
    {
        std::lock_guard<std::mutex> lk(ctx->live_mu);
        ctx->live_bufs.insert(b);
    }
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
    {
        std::lock_guard<std::mutex> lk(ctx->live_mu);
        ctx->live_bufs.insert(b);
    }
    *out = b;
    return FWA_OK;
}

int32_t fwa_buf_free(fwa_buf *buf)
{
    if (!buf) return FWA_OK;
    if (buf->ctx) {
        std::lock_guard<std::mutex> lk(buf->ctx->live_mu);
        buf->ctx->live_bufs.erase(buf);
    }
    if (buf->owned && buf->p) { (void)hipSetDevice(buf->device); (void)hipFree(buf->p); }
    delete buf;
    return FWA_OK;
}

int32_t fwa_buf_upload(fwa_buf *dst, uint64_t dst_offset, const void *host, uint64_t bytes, fwa_stream *stream)
{
    if (!dst || (!host && bytes)) return fail(dst ? dst->ctx : nullptr, FWA_ERR_INVALID_ARG, "dst/host is NULL");
    LIVE_HANDLE(dst, "the buffer");
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
    LIVE_HANDLE(src, "the buffer");
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
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            return fail_hip(c, e, "hipDeviceEnablePeerAccess");
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
    LIVE_HANDLE(dst, "the destination buffer");
    LIVE_HANDLE(src, "the source buffer");
    if (dst_offset > dst->bytes || bytes > dst->bytes - dst_offset || src_offset > src->bytes ||
        bytes > src->bytes - src_offset)
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "copy range exceeds buffer");
    if (stream && !stream->ctx) return fail(dst->ctx, FWA_ERR_INVALID_ARG, "the stream's context has been destroyed");
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
    // two devices (one process driving several contexts: SURVEY.md 8(e)): an explicit peer copy, or a status code --
    // never a pointer the current device cannot reach handed to a plain device-to-device copy
    int32_t kind = 0;
    int32_t st = peer_kind(dst->ctx, const_cast<fwa_ctx *>(src->ctx), &kind);
    if (st) return st;
    if (kind != 2)
        return fail(dst->ctx, FWA_ERR_UNSUPPORTED,
                    "devices " + std::to_string(src->ctx->device) + " and " + std::to_string(dst->ctx->device) +
                        " have no peer access: stage through the host (fwa_buf_download / fwa_buf_upload) or move the "
                        "slab with fwa_comm_*");
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
    LIVE_HANDLE(src, "the buffer");
    if (src_offset > src->bytes || bytes > src->bytes - src_offset)
        return fail(src->ctx, FWA_ERR_INVALID_ARG, "download range exceeds buffer");
    if (!bytes) return FWA_OK;
    USE_DEVICE(src->ctx);
    HIP_TRY(src->ctx, hipMemcpyAsync(host, static_cast<const char *>(src->p) + src_offset, bytes,
                                     hipMemcpyDeviceToHost, raw(stream)));
    return FWA_OK;
}

void *fwa_buf_device_ptr(const fwa_buf *buf) { return buf ? buf->p : nullptr; }
uint64_t fwa_buf_size(const fwa_buf *buf) { return buf ? buf->bytes : 0; }

// ---- synthetic data / calibration -------------------------------------------
int32_t fwa_fill_synthetic(fwa_buf *dst, uint64_t seed, uint64_t first_transform, uint32_t fft_len, float scale,
                           fwa_stream *stream)
{
    if (!dst || !fft_len) return fail(dst ? dst->ctx : nullptr, FWA_ERR_INVALID_ARG, "dst NULL or fft_len 0");
    LIVE_HANDLE(dst, "the buffer");
    USE_DEVICE(dst->ctx);
    hipError_t e = fwa::launch_fill(static_cast<v2f *>(dst->p), seed, first_transform * (uint64_t)fft_len,
                                    dst->bytes / 8, scale, raw(stream));
    if (e != hipSuccess) return fail_hip(dst->ctx, e, "fill launch", FWA_ERR_LAUNCH);
    return FWA_OK;
}

int32_t fwa_calib_copy(fwa_buf *dst, const fwa_buf *src, uint64_t bytes, fwa_stream *stream)
{
    if (!dst || !src) return fail(nullptr, FWA_ERR_INVALID_ARG, "dst/src is NULL");
    LIVE_HANDLE(dst, "the destination buffer");
    LIVE_HANDLE(src, "the source buffer");
    if (bytes > dst->bytes || bytes > src->bytes || (bytes & 15))
        return fail(dst->ctx, FWA_ERR_INVALID_ARG, "copy size exceeds a buffer or is not a multiple of 16");
    USE_DEVICE(dst->ctx);
    // dst == src: an in-place streaming pass (every line read, then written back) -- the normalize kernel with scale
    // 1: 64-KiB chunk per workgroup, every wave walks 16 KiB with 32 nt loads in flight: the fastest streaming shape
    // on this part
    hipError_t e = (dst->p == src->p) ? fwa::launch_scale(static_cast<const v2f *>(src->p), static_cast<v2f *>(dst->p),
                                                          bytes / 8, 1.0f, raw(stream))
                                      : fwa::launch_copy(src->p, dst->p, bytes, raw(stream));
    if (e != hipSuccess) return fail_hip(dst->ctx, e, "copy launch", FWA_ERR_LAUNCH);
    return FWA_OK;
}

}  // extern "C"
