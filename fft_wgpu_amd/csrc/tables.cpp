// tables.cpp -- twiddle tables (shared through the context's plan cache), the ring pool and the pipeline objects
// (ring slab, chain streams, fork / join events) of the multi-pass plans.
//
// The reference builds one n/2-entry table per plan from f64 math rounded to f32 (src/processor.rs:43-49); every table
// here follows that rule (tw_f64), factorised into two levels where a table of n entries would be too large.
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
        for (int wi = 0; wi < 1 && !st; ++wi) {   // one tile width ships: 16 columns
            const uint32_t W = 16, tiles = 1024 / W;
            std::vector<v2f> outer((size_t)tiles * 64 * W);
            for (uint32_t tile = 0; tile < tiles; ++tile)
                for (uint32_t k = 0; k < 32; ++k)
                    for (uint32_t c = 0; c < W; ++c) {
                        const uint64_t n2 = (uint64_t)W * tile + c;
                        outer[(size_t)tile * 64 * W + k * W + c] = tw_f64(n2 * k, N);                // A[k1][c]
                        outer[(size_t)tile * 64 * W + 32 * W + k * W + c] = tw_f64(32 * n2 * k, N);  // B[k2][c]
                    }
            st = upload_table(ctx, outer, &t->tw_outer);
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
            if (e != hipSuccess) { destroy_pipeline_objects(ctx, pl, false); return fail_hip(ctx, e,
                "hipMalloc(ring control)"); }
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
    const uint64_t slots = (uint64_t)group * (uint64_t)n_streams * (uint64_t)p->ring_rotate;  // transforms in the ring
    pl.ring_bytes = p->batch ? slots * (uint64_t)p->n * sizeof(v2f) : 0;
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

}  // namespace fwa_int
