// tuning.cpp -- fwa_plan_get_i64 / fwa_plan_set_i64: what a plan chose at creation, and the keys that re-tune it
// before its first exec.  No key changes what a plan computes (tests/test_gpu_parity.py: every alternative is
// bit-identical where it shares arithmetic); the reference has no counterpart (one radix-2 loop, src/kernel/fft4.wgsl).
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
    else if (k == "tile_w") *value = 16;   // one tile width ships (kept as a key: bench lines of every round carry it)
    else if (k == "xcd_swizzle") {
        const int64_t dflt = plan->path == PATH_TWOPASS_1M ? 5
                             : (plan->path == PATH_TILED ? (int64_t)tiled_swizzle_default(plan) : 0);
        *value = plan->xcd_swizzle < 0 ? dflt : plan->xcd_swizzle;
    }
    else if (k == "depth") *value = plan->depth;
    else if (k == "ring_slots") *value = plan->ring_slots;
    else if (k == "wgs") *value = plan->wgs;
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
    // other holders: cache + plans
    else if (k == "tables_shared") *value = plan->tb ? (int64_t)plan->tb.use_count() - 1 : 0;
    else if (k == "scratch_bytes")
        *value = (int64_t)plan->ring_bytes + (plan->second_owned ? (int64_t)plan->own_second.bytes : 0) +
                 (plan->ring_ctl ? (int64_t)ctl_bytes(plan) : 0);
    else if (k == "launches_per_exec") {
        switch (plan->path) {
            case PATH_TWOPASS_1M: *value = 2 * ng; break;
            case PATH_RING_1M: *value = 1; break;
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
        if (value < 1 || value > (k == "streams" ? 16 : 4096))
            return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        return build_pipeline(plan, k == "group" ? value : plan->group, k == "streams" ? value : plan->n_streams);
    }
    if (k == "inject_launch_failure") {
        // laboratory: the launch of group `value` fails once (nothing is enqueued for it): the error path of run_groups
        if (!kLab)
            return fail(ctx, FWA_ERR_UNSUPPORTED,
                        "inject_launch_failure is a laboratory knob (libfft_wgpu_amd_lab.so)");
        if (plan->path != PATH_TWOPASS_1M && plan->path != PATH_TILED)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the pipelined paths");
        plan->inject_fail_group = value;
        return FWA_OK;
    }
    if (k == "ring_rotate") {
        if (!kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "ring_rotate is a laboratory knob (libfft_wgpu_amd_lab.so)");
        if (plan->path != PATH_TWOPASS_1M)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the 2^20 two-pass path");
        if (value < 1 || value > 64) return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        const int64_t old = plan->ring_rotate;
        plan->ring_rotate = value;
        const int32_t st = build_pipeline(plan, plan->group, plan->n_streams);
        if (st) plan->ring_rotate = old;
        return st;
    }
    if (k == "tile_w") {
        if (plan->path != PATH_TWOPASS_1M)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the 2^20 two-pass path");
        if (value != 16) return fail(ctx, FWA_ERR_UNSUPPORTED, "tile_w is 16 (the 32-column tile left the tree in round 6)");
        return FWA_OK;
    }
    if (k == "depth" || k == "ring_slots" || k == "wgs") {
        if (!kLab) return fail(ctx, FWA_ERR_UNSUPPORTED, "key belongs to a laboratory path (libfft_wgpu_amd_lab.so)");
        if (plan->path != PATH_RING_1M)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to the persistent 2^20 path");
        if (value < 1 || value > 65536) return fail(ctx, FWA_ERR_INVALID_ARG, "value out of range");
        if (k == "wgs") { plan->wgs = value; return FWA_OK; }
        const int64_t d = k == "depth" ? value : plan->depth, r = k == "ring_slots" ? value : plan->ring_slots;
        if (k == "depth") { plan->depth = d; if (r < d + 1) plan->ring_slots = d + 1; }
        else {
            if (r < plan->depth + 1) return fail(ctx, FWA_ERR_INVALID_ARG, "ring_slots must exceed depth");
            plan->ring_slots = r;
        }
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
            if (f[i] < 6 || f[i] > top)
                return fail(ctx, FWA_ERR_INVALID_ARG,
                            "every factor must be 2^6..2^10 (2^11: first; 2^11, 2^12: second of two)");
            sum += f[i];
        }
        if (sum != plan->lg || (value >> 24))
            return fail(ctx, FWA_ERR_INVALID_ARG, "factors do not multiply to fft_len");
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
        int64_t &flag = k == "p1_gen" ? plan->p1_gen : k == "rows32" ? plan->rows32
                        : k == "colsw" ? plan->colsw : plan->tile_ring;
        flag = value != 0;
        return FWA_OK;
    }
    if (k == "small_reg") {
        if (plan->path != PATH_SMALL) return fail(ctx, FWA_ERR_UNSUPPORTED, "key only applies to n <= 32768");
        if (value != 1 && !kLab)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "small_reg != 1 selects laboratory kernels (libfft_wgpu_amd_lab.so)");
        // 1: k_chunk (2 .. 256) and k_small32 (from 512), the default; 3: the direct-addressing kernel k_small16 at
        // 16 .. 4096 (A/B); 2: as 3 up to 256 with the wavefront-shuffle exchange at n = 32 / 64 / 128
        if (value < 1 || value > 3) return fail(ctx, FWA_ERR_INVALID_ARG, "small_reg is 1, 2 or 3");
        plan->small_reg = value;
        return FWA_OK;
    }
    if (k == "path") {
        if (plan->kind == FWA_NORMALIZE) return fail(ctx, FWA_ERR_UNSUPPORTED, "normalize has one path");
        if (value == plan->path) return FWA_OK;
        if (value == PATH_RING_1M && !kLab)
            return fail(ctx, FWA_ERR_UNSUPPORTED, "path 5 is a laboratory path (libfft_wgpu_amd_lab.so)");
        if ((value == PATH_RING_1M || value == PATH_TWOPASS_1M)
            && (plan->path == PATH_RING_1M || plan->path == PATH_TWOPASS_1M)) {
            // the two forms of the 2^20 pipeline: per-group launches with a large ring, or one persistent launch
            const int64_t old = plan->path;
            plan->path = value;
            const int32_t st = setup_path(plan);
            if (st) plan->path = old;
            if (!st && value == PATH_TWOPASS_1M
                && plan->ring_ctl) { (void)hipFree(plan->ring_ctl); plan->ring_ctl = nullptr; }
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
        return fail(ctx, FWA_ERR_UNSUPPORTED,
                    "only path = 2 (the literal radix-2 recurrence) or, at n = 2^20, 1 / 5 can be set");
    }
    return fail(ctx, FWA_ERR_INVALID_ARG, "unknown key: " + k);
}

}  // extern "C"
