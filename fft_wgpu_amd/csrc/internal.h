// internal.h -- what comm.cpp needs from the handles defined in api.cpp (internal to the library).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fft_wgpu_amd.h"

namespace fwa_int {
int32_t fail(const fwa_ctx *ctx, int32_t status, const std::string &msg);          // records the message, returns status
int32_t fail_hip(const fwa_ctx *ctx, hipError_t e, const char *what);
int32_t use_device(fwa_ctx *ctx);                                                    // hipSetDevice(ctx's ordinal)
int ctx_device(const fwa_ctx *ctx);
fwa_ctx *buf_ctx(const fwa_buf *b);
hipStream_t stream_raw(fwa_stream *s);                                               // nullptr -> the null stream
fwa_ctx *stream_ctx(fwa_stream *s);
}  // namespace fwa_int
