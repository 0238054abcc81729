// fft_wgpu.hpp -- header-only C++ mirror of the reference crate's public API over the C ABI
// (include/fft_wgpu_amd.h).  Same names and call shape as reference src/processor.rs:
//   Forward / Inverse / Onlyinverse / Normalize :: new(device, queue, src[, src2], fft_len), proc(encoder)
// plus the few wgpu objects its callers touch (src/examples/basic.rs:6-122).
// Errors: the reference unwraps / panics (examples/basic.rs:14,30,106); this mirror throws
// fft_wgpu::Error carrying the fwa_status and fwa_last_error_string().
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "fft_wgpu_amd.h"

namespace fft_wgpu {

struct Complex {  // src/lib.rs:10-15
    float real, imag;
};

class Error : public std::runtime_error {
public:
    Error(int32_t st, const std::string &what) : std::runtime_error(what), status(st) {}
    int32_t status;
};

class Device {  // wgpu::Device (+ Instance/Adapter/Queue), lib.rs:29-62
public:
    explicit Device(int ordinal = 0)
    {
        int32_t st = fwa_ctx_create(ordinal, &h_);
        if (st) throw Error(st, std::string("fwa_ctx_create: ") + fwa_last_error_string(nullptr));
    }
    ~Device() { fwa_ctx_destroy(h_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    fwa_ctx *raw() const { return h_; }
    void check(int32_t st, const char *where) const
    {
        if (st) throw Error(st, std::string(where) + ": " + fwa_last_error_string(h_));
    }

private:
    fwa_ctx *h_ = nullptr;
};
using Queue = Device;  // the reference passes both (&device, &queue); one context plays both roles here

class CommandEncoder {  // wgpu::CommandEncoder: an in-order HIP stream
public:
    explicit CommandEncoder(const Device &d) : d_(d) { d.check(fwa_stream_create(d.raw(), &h_), "fwa_stream_create"); }
    ~CommandEncoder() { fwa_stream_destroy(h_); }
    CommandEncoder(const CommandEncoder &) = delete;
    fwa_stream *raw() const { return h_; }
    void synchronize() { d_.check(fwa_stream_synchronize(h_), "fwa_stream_synchronize"); }  // submit + poll(wait)

private:
    const Device &d_;
    fwa_stream *h_ = nullptr;
};

class Buffer {  // wgpu::Buffer (examples/basic.rs:50-64)
public:
    Buffer(const Device &d, uint64_t bytes) : d_(&d), own_(true) { d.check(fwa_buf_alloc(d.raw(), bytes, &h_), "fwa_buf_alloc"); }
    Buffer(const Device &d, fwa_buf *borrowed) : d_(&d), h_(borrowed), own_(false) {}
    ~Buffer() { if (own_) fwa_buf_free(h_); }
    Buffer(const Buffer &) = delete;
    fwa_buf *raw() const { return h_; }
    uint64_t size() const { return fwa_buf_size(h_); }
    void write(const void *host, uint64_t bytes, CommandEncoder *e = nullptr)  // queue.write_buffer
    {
        d_->check(fwa_buf_upload(h_, 0, host, bytes, e ? e->raw() : nullptr), "fwa_buf_upload");
    }
    void read(void *host, uint64_t bytes, CommandEncoder *e = nullptr) const  // map_async + poll + get_mapped_range
    {
        // without an encoder: device.poll(Maintain::wait()) semantics, all submitted work has finished first
        if (!e) d_->check(fwa_ctx_synchronize(d_->raw()), "fwa_ctx_synchronize");
        d_->check(fwa_buf_download(host, h_, 0, bytes, e ? e->raw() : nullptr), "fwa_buf_download");
    }

private:
    const Device *d_;
    fwa_buf *h_ = nullptr;
    bool own_;
};

namespace detail {
class Plan {
public:
    Plan(const Device &d, int32_t kind, Buffer &src, Buffer *src2, uint32_t fft_len)
        : fft_len(fft_len), d_(d), a_(src), b_(src2), other_(d, nullptr)
    {
        d.check(fwa_plan_create(d.raw(), kind, fft_len, src.raw(), src2 ? src2->raw() : nullptr, &h_), "fwa_plan_create");
    }
    ~Plan() { fwa_plan_destroy(h_); }
    Plan(const Plan &) = delete;
    // proc(&self, &mut encoder) -> &wgpu::Buffer (processor.rs:110,293,467,622)
    Buffer &proc(CommandEncoder &enc)
    {
        fwa_buf *res = nullptr;
        d_.check(fwa_plan_exec(h_, enc.raw(), &res), "fwa_plan_exec");
        if (res == a_.raw()) return a_;
        if (b_ && res == b_->raw()) return *b_;
        other_.~Buffer();
        new (&other_) Buffer(d_, res);  // plan-owned partner (Forward/Inverse, odd log2 n): borrowed view
        return other_;
    }
    const uint32_t fft_len;

private:
    const Device &d_;
    Buffer &a_;
    Buffer *b_;
    Buffer other_;
    fwa_plan *h_ = nullptr;
};
}  // namespace detail

struct Forward : detail::Plan {  // processor.rs:7-159
    Forward(const Device &device, const Queue &, Buffer &src, uint32_t fft_len) : Plan(device, FWA_FORWARD, src, nullptr, fft_len) {}
};
struct Inverse : detail::Plan {  // processor.rs:231-341
    Inverse(const Device &device, const Queue &, Buffer &src, uint32_t fft_len) : Plan(device, FWA_INVERSE_SCALED, src, nullptr, fft_len) {}
};
struct Onlyinverse : detail::Plan {  // processor.rs:566-670
    Onlyinverse(const Device &device, const Queue &, Buffer &src, Buffer &src2, uint32_t fft_len) : Plan(device, FWA_INVERSE_UNSCALED, src, &src2, fft_len) {}
};
struct Normalize : detail::Plan {  // processor.rs:409-505
    Normalize(const Device &device, const Queue &, Buffer &buffer1, Buffer &buffer2, uint32_t fft_len) : Plan(device, FWA_NORMALIZE, buffer1, &buffer2, fft_len) {}
};

}  // namespace fft_wgpu
