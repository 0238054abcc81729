// fft_wgpu.hpp -- header-only C++ mirror of the reference crate's public API over the C ABI
// (include/fft_wgpu_amd.h).  Same names and call shape as reference src/processor.rs:
//   Forward / Inverse / Onlyinverse / Normalize :: new(device, queue, src[, src2], fft_len), proc(encoder)
// plus the few wgpu objects its callers touch (src/examples/basic.rs:6-122).
// Errors: the reference unwraps / panics (examples/basic.rs:14,30,106); this mirror throws
// fft_wgpu::Error carrying the fwa_status and fwa_last_error_string().
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <type_traits>
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

// The shared library found at run time must implement the header this file was compiled against: a stale .so is refused
// when the first Device is made (or the devices are enumerated), not at the first call whose signature has changed.
inline void check_abi()
{
    const int32_t got = fwa_abi_version();
    if (got != FWA_ABI_VERSION)
        throw Error(FWA_ERR_UNSUPPORTED, "libfft_wgpu_amd.so reports ABI version " + std::to_string(got) +
                                             ", include/fft_wgpu_amd.h is version " + std::to_string(FWA_ABI_VERSION));
}

// wgpu::AdapterInfo of one device ordinal (instance.enumerate_adapters(..), lib.rs:33-35)
struct AdapterInfo {
    int ordinal = 0;
    std::string name;         // gcnArchName
    int32_t compute_units = 0;
    uint64_t hbm_bytes = 0;
    bool usable = false;      // Device(ordinal) would succeed (gfx950)
};

// One entry per visible device ordinal, no context created.  Empty when no device is visible (prepare_gpu -> None).
inline std::vector<AdapterInfo> enumerate_devices()
{
    check_abi();
    std::vector<AdapterInfo> out;
    int32_t n = 0;
    if (fwa_device_count(&n) != FWA_OK) return out;
    for (int32_t o = 0; o < n; ++o) {
        char name[256];
        AdapterInfo a;
        int32_t ok = 0;
        int32_t st = fwa_device_info(o, name, sizeof name, &a.compute_units, &a.hbm_bytes, &ok);
        if (st) throw Error(st, std::string("fwa_device_info: ") + fwa_last_error_string(nullptr));
        a.ordinal = o; a.name = name; a.usable = ok != 0;
        out.push_back(a);
    }
    return out;
}

// The slab rule: rank r of `world` owns transforms [first, first + count); = fwa_slab = sharding.slab.
struct Slab {
    uint64_t first = 0, count = 0;
};
inline Slab slab(uint64_t batch, int rank, int world)
{
    Slab s;
    int32_t st = fwa_slab(batch, rank, world, &s.first, &s.count);
    if (st) throw Error(st, std::string("fwa_slab: ") + fwa_last_error_string(nullptr));
    return s;
}

class Device {  // wgpu::Device (+ Instance/Adapter/Queue), lib.rs:29-62
public:
    explicit Device(int ordinal = 0) : ordinal_(ordinal)
    {
        check_abi();
        int32_t st = fwa_ctx_create(ordinal, &h_);
        if (st) throw Error(st, std::string("fwa_ctx_create: ") + fwa_last_error_string(nullptr));
    }
    int ordinal() const { return ordinal_; }
    ~Device() { fwa_ctx_destroy(h_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    fwa_ctx *raw() const { return h_; }
    void check(int32_t st, const char *where) const
    {
        if (st) throw Error(st, std::string(where) + ": " + fwa_last_error_string(h_));
    }

private:
    int ordinal_ = 0;
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
    const Device &device() const { return d_; }

private:
    const Device &d_;
    fwa_stream *h_ = nullptr;
};

class Event {  // one recorded point of an encoder's stream (no wgpu analogue: what map_async's callback stands for)
public:
    explicit Event(const Device &d) : d_(d) { d.check(fwa_event_create(d.raw(), &h_), "fwa_event_create"); }
    ~Event() { fwa_event_destroy(h_); }
    Event(const Event &) = delete;
    void record(CommandEncoder &e) { d_.check(fwa_event_record(h_, e.raw()), "fwa_event_record"); }
    void synchronize() { d_.check(fwa_event_synchronize(h_), "fwa_event_synchronize"); }          // host waits
    void make_wait(CommandEncoder &e) { d_.check(fwa_stream_wait_event(e.raw(), h_), "fwa_stream_wait_event"); }  // device-side
    float elapsed_ms(Event &end)
    {
        float ms = 0;
        d_.check(fwa_event_elapsed_ms(h_, end.h_, &ms), "fwa_event_elapsed_ms");
        return ms;
    }

private:
    const Device &d_;
    fwa_event *h_ = nullptr;
};

class PinnedArray {  // page-locked host staging (the reference's MAP_READ staging buffer, examples/basic.rs:50-55)
public:
    PinnedArray(const Device &d, uint64_t bytes) : d_(d), bytes_(bytes)
    {
        d.check(fwa_host_alloc(d.raw(), bytes, &p_), "fwa_host_alloc");
    }
    ~PinnedArray() { fwa_host_free(d_.raw(), p_); }
    PinnedArray(const PinnedArray &) = delete;
    void *data() const { return p_; }
    uint64_t size() const { return bytes_; }

private:
    const Device &d_;
    void *p_ = nullptr;
    uint64_t bytes_;
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
    void read_async(void *pinned_host, uint64_t bytes, CommandEncoder &e) const  // map_async without the poll: stream-ordered
    {
        d_->check(fwa_buf_download_async(pinned_host, h_, 0, bytes, e.raw()), "fwa_buf_download_async");
    }
    void copy_to(Buffer &dst, uint64_t bytes, CommandEncoder &e) const  // encoder.copy_buffer_to_buffer (basic.rs:84-90)
    {
        d_->check(fwa_buf_copy(dst.raw(), 0, h_, 0, bytes, e.raw()), "fwa_buf_copy");
    }
    // the five-argument form of copy_buffer_to_buffer; `dst` may live on another device (peer copy, fwa_buf_copy)
    void copy_to(Buffer &dst, uint64_t dst_offset, uint64_t src_offset, uint64_t bytes, CommandEncoder &e) const
    {
        e.device().check(fwa_buf_copy(dst.raw(), dst_offset, h_, src_offset, bytes, e.raw()), "fwa_buf_copy");
    }
    void write_at(uint64_t offset, const void *host, uint64_t bytes, CommandEncoder *e = nullptr)
    {
        d_->check(fwa_buf_upload(h_, offset, host, bytes, e ? e->raw() : nullptr), "fwa_buf_upload");
    }
    void read_at(void *host, uint64_t offset, uint64_t bytes, CommandEncoder &e) const  // blocking, stream-ordered on e
    {
        d_->check(fwa_buf_download(host, h_, offset, bytes, e.raw()), "fwa_buf_download");
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
    // fwa_plan_get_i64 ("launches_per_exec", "path", "group", ...)
    int64_t get(const char *key) const
    {
        int64_t v = 0;
        d_.check(fwa_plan_get_i64(h_, key, &v), "fwa_plan_get_i64");
        return v;
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

// The reference's benchmark loop (examples/basic.rs:72-127: write_buffer -> proc -> copy_buffer_to_buffer -> map / read back,
// every iteration) as a pipeline -- the C++ twin of fft_wgpu_amd/pipeline.py::HostPipeline: pinned staging, `slots`-fold
// buffered device buffers, two HIP streams (A: upload + transform + device copy, B: read-back) so that the read-back of
// iteration i overlaps the upload of i + 1 (the host link is full duplex).  One device-side dependency per iteration
// (A -> B); slot reuse is guarded by a HOST wait on the slot's read-back event, which the caller needs anyway before it
// touches the result (a second device-side dependency B -> A costs 2x on this stack: profiles/round2/host_pipeline_probe.jsonl).
class HostPipeline {
public:
    using PlanFactory = std::function<std::unique_ptr<detail::Plan>(const Device &, const Queue &, Buffer &)>;
    HostPipeline(const Device &device, const Queue &queue, PlanFactory make_plan, uint64_t n_samples, int slots = 2)
        : d_(device), bytes_(n_samples * sizeof(Complex)), ex_(device), down_(device)
    {
        for (int s = 0; s < slots; ++s) {
            hin_.emplace_back(new PinnedArray(device, bytes_));
            hout_.emplace_back(new PinnedArray(device, bytes_));
            src_.emplace_back(new Buffer(device, bytes_));
            staging_.emplace_back(new Buffer(device, bytes_));
            plans_.emplace_back(make_plan(device, queue, *src_.back()));
            e_ex_.emplace_back(new Event(device));
            e_down_.emplace_back(new Event(device));
        }
    }
    // Enqueue one iteration; blocks only until the slot's previous read-back (`slots` iterations ago) has finished.
    // `data` (n_samples Complex) is copied into the slot's pinned input; nullptr re-sends what the slot holds.
    int submit(const Complex *data = nullptr)
    {
        const int s = (int)(it_ % src_.size());
        if (it_ >= src_.size()) e_down_[s]->synchronize();  // hout / staging / src / hin of iteration it - slots are free
        if (data) std::memcpy(hin_[s]->data(), data, bytes_);
        src_[s]->write(hin_[s]->data(), bytes_, &ex_);                 // queue.write_buffer          basic.rs:73
        Buffer &out = plans_[s]->proc(ex_);                            // proc(&mut encoder)          basic.rs:79
        out.copy_to(*staging_[s], bytes_, ex_);                        // copy_buffer_to_buffer       basic.rs:84-90
        e_ex_[s]->record(ex_);
        e_ex_[s]->make_wait(down_);
        staging_[s]->read_async(hout_[s]->data(), bytes_, down_);      // map_async + get_mapped_range basic.rs:105-122
        e_down_[s]->record(down_);
        ++it_;
        return s;
    }
    Complex *input(int slot) { return static_cast<Complex *>(hin_[slot]->data()); }  // fill in place, then submit(nullptr)
    int next_slot() const { return (int)(it_ % src_.size()); }
    void wait_slot_free(int slot) { if (it_ >= src_.size()) e_down_[slot]->synchronize(); }
    // Wait for the read-back of `slot`; the pinned result stays valid until the slot is submitted again.
    const Complex *result(int slot)
    {
        e_down_[slot]->synchronize();
        return static_cast<const Complex *>(hout_[slot]->data());
    }
    void drain() { ex_.synchronize(); down_.synchronize(); }
    uint64_t bytes_per_iteration() const { return bytes_; }

private:
    const Device &d_;
    uint64_t bytes_;
    CommandEncoder ex_, down_;
    std::vector<std::unique_ptr<PinnedArray>> hin_, hout_;
    std::vector<std::unique_ptr<Buffer>> src_, staging_;
    std::vector<std::unique_ptr<detail::Plan>> plans_;
    std::vector<std::unique_ptr<Event>> e_ex_, e_down_;
    uint64_t it_ = 0;
};

// One batch sharded over several devices driven by ONE process (the Python twin is
// fft_wgpu_amd.ShardedBatch).  Transforms are independent (kernel/fft4.wgsl:21-23), so shard i is simply a Device + a
// CommandEncoder + a slab buffer + a plan of kind `PlanT` on ordinal `ordinals[i]`, and slab i = slab(batch, i, shards).
// proc() enqueues every shard's transform and returns at once; synchronize() joins.  Nothing communicates during the
// transform; scatter() / gather() move slabs from / to one buffer with peer copies (fwa_buf_copy across contexts) for callers
// whose batch starts on one device, write() / read() move them from / to host memory, each device taking only its own slab.
// `ordinals` empty = every visible device; an ordinal may repeat (two contexts on one device: what a one-GPU box can test).
template <class PlanT>
class ShardedBatch {
public:
    ShardedBatch(uint32_t fft_len, uint64_t batch, std::vector<int> ordinals = {}) : fft_len_(fft_len), batch_(batch)
    {
        if (ordinals.empty())
            for (const AdapterInfo &a : enumerate_devices())
                if (a.usable) ordinals.push_back(a.ordinal);
        if (ordinals.empty()) throw Error(FWA_ERR_NO_DEVICE, "ShardedBatch: no usable device");
        const int world = (int)ordinals.size();
        for (int r = 0; r < world; ++r) {
            dev_.emplace_back(new Device(ordinals[(size_t)r]));
            enc_.emplace_back(new CommandEncoder(*dev_.back()));
            slabs_.push_back(slab(batch, r, world));
            buf_.emplace_back(new Buffer(*dev_.back(), slabs_.back().count * 8ull * fft_len));
            if constexpr (std::is_constructible<PlanT, const Device &, const Queue &, Buffer &, Buffer &, uint32_t>::value) {
                second_.emplace_back(new Buffer(*dev_.back(), slabs_.back().count * 8ull * fft_len));  // Onlyinverse / Normalize
                plan_.emplace_back(new PlanT(*dev_.back(), *dev_.back(), *buf_.back(), *second_.back(), fft_len));
            } else {
                plan_.emplace_back(new PlanT(*dev_.back(), *dev_.back(), *buf_.back(), fft_len));
            }
            result_.push_back(buf_.back().get());
        }
    }
    size_t shards() const { return dev_.size(); }
    Slab slab_of(size_t i) const { return slabs_[i]; }
    Device &device(size_t i) { return *dev_[i]; }
    CommandEncoder &encoder(size_t i) { return *enc_[i]; }
    Buffer &buffer(size_t i) { return *buf_[i]; }      // shard i's slab (input of the transform)
    Buffer &result(size_t i) { return *result_[i]; }   // where shard i's output is after proc() (processor.rs:153-157)
    // Enqueue the transform of every slab on its device; returns when every shard's launches are QUEUED, without waiting
    // for them to run.  A C3-sized slab is 512 kernel launches + 512 event operations = about 3 ms of host time per device
    // against 21 ms of GPU work (profiles/round5/enqueue_cost.jsonl): eight devices enqueued one after another would be
    // host-bound, so shards whose exec is many launches are enqueued from one thread per shard (every entry point makes its
    // context's device current on the calling thread; plans of different contexts share nothing).  Link with -pthread.
    enum class Enqueue { automatic, serial, threaded };
    void set_enqueue(Enqueue e) { enqueue_ = e; }
    static constexpr int64_t thread_min_launches = 16;  // below this an exec returns in < 0.1 ms: a thread costs as much
    void proc()
    {
        const size_t n = dev_.size();
        bool threaded = enqueue_ == Enqueue::threaded;
        if (enqueue_ == Enqueue::automatic) {
            if (many_launches_ < 0) {   // the plans do not change after construction
                many_launches_ = 0;
                for (size_t i = 0; i < n; ++i) many_launches_ |= plan_[i]->get("launches_per_exec") >= thread_min_launches;
            }
            threaded = many_launches_ != 0;
        }
        if (n < 2 || !threaded) {
            for (size_t i = 0; i < n; ++i) result_[i] = &plan_[i]->proc(*enc_[i]);
            return;
        }
        std::vector<std::exception_ptr> err(n);
        auto one = [&](size_t i) {
            try { result_[i] = &plan_[i]->proc(*enc_[i]); } catch (...) { err[i] = std::current_exception(); }
        };
        std::vector<std::thread> workers;
        workers.reserve(n - 1);
        size_t started = 1;   // shards [1, started) have a worker; the rest are enqueued from this thread
        try {
            for (; started < n; ++started) workers.emplace_back(one, started);
        } catch (const std::system_error &) {
            // no more threads to be had: a vector of joinable threads must never be destroyed (std::terminate), so
            // the shards without a worker fall back to serial enqueue below and the started ones are joined as usual
        }
        one(0);
        for (size_t i = started; i < n; ++i) one(i);
        for (std::thread &t : workers) t.join();
        for (const std::exception_ptr &e : err)
            if (e) std::rethrow_exception(e);
    }
    void synchronize()
    {
        for (auto &e : enc_) e->synchronize();
    }
    // host memory -> slabs (batch * fft_len samples) / results -> host memory
    void write(const Complex *data)
    {
        for (size_t i = 0; i < dev_.size(); ++i)
            if (slabs_[i].count) buf_[i]->write_at(0, data + slabs_[i].first * fft_len_, slabs_[i].count * 8ull * fft_len_, enc_[i].get());
        synchronize();
    }
    void read(Complex *out)
    {
        for (size_t i = 0; i < dev_.size(); ++i)
            if (slabs_[i].count) result_[i]->read_at(out + slabs_[i].first * fft_len_, 0, slabs_[i].count * 8ull * fft_len_, *enc_[i]);
    }
    // `full` (the whole batch, on any shard's device, complete before the call) -> every shard's slab buffer
    void scatter(const Buffer &full)
    {
        for (size_t i = 0; i < dev_.size(); ++i)
            if (slabs_[i].count) full.copy_to(*buf_[i], 0, slabs_[i].first * 8ull * fft_len_, slabs_[i].count * 8ull * fft_len_, *enc_[i]);
    }
    // every shard's result -> `full`, each copy ordered behind its shard's transform; waits for all of them
    void gather(Buffer &full)
    {
        for (size_t i = 0; i < dev_.size(); ++i)
            if (slabs_[i].count) result_[i]->copy_to(full, slabs_[i].first * 8ull * fft_len_, 0, slabs_[i].count * 8ull * fft_len_, *enc_[i]);
        synchronize();
    }

private:
    uint32_t fft_len_;
    uint64_t batch_;
    std::vector<std::unique_ptr<Device>> dev_;
    std::vector<std::unique_ptr<CommandEncoder>> enc_;
    std::vector<Slab> slabs_;
    std::vector<std::unique_ptr<Buffer>> buf_, second_;
    std::vector<std::unique_ptr<PlanT>> plan_;
    std::vector<Buffer *> result_;
    Enqueue enqueue_ = Enqueue::automatic;
    int many_launches_ = -1;
};

// One rank of a slab communicator over RCCL (fwa_comm_*), for hosts that run ONE PROCESS PER GPU and do not hold each other's
// pointers (the twin of fft_wgpu_amd.sharding.Comm and rust_shim's sharded::Comm).  Rank 0 calls Comm::unique_id() and hands the
// bytes to the other ranks (file, socket, launcher environment); every rank then constructs its Comm (collective).  scatter /
// gather move whole-transform slabs by the slab rule, stream-ordered on `enc`; the transform itself never communicates.
class Comm {
public:
    using Id = std::array<uint8_t, FWA_COMM_ID_BYTES>;
    static Id unique_id()
    {
        Id id{};
        int32_t st = fwa_comm_unique_id(id.data());
        if (st) throw Error(st, std::string("fwa_comm_unique_id: ") + fwa_last_error_string(nullptr));
        return id;
    }
    Comm(const Device &d, const Id &id, int world, int rank) : d_(d), world_(world), rank_(rank)
    {
        d.check(fwa_comm_create(d.raw(), id.data(), world, rank, &h_), "fwa_comm_create");
    }
    ~Comm() { fwa_comm_destroy(h_); }
    Comm(const Comm &) = delete;
    int world() const { return world_; }
    int rank() const { return rank_; }
    Slab my_slab(uint64_t batch) const { return slab(batch, rank_, world_); }
    // root's `full` (nullptr elsewhere) -> every rank's `slab_buf`
    void scatter(int root, const Buffer *full, Buffer &slab_buf, uint32_t fft_len, uint64_t batch, CommandEncoder &enc)
    {
        d_.check(fwa_comm_scatter(h_, root, full ? full->raw() : nullptr, slab_buf.raw(), fft_len, batch, enc.raw()), "fwa_comm_scatter");
    }
    // every rank's `slab_buf` (e.g. the buffer proc() returned) -> root's `full`
    void gather(int root, const Buffer &slab_buf, Buffer *full, uint32_t fft_len, uint64_t batch, CommandEncoder &enc)
    {
        d_.check(fwa_comm_gather(h_, root, slab_buf.raw(), full ? full->raw() : nullptr, fft_len, batch, enc.raw()), "fwa_comm_gather");
    }
    // one send and / or one receive in one group (rank < 0: none); to the own rank only with the matching receive
    void sendrecv(const Buffer *send, uint64_t send_offset, uint64_t send_bytes, int send_to, Buffer *recv, uint64_t recv_offset,
                  uint64_t recv_bytes, int recv_from, CommandEncoder &enc)
    {
        d_.check(fwa_comm_sendrecv(h_, send ? send->raw() : nullptr, send_offset, send_bytes, send_to, recv ? recv->raw() : nullptr,
                                   recv_offset, recv_bytes, recv_from, enc.raw()),
                 "fwa_comm_sendrecv");
    }

private:
    const Device &d_;
    fwa_comm *h_ = nullptr;
    int world_, rank_;
};

}  // namespace fft_wgpu
