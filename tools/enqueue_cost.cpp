// Host cost of enqueueing one exec, and of one process driving several devices.
//   enqueue_cost [shards] [transforms_per_shard] [reps]        default 8 shards x 512 transforms of 2^20 on device 0
// Prints JSON lines: the wall time for fwa_plan_exec / ShardedBatch::proc() to RETURN (launches queued, nothing waited
// for) and the time until the work has finished, serial and threaded enqueue.  One GPU suffices: the host cost of a
// launch does not depend on which device it goes to; the shards then share the GPU, so "done" times are not a
// multi-GPU figure -- only the "returns" columns are the point.
// Build: g++ -O2 -std=c++17 -Iinclude tools/enqueue_cost.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -pthread
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_wgpu.hpp"

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    using namespace fft_wgpu;
    const int shards = argc > 1 ? std::atoi(argv[1]) : 8;
    const uint64_t per = argc > 2 ? (uint64_t)std::atoll(argv[2]) : 512;
    const int reps = argc > 3 ? std::atoi(argv[3]) : 5;
    const uint32_t n = 1u << 20;
    try {
        ShardedBatch<Forward> sb(n, per * (uint64_t)shards, std::vector<int>((size_t)shards, 0));
        for (size_t i = 0; i < sb.shards(); ++i)   // finite data at scale 2^-40
            sb.device(i).check(fwa_fill_synthetic(sb.buffer(i).raw(), 0x5EED, sb.slab_of(i).first, n, 1.0f / 1099511627776.0f, sb.encoder(i).raw()), "fill");
        sb.synchronize();
        for (auto mode : {ShardedBatch<Forward>::Enqueue::serial, ShardedBatch<Forward>::Enqueue::threaded}) {
            sb.set_enqueue(mode);
            std::vector<double> ret, done;
            for (int r = 0; r < reps + 1; ++r) {
                for (size_t i = 0; i < sb.shards(); ++i)
                    sb.device(i).check(fwa_fill_synthetic(sb.buffer(i).raw(), 0x5EED, sb.slab_of(i).first, n, 1.0f / 1099511627776.0f, sb.encoder(i).raw()), "fill");
                sb.synchronize();
                const double t0 = now_ms();
                sb.proc();
                const double t1 = now_ms();
                sb.synchronize();
                const double t2 = now_ms();
                if (r) { ret.push_back(t1 - t0); done.push_back(t2 - t0); }   // first repetition: warm-up
            }
            std::sort(ret.begin(), ret.end());
            std::sort(done.begin(), done.end());
            std::printf("{\"host\": \"c++\", \"what\": \"ShardedBatch<Forward>::proc\", \"enqueue\": \"%s\", \"shards\": %d, \"fft_len\": %u, "
                        "\"transforms_per_shard\": %llu, \"launches_per_shard\": %lld, \"returns_ms_median\": %.3f, \"returns_ms_min\": %.3f, "
                        "\"done_ms_median\": %.3f, \"reps\": %d}\n",
                        mode == ShardedBatch<Forward>::Enqueue::serial ? "serial" : "threaded", shards, n, (unsigned long long)per,
                        (long long)(2 * ((per + 15) / 16)), ret[ret.size() / 2], ret.front(), done[done.size() / 2], reps);
        }
        return 0;
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 2;
    }
}
