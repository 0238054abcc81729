// One batch sharded over several contexts from ONE process through include/fft_wgpu.hpp (fft_wgpu::ShardedBatch), checked
// bit for bit against the unsharded transform of the same samples on device 0.
//   example_sharded [log2_fft_len] [batch] [shards]      shards = 0: one per visible device; an ordinal repeats when
//                                                        shards exceeds the device count (two contexts on one GPU)
// Legs: (1) host -> slabs -> proc -> host; (2) the batch starts on device 0: scatter (peer copies) -> proc -> gather;
// (2b) proc() enqueued from one thread per shard and serially;
// (3) Onlyinverse + Normalize shards (caller-supplied second buffers) undo leg 1.  Exit 0 = all identical, 1 = mismatch,
// 2 = library error (no device: "error 5").
// Build: g++ -std=c++17 -Iinclude tools/example_sharded.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -pthread
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fft_wgpu.hpp"

int main(int argc, char **argv)
{
    using namespace fft_wgpu;
    const uint32_t lg = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 10;
    const uint64_t batch = argc > 2 ? (uint64_t)std::atoll(argv[2]) : 37;
    int shards = argc > 3 ? std::atoi(argv[3]) : 0;
    const uint32_t n = 1u << lg;
    try {
        const std::vector<AdapterInfo> adapters = enumerate_devices();   // instance.enumerate_adapters (lib.rs:33-35)
        for (const AdapterInfo &a : adapters)
            std::printf("adapter %d: %s, %d CUs, %.1f GiB%s\n", a.ordinal, a.name.c_str(), a.compute_units, a.hbm_bytes / 1073741824.0,
                        a.usable ? "" : " (not usable)");
        if (adapters.empty()) throw Error(FWA_ERR_NO_DEVICE, "no device visible");
        if (shards <= 0) shards = (int)adapters.size();
        std::vector<int> ordinals;
        for (int i = 0; i < shards; ++i) ordinals.push_back(adapters[(size_t)i % adapters.size()].ordinal);

        std::vector<Complex> x(batch * n), ref(batch * n), y(batch * n);
        uint32_t s = 0x5EEDu;
        for (Complex &c : x) {
            s = s * 1664525u + 1013904223u; c.real = (float)(int32_t)s * (1.0f / 2147483648.0f);
            s = s * 1664525u + 1013904223u; c.imag = (float)(int32_t)s * (1.0f / 2147483648.0f);
        }
        const uint64_t bytes = batch * n * sizeof(Complex);

        // the unsharded transform on device 0
        Device d0(adapters[0].ordinal);
        CommandEncoder e0(d0);
        Buffer full(d0, bytes);
        if (bytes) full.write(x.data(), bytes, &e0);
        {
            Forward fwd(d0, d0, full, n);
            Buffer &out = fwd.proc(e0);
            if (bytes) out.read(ref.data(), bytes, &e0);
        }

        // leg 1: host -> slabs -> proc on every shard -> host
        ShardedBatch<Forward> sb(n, batch, ordinals);
        uint64_t covered = 0;
        for (size_t i = 0; i < sb.shards(); ++i) {
            if (sb.slab_of(i).first != covered) { std::fprintf(stderr, "slabs do not tile the batch\n"); return 1; }
            covered += sb.slab_of(i).count;
        }
        if (covered != batch) { std::fprintf(stderr, "slabs do not cover the batch\n"); return 1; }
        sb.write(x.data());
        sb.proc();
        sb.read(y.data());
        if (bytes && std::memcmp(y.data(), ref.data(), bytes) != 0) { std::fprintf(stderr, "leg 1: sharded != unsharded\n"); return 1; }

        // leg 2: the batch starts on device 0 and returns to it: scatter -> proc -> gather (peer copies across contexts)
        if (bytes) full.write(x.data(), bytes, &e0);
        e0.synchronize();
        sb.scatter(full);
        sb.proc();
        Buffer back(d0, bytes);
        sb.gather(back);
        std::memset(y.data(), 0, bytes);
        if (bytes) back.read(y.data(), bytes, &e0);
        if (bytes && std::memcmp(y.data(), ref.data(), bytes) != 0) { std::fprintf(stderr, "leg 2: scatter/proc/gather != unsharded\n"); return 1; }

        // leg 2b: both enqueue forms of proc() -- one host thread per shard, and one after another -- give the same bits
        for (auto mode : {ShardedBatch<Forward>::Enqueue::threaded, ShardedBatch<Forward>::Enqueue::serial}) {
            sb.set_enqueue(mode);
            sb.write(x.data());
            sb.proc();
            std::memset(y.data(), 0, bytes);
            sb.read(y.data());
            if (bytes && std::memcmp(y.data(), ref.data(), bytes) != 0) { std::fprintf(stderr, "leg 2b: enqueue mode %d != unsharded\n", (int)mode); return 1; }
        }
        sb.set_enqueue(ShardedBatch<Forward>::Enqueue::automatic);

        // leg 3: Onlyinverse then Normalize per shard brings the samples back (<= 1e-5 relative, the reference's bound)
        ShardedBatch<Onlyinverse> inv(n, batch, ordinals);
        inv.write(ref.data());
        inv.proc();
        double worst = 0, scale = 0;
        for (size_t i = 0; i < inv.shards(); ++i) {
            const Slab sl = inv.slab_of(i);
            if (!sl.count) continue;
            // Normalize(buffer1, buffer2): reads the buffer that holds Onlyinverse's result, writes the other (processor.rs:433-439)
            Buffer &res = inv.result(i);
            Buffer other(inv.device(i), sl.count * 8ull * n);
            Buffer &b1 = (lg % 2 == 0) ? res : other, &b2 = (lg % 2 == 0) ? other : res;
            Normalize norm(inv.device(i), inv.device(i), b1, b2, n);
            Buffer &z = norm.proc(inv.encoder(i));
            std::vector<Complex> zs(sl.count * n);
            z.read_at(zs.data(), 0, zs.size() * sizeof(Complex), inv.encoder(i));
            for (size_t k = 0; k < zs.size(); ++k) {
                const Complex &a = zs[k], &b = x[sl.first * n + k];
                worst = std::max(worst, (double)std::max(std::abs(a.real - b.real), std::abs(a.imag - b.imag)));
                scale = std::max(scale, (double)std::max(std::abs(b.real), std::abs(b.imag)));
            }
        }
        if (scale > 0 && worst > 1e-5 * scale) { std::fprintf(stderr, "leg 3: round trip error %g\n", worst / scale); return 1; }
        std::printf("sharded ok: n=2^%u batch=%llu shards=%zu devices=%zu round-trip %.3g\n", lg, (unsigned long long)batch, sb.shards(),
                    adapters.size(), scale > 0 ? worst / scale : 0.0);
        return 0;
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 2;
    }
}
