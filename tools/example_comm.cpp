// One rank of a one-process-per-GPU host in C++ (include/fft_wgpu.hpp: fft_wgpu::Comm over fwa_comm_*, RCCL): scatter a batch
// from rank 0, transform the own slab, gather -- and compare with the unsharded transform on the root.
//   example_comm [log2_fft_len] [batch]          world / rank / device from WORLD_SIZE / RANK / LOCAL_RANK (default 1 / 0 / 0);
//                                                the communicator id travels through the file $FWA_COMM_ID_FILE (rank 0 writes it)
// On a one-GPU box only the world of one rank can run (RCCL refuses two ranks on one device): that is what the tests start.
// Build: g++ -std=c++17 -Iinclude tools/example_comm.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -pthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include "fft_wgpu.hpp"

static int env_int(const char *k, int d) { const char *v = std::getenv(k); return v ? std::atoi(v) : d; }

int main(int argc, char **argv)
{
    using namespace fft_wgpu;
    const uint32_t lg = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 12;
    const uint64_t batch = argc > 2 ? (uint64_t)std::atoll(argv[2]) : 13;
    const uint32_t n = 1u << lg;
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", 0);
    const char *id_file = std::getenv("FWA_COMM_ID_FILE");
    try {
        Device dev(local);
        CommandEncoder enc(dev);
        Comm::Id id{};
        if (rank == 0) {
            id = Comm::unique_id();
            if (id_file) { std::ofstream f(std::string(id_file) + ".tmp", std::ios::binary); f.write((const char *)id.data(), id.size()); f.close(); std::rename((std::string(id_file) + ".tmp").c_str(), id_file); }
        } else {
            if (!id_file) throw Error(FWA_ERR_INVALID_ARG, "FWA_COMM_ID_FILE is needed with more than one rank");
            for (int tries = 0;; ++tries) {
                std::ifstream f(id_file, std::ios::binary);
                if (f && f.read((char *)id.data(), id.size())) break;
                if (tries > 600) throw Error(FWA_ERR_INVALID_ARG, "no communicator id after 60 s");
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        Comm comm(dev, id, world, rank);
        const Slab mine = comm.my_slab(batch);
        const uint64_t bytes = batch * n * sizeof(Complex), my_bytes = mine.count * n * sizeof(Complex);
        std::vector<Complex> x, ref, y;
        Buffer slab_buf(dev, my_bytes);
        std::unique_ptr<Buffer> full, back;
        if (rank == 0) {
            x.resize(batch * n); ref.resize(batch * n); y.resize(batch * n);
            uint32_t s = 0xC0FFEEu;
            for (Complex &c : x) {
                s = s * 1664525u + 1013904223u; c.real = (float)(int32_t)s * (1.0f / 2147483648.0f);
                s = s * 1664525u + 1013904223u; c.imag = (float)(int32_t)s * (1.0f / 2147483648.0f);
            }
            full.reset(new Buffer(dev, bytes)); back.reset(new Buffer(dev, bytes));
            Buffer tmp(dev, bytes);
            tmp.write(x.data(), bytes, &enc);
            { Forward f(dev, dev, tmp, n); f.proc(enc).read(ref.data(), bytes, &enc); }   // the unsharded transform
            full->write(x.data(), bytes, &enc);
        }
        comm.scatter(0, full.get(), slab_buf, n, batch, enc);
        Forward fwd(dev, dev, slab_buf, n);
        Buffer &out = fwd.proc(enc);                                                        // no communication
        comm.gather(0, out, back.get(), n, batch, enc);
        enc.synchronize();
        if (rank == 0) {
            back->read(y.data(), bytes, &enc);
            if (std::memcmp(y.data(), ref.data(), bytes) != 0) { std::fprintf(stderr, "gathered result != unsharded transform\n"); return 1; }
            // the primitive: a ring shift of the first kilobyte (to / from the own rank in a world of one)
            Buffer ring(dev, 1024);
            const int to = (rank + 1) % world, from = (rank + world - 1) % world;
            comm.sendrecv(back.get(), 0, 1024, to, &ring, 0, 1024, from, enc);
            std::vector<unsigned char> head(1024);
            ring.read(head.data(), 1024, &enc);
            if (world == 1 && std::memcmp(head.data(), ref.data(), 1024) != 0) { std::fprintf(stderr, "sendrecv moved the wrong bytes\n"); return 1; }
        } else {
            Buffer ring(dev, 1024), mine_head(dev, 1024);
            comm.sendrecv(&mine_head, 0, 1024, (rank + 1) % world, &ring, 0, 1024, (rank + world - 1) % world, enc);
            enc.synchronize();
        }
        std::printf("comm ok: rank %d of %d on device %d, n=2^%u batch=%llu, slab [%llu, +%llu)\n", rank, world, local, lg,
                    (unsigned long long)batch, (unsigned long long)mine.first, (unsigned long long)mine.count);
        return 0;
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 2;
    }
}
