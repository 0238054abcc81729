// C++ replay of the reference's benchmark loop, src/examples/basic.rs:3-131 (N = 512, 2500 transforms = 1.28 M samples per
// iteration, upload + Forward::proc + copy + read-back EVERY iteration), through fft_wgpu::HostPipeline of
// include/fft_wgpu.hpp -- and with every read-back sample checked: transform t of iteration i is an impulse of amplitude a
// at position p (both functions of (i, t)), whose DFT is a * exp(-2 pi i p k / n) for every k.
//   g++ -O2 -std=c++17 -Iinclude tools/example_basic_pipeline.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -o example_basic_pipeline
// Prints the PCIe-inclusive rate (never the headline value: DESIGN.md 4) and exits non-zero on any wrong sample.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fft_wgpu.hpp"

int main(int argc, char **argv)
{
    using namespace fft_wgpu;
    const int iters = argc > 1 ? std::atoi(argv[1]) : 300;
    const int slots = argc > 2 ? std::atoi(argv[2]) : 3;
    const uint32_t n = 512, batch = 2500;                       // basic.rs:32,66
    const uint64_t samples = (uint64_t)n * batch;
    try {
        Device device(0);
        const Queue &queue = device;
        HostPipeline pipe(device, queue,
                          [n](const Device &d, const Queue &q, Buffer &src) { return std::unique_ptr<detail::Plan>(new Forward(d, q, src, n)); },
                          samples, slots);
        std::vector<double> cs(n), sn(n);
        for (uint32_t k = 0; k < n; ++k) { cs[k] = std::cos(-2.0 * M_PI * k / n); sn[k] = std::sin(-2.0 * M_PI * k / n); }
        auto pos = [&](int it, uint32_t t) { return (uint32_t)((it * 7919u + t * 104729u + 13u) % n); };
        auto amp = [&](int it, uint32_t t) { return 0.5f + (float)((it * 31 + t * 17) % 97) / 97.0f; };
        auto fill = [&](int it, Complex *dst) {
            std::memset(dst, 0, samples * sizeof(Complex));
            for (uint32_t t = 0; t < batch; ++t) dst[(uint64_t)t * n + pos(it, t)] = Complex{amp(it, t), -0.5f * amp(it, t)};
        };
        double worst = 0.0;
        uint64_t checked = 0;
        auto check = [&](int it, const Complex *y) {
            for (uint32_t t = 0; t < batch; ++t) {
                const uint32_t p = pos(it, t);
                const double ar = amp(it, t), ai = -0.5 * ar;
                for (uint32_t k = 0; k < n; ++k) {
                    const uint32_t e = (uint32_t)(((uint64_t)p * k) % n);
                    const double er = ar * cs[e] - ai * sn[e], ei = ar * sn[e] + ai * cs[e];
                    const double d = std::fmax(std::fabs(y[(uint64_t)t * n + k].real - er), std::fabs(y[(uint64_t)t * n + k].imag - ei));
                    if (d > worst) worst = d;
                }
            }
            checked += samples;
        };
        // warm-up (plan creation, first touches), then the timed loop; results are checked one iteration behind the submit
        std::vector<int> slot_of(iters + slots);
        for (int w = 0; w < slots; ++w) { int s = pipe.next_slot(); pipe.wait_slot_free(s); fill(-1 - w, pipe.input(s)); pipe.submit(); }
        pipe.drain();
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) {
            const int s = pipe.next_slot();
            if (it >= slots) check(it - slots, pipe.result(s));   // the slot's previous result, before it is reused
            else pipe.wait_slot_free(s);
            fill(it, pipe.input(s));
            slot_of[it] = pipe.submit();
        }
        for (int it = std::max(0, iters - slots); it < iters; ++it) check(it, pipe.result(slot_of[it]));
        pipe.drain();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // the same loop without the host-side fill / check (pure transfer + transform pipeline): the link-bound rate
        const auto t1 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) pipe.submit();
        pipe.drain();
        const double sec2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        // the host link alone, measured in this process right after the pipeline: the same 10.24 MB each way per iteration on
        // two streams (pinned host memory, both directions busy), no transform, no device copy, no dependency between the
        // directions -- the ceiling the pipeline is judged against (tests assert pipeline >= 0.8 x this figure)
        double link_sec = 0;
        {
            CommandEncoder up(device), down(device);
            PinnedArray hin(device, pipe.bytes_per_iteration()), hout(device, pipe.bytes_per_iteration());
            Buffer a(device, pipe.bytes_per_iteration()), b(device, pipe.bytes_per_iteration());
            std::memset(hin.data(), 0, hin.size());
            for (int rep = 0; rep < 2; ++rep) {   // first round: warm-up
                const auto t2 = std::chrono::steady_clock::now();
                for (int it = 0; it < iters; ++it) {
                    a.write(hin.data(), hin.size(), &up);
                    b.read_async(hout.data(), hout.size(), down);
                }
                up.synchronize();
                down.synchronize();
                link_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count();
            }
        }
        const double gb = (double)pipe.bytes_per_iteration() * iters / 1e9;
        std::printf("iterations %d slots %d samples checked %llu max abs error %.3g\n", iters, slots, (unsigned long long)checked, worst);
        std::printf("with host fill+check: %.1f iterations/s, %.2f GB/s each way\n", iters / sec, gb / sec);
        std::printf("pipeline only: %.1f iterations/s, %.2f GB/s each way, %.3f Gsamples/s PCIe-inclusive\n", iters / sec2, gb / sec2,
                    (double)samples * iters / sec2 / 1e9);
        std::printf("link only (both directions busy, no transform): %.2f GB/s each way\n", gb / link_sec);
        return (worst <= 1e-5 && checked == (uint64_t)iters * samples) ? 0 : 1;
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 2;
    }
}
