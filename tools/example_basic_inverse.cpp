// C++ replay of the asserting test of reference src/examples/basic_inverse.rs (test_ifft :130-258) through
// include/fft_wgpu.hpp: Inverse (1/n fused), n = 512, 1.28 M samples of 2 + 42i (:160), result copied to a staging buffer
// and read back (:184-205); expected: the constant at bin 0 of every transform, 0 elsewhere, max |error| < 1e-5 (:238-253 --
// the reference compares with rustfft's inverse / 512, which is exactly that for a constant input).
// Build: g++ -std=c++17 -Iinclude tools/example_basic_inverse.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -pthread
#include <cmath>
#include <cstdio>
#include <vector>

#include "fft_wgpu.hpp"

int main()
{
    using namespace fft_wgpu;
    try {
        Device device(0);
        const Queue &queue = device;
        const uint32_t n = 512;
        std::vector<Complex> data(512 * 500 * 5, Complex{2.0f, 42.0f});  // basic_inverse.rs:160
        std::vector<Complex> ans(data.size());
        const uint64_t bytes = data.size() * sizeof(Complex);
        Buffer staging(device, bytes), src(device, bytes);           // :163-177
        Inverse fft_inverse(device, queue, src, n);                  // :178
        CommandEncoder encoder(device);
        src.write(data.data(), bytes, &encoder);                     // queue.write_buffer :181
        Buffer &output = fft_inverse.proc(encoder);                  // :186 -- log2 512 is odd: the plan's second buffer
        if (&output == &src) { std::fprintf(stderr, "result rule: n = 512 must land in the second buffer (processor.rs:335-339)\n"); return 1; }
        output.copy_to(staging, bytes, encoder);                     // copy_buffer_to_buffer :189-195
        staging.read(ans.data(), bytes, &encoder);                   // submit + map_async + poll + get_mapped_range :196-205
        float max_error = 0.f;
        for (size_t i = 0; i < ans.size(); ++i) {
            const float er = (i % n == 0) ? 2.0f : 0.f, ei = (i % n == 0) ? 42.0f : 0.f;
            max_error = std::fmax(max_error, std::fmax(std::fabs(ans[i].real - er), std::fabs(ans[i].imag - ei)));
        }
        std::printf("max error %g\n", max_error);
        return max_error < 1e-5f ? 0 : 1;
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 2;
    }
}
