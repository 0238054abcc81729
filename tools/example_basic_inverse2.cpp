// C++ replay of reference src/examples/basic_inverse2.rs (main :3-137 and test_ifft :139-286) through
// include/fft_wgpu.hpp.  Build: g++ -std=c++17 -Iinclude tools/example_basic_inverse2.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -pthread
#include <cmath>
#include <cstdio>
#include <new>
#include <vector>

#include "fft_wgpu.hpp"

int main()
{
    using namespace fft_wgpu;
    try {
        Device device(0);
        const Queue &queue = device;
        const uint32_t n = 512;
        std::vector<Complex> data(512 * 500 * 5, Complex{2.1327392395f, 3.033729f});  // basic_inverse2.rs:169
        std::vector<Complex> ans(data.size());
        const uint64_t bytes = data.size() * sizeof(Complex);
        Buffer src(device, bytes), src2(device, bytes);
        Onlyinverse fft_onlyinverse(device, queue, src, src2, n);   // :206
        Normalize normalize(device, queue, src, src2, n);           // :208
        CommandEncoder encoder(device);
        src.write(data.data(), bytes, &encoder);                    // queue.write_buffer :212
        fft_onlyinverse.proc(encoder);                              // :216
        Buffer &out = normalize.proc(encoder);                      // :218
        out.read(ans.data(), bytes, &encoder);                      // copy to staging + map :219-238
        float max_error = 0.f;                                      // :269-284 (expected: c at bin 0, 0 elsewhere)
        for (size_t i = 0; i < ans.size(); ++i) {
            const float er = (i % n == 0) ? 2.1327392395f : 0.f, ei = (i % n == 0) ? 3.033729f : 0.f;
            max_error = std::fmax(max_error, std::fmax(std::fabs(ans[i].real - er), std::fabs(ans[i].imag - ei)));
        }
        std::printf("max error %g\n", max_error);
        return max_error < 1e-5f ? 0 : 1;
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 2;
    }
}
