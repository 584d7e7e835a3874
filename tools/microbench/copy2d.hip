// Device -> host copies of part of an 8K RGBA8 frame: one 1D copy of whole rows against strips of hipMemcpy2DAsync that leave out the
// columns nothing was drawn in, into pageable and into page-locked host memory.  (rxr_render_download: is a column-trimmed download
// of a sparse frame worth its copies?)   build: hipcc --offload-arch=gfx950 -O2 -o /tmp/copy2d tools/microbench/copy2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t W = 7680, H = 4320, row0 = 1344, row1 = 4320;   // the content rows of the box grid
    uint8_t *d = nullptr;
    CK(hipMalloc(&d, W * H * 4));
    CK(hipMemset(d, 7, W * H * 4));
    uint8_t *pageable = (uint8_t *)malloc(W * H * 4), *pinned = nullptr;
    memset(pageable, 1, W * H * 4);
    CK(hipHostMalloc((void **)&pinned, W * H * 4, hipHostMallocDefault));
    memset(pinned, 1, W * H * 4);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int which = 0; which < 2; ++which) {
        uint8_t *dst = which ? pinned : pageable;
        const char *name = which ? "page-locked" : "pageable";
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = clk::now();
            CK(hipMemcpyAsync(dst + row0 * W * 4, d + row0 * W * 4, (row1 - row0) * W * 4, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            auto t1 = clk::now();
            if (rep) printf("%-12s 1D  rows %zu..%zu  %6.1f MB  %.3f ms  %.1f GB/s\n", name, row0, row1, (row1 - row0) * W * 4 / 1e6, ms(t0, t1), (row1 - row0) * W * 4 / 1e6 / ms(t0, t1));
        }
        for (int strips : {4, 24, 93}) {
            for (int rep = 0; rep < 3; ++rep) {
                size_t bytes = 0;
                auto t0 = clk::now();
                for (int k = 0; k < strips; ++k) {
                    const size_t a = row0 + (row1 - row0) * k / strips, b = row0 + (row1 - row0) * (k + 1) / strips;
                    // a trapezoid: 30 % of the width at the top of the content, 95 % at the bottom (tile columns)
                    const double f = 0.30 + 0.65 * (double)(b - row0) / (double)(row1 - row0);
                    const size_t wpx = ((size_t)(W * f) + 15) / 16 * 16, x0 = ((W - wpx) / 2) / 16 * 16;
                    CK(hipMemcpy2DAsync(dst + (a * W + x0) * 4, W * 4, d + (a * W + x0) * 4, W * 4, wpx * 4, b - a, hipMemcpyDeviceToHost, s));
                    bytes += wpx * 4 * (b - a);
                }
                CK(hipStreamSynchronize(s));
                auto t1 = clk::now();
                if (rep) printf("%-12s 2D  %2d strips      %6.1f MB  %.3f ms  %.1f GB/s\n", name, strips, bytes / 1e6, ms(t0, t1), bytes / 1e6 / ms(t0, t1));
            }
        }
    }
    return 0;
}
