// What does it cost to run a short pre-pass kernel of step i+1 on a second stream under the long kernel of step i?  Three ways to queue K steps
// of [A (one workgroup, ~5 us); B (a full grid, ~100 us), B needs A's output]:
//   0  one stream: A B A B ...
//   1  two streams, hipEventRecord + hipStreamWaitEvent (A(i+1) waits for B(i-1), B(i+1) waits for A(i+1))
//   2  two streams, the events are the kernels' own stop events (hipExtLaunchKernelGGL): no marker packets, only the waits
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_prepass overlap_prepass.hip ; run: ./overlap_prepass [a_iters b_iters]
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_a(float *out, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    out[threadIdx.x] = x;
}
__global__ void k_b(const float *in, float *out, int iters) {
    float x = in[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) out[blockIdx.x] = x;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int a_iters = argc > 1 ? atoi(argv[1]) : 4000, b_iters = argc > 2 ? atoi(argv[2]) : 2300, K = 400;
    float *rec[2], *out;
    CK(hipMalloc(&rec[0], 4096)); CK(hipMalloc(&rec[1], 4096)); CK(hipMalloc(&out, 1 << 20));
    hipStream_t s, aux;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
    std::vector<hipEvent_t> ea(K + 2), eb(K + 2);
    for (auto &e : ea) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : eb) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<hipEvent_t> ta(K + 2), tb(K + 2);  // mode 2 binds stop events to dispatches: timing-capable events
    for (auto &e : ta) CK(hipEventCreate(&e));
    for (auto &e : tb) CK(hipEventCreate(&e));
    const dim3 grid_b(32400), blk(256);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < K; ++i) {
                const int p = i & 1;
                if (mode == 0) {
                    hipLaunchKernelGGL(k_a, dim3(1), blk, 0, s, rec[0], a_iters);
                    hipLaunchKernelGGL(k_b, grid_b, blk, 0, s, rec[0], out, b_iters);
                } else if (mode == 1) {
                    if (i >= 2) CK(hipStreamWaitEvent(aux, eb[i - 2], 0));
                    hipLaunchKernelGGL(k_a, dim3(1), blk, 0, aux, rec[p], a_iters);
                    CK(hipEventRecord(ea[i], aux));
                    CK(hipStreamWaitEvent(s, ea[i], 0));
                    hipLaunchKernelGGL(k_b, grid_b, blk, 0, s, rec[p], out, b_iters);
                    CK(hipEventRecord(eb[i], s));
                } else {
                    if (i >= 2) CK(hipStreamWaitEvent(aux, tb[i - 2], 0));
                    hipExtLaunchKernelGGL(k_a, dim3(1), blk, 0, aux, nullptr, ta[i], 0, rec[p], a_iters);
                    CK(hipStreamWaitEvent(s, ta[i], 0));
                    hipExtLaunchKernelGGL(k_b, grid_b, blk, 0, s, nullptr, tb[i], 0, rec[p], out, b_iters);
                }
            }
            auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            auto t2 = std::chrono::steady_clock::now();
            printf("mode %d rep %d: %.2f us per step (host issue %.2f us per step)\n", mode, rep,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / K, std::chrono::duration<double, std::micro>(t1 - t0).count() / K);
        }
    }
    // the kernels alone
    hipEvent_t x0, x1;
    CK(hipEventCreate(&x0)); CK(hipEventCreate(&x1));
    float ms = 0;
    hipExtLaunchKernelGGL(k_a, dim3(1), blk, 0, s, x0, x1, 0, rec[0], a_iters);
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, x0, x1)); printf("A alone %.2f us\n", ms * 1e3);
    hipExtLaunchKernelGGL(k_b, grid_b, blk, 0, s, x0, x1, 0, rec[0], out, b_iters);
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, x0, x1)); printf("B alone %.2f us\n", ms * 1e3);
    return 0;
}
