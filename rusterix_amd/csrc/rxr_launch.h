// rxr_launch.h -- every kernel launch of the library goes through RXR_LAUNCH.
//
// Kernel timing (rxr_profile_begin, include/rxr.h) without event records on the stream: while a render of the calling thread has a
// collector installed, a kernel is launched through hipExtLaunchKernelGGL with a (start, stop) event pair of its own, which the
// runtime fills with the DISPATCH's begin / end timestamps -- the figures a profiler's kernel trace reports.  hipEventRecord, by
// contrast, is a barrier packet of its own: the time between two recorded events holds the kernel's launch latency, ramp and drain
// on an otherwise idle GPU (measured in round 3: set-up 8.6 us against 5.3 us in the rocprofv3 trace, the raster kernel 112.7 against
// 109.3), and the records themselves idle the GPU for microseconds per frame.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#define RXR_PROF_MAX_KERNELS 16u
struct LaunchTimes {
    hipEvent_t ev[2 * RXR_PROF_MAX_KERNELS];  // (start, stop) per launched kernel, created when first needed
    uint32_t n;                               // kernels launched into this slot
    uint32_t raster_first;                    // launches [raster_first, n) are the raster kernel, those before it the set-up kernels
};
extern thread_local LaunchTimes *rxr_launch_times;  // (rxr_api.hip) the slot of the render this thread is queueing, or null

static inline bool rxr_launch_pair(hipEvent_t *a, hipEvent_t *b) {
    LaunchTimes *t = rxr_launch_times;
    if (!t || t->n >= RXR_PROF_MAX_KERNELS) return false;
    hipEvent_t *e = &t->ev[2u * t->n];
    if (!e[0] && hipEventCreate(&e[0]) != hipSuccess) return false;
    if (!e[1] && hipEventCreate(&e[1]) != hipSuccess) return false;
    *a = e[0];
    *b = e[1];
    t->n++;
    return true;
}
#define RXR_LAUNCH(kernel, grid, block, stream, ...)                                                                  \
    do {                                                                                                              \
        hipEvent_t rxr_e0_, rxr_e1_;                                                                                  \
        if (rxr_launch_times && rxr_launch_pair(&rxr_e0_, &rxr_e1_))                                                  \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, rxr_e0_, rxr_e1_, 0, __VA_ARGS__);                  \
        else                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                          \
    } while (0)
