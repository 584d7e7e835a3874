// Exhaustive search for a cheaper instruction sequence that is BIT-IDENTICAL to hipcc's sqrtf on gfx950:
// every candidate is compared with sqrtf(x) for every float x in [2^-96, inf) (the window of
// rxm::sqrt_exact, rusterix_amd/csrc/rxr_exact_math.h) -- 2^31 - 2^27.x values, a fraction of a second.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/sqrt_variants tools/microbench/sqrt_variants.hip
// Output: per candidate the number of differing results and the first differing operand.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define NV 8

// A: the sequence in use (v_sqrt_f32 + two one-ulp probes): 36 issue cycles
__device__ __forceinline__ float cand_a(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    float r_dn = fmaf(-s_dn, s, x);
    float r_up = fmaf(-s_up, s, x);
    float t = (r_dn <= 0.0f) ? s_dn : s;
    return (r_up > 0.0f) ? s_up : t;
}
// B: rsq + one Newton correction of s = x * y with h = y / 2: 19 cycles
__device__ __forceinline__ float cand_b(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
// C: the Goldschmidt form LLVM uses when denormals are flushed: 27 cycles
__device__ __forceinline__ float cand_c(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float e = fmaf(-h, s, 0.5f);
    h = fmaf(h, e, h);
    s = fmaf(s, e, s);
    float d = fmaf(-s, s, x);
    return fmaf(d, h, s);
}
// D: B with a second correction: 24 cycles
__device__ __forceinline__ float cand_d(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float r = fmaf(-s, s, x);
    s = fmaf(r, h, s);
    r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
// E: v_sqrt_f32 corrected once with h from v_rsq_f32: 27 cycles
__device__ __forceinline__ float cand_e(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
// F: C with the refined h only (s corrected once by the refined h)
__device__ __forceinline__ float cand_f(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float e = fmaf(-h, s, 0.5f);
    h = fmaf(h, e, h);
    float d = fmaf(-s, s, x);
    return fmaf(d, h, s);
}

// G, H: controls that MUST show mismatches (raw v_sqrt_f32 is 1 ulp, so is x * rsq(x))
__device__ __forceinline__ float cand_g(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float cand_h(float x) { return x * __builtin_amdgcn_rsqf(x); }

__global__ void __launch_bounds__(256) k_check(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first) {
    unsigned long long cnt[NV] = {};
    for (uint64_t b = (uint64_t)lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
        float x = __uint_as_float((uint32_t)b);
        uint32_t ref = __float_as_uint(sqrtf(x));
        float v[NV] = {cand_a(x), cand_b(x), cand_c(x), cand_d(x), cand_e(x), cand_f(x), cand_g(x), cand_h(x)};
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (__float_as_uint(v[k]) != ref) {
                if (cnt[k]++ == 0) atomicMin(&first[k], (uint32_t)b);
            }
    }
    for (int k = 0; k < NV; ++k)
        if (cnt[k]) atomicAdd(&bad[k], cnt[k]);
}

int main() {
    unsigned long long *bad;
    uint32_t *first;
    hipMalloc(&bad, NV * sizeof(*bad));
    hipMalloc(&first, NV * sizeof(*first));
    hipMemset(bad, 0, NV * sizeof(*bad));
    hipMemset(first, 0xFF, NV * sizeof(*first));
    const uint32_t lo = 0x0f800000u, hi = 0x7f800000u;  // 2^-96 .. inf (exclusive)
    hipLaunchKernelGGL(k_check, dim3(256 * 64), dim3(256), 0, 0, lo, hi, bad, first);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed\n");
        return 1;
    }
    unsigned long long hb[NV];
    uint32_t hf[NV];
    hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost);
    const char *names[NV] = {"A sqrt+probes (in use)", "B rsq+1 step", "C rsq Goldschmidt (LLVM)", "D rsq+2 steps", "E sqrt+rsq step", "F rsq, refined h", "G raw v_sqrt (control)", "H x*rsq (control)"};
    printf("# operands 0x%08x .. 0x%08x (%llu values)\n", lo, hi, (unsigned long long)(hi - lo));
    for (int k = 0; k < NV; ++k) printf("%-28s mismatches %llu first 0x%08x\n", names[k], hb[k], hf[k]);
    return 0;
}
