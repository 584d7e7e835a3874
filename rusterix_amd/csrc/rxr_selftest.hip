// rxr_selftest.hip -- on-device self-test of rxr_exact_math.h: every short sequence against the
// compiler's expansion of the plain operator, bit for bit, over seeded operand tuples.
// Diagnostics only (rxr_selftest_math in include/rxr.h); not on the frame path.
#include <hip/hip_runtime.h>

#include "../../include/rxr.h"
#include "rxr_exact_math.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rng {
    uint64_t s;
    __device__ uint32_t next() {
        s = mix64(s);
        return (uint32_t)(s >> 32);
    }
};

// operand generators; `mode` is the same for all lanes of a wave so that whole waves stay inside the
// window (short sequence runs) or leave it (compiler's sequence runs)
//   0: exponents well inside the division window      1: the two ends of the window, inside
//   2: the two ends of the window, both sides         3: raw bits and special values
__device__ float gen(Rng &g, uint32_t mode, int win_lo_exp, int win_hi_exp) {
    uint32_t r = g.next(), m = g.next() & 0x7FFFFFu, sign = r & 0x80000000u;
    uint32_t pick = (r >> 8) & 7u;
    if (pick == 0) m = 0;
    if (pick == 1) m = 0x7FFFFFu;
    if (pick == 2) m &= 0x3u;
    int e;
    if (mode == 0) {
        int span = win_hi_exp - win_lo_exp - 3;
        e = win_lo_exp + 2 + (int)((r >> 12) % (uint32_t)span);
    } else if (mode == 1) {
        uint32_t k = (r >> 12) % 6u;
        e = k < 3 ? win_lo_exp + (int)k : win_hi_exp - 1 - (int)(k - 3);  // hi_exp itself only with m == 0
        if (((r >> 20) & 15u) == 0) { e = win_hi_exp; m = 0; }
    } else if (mode == 2) {
        uint32_t k = (r >> 12) % 12u;
        e = k < 6 ? win_lo_exp - 3 + (int)k : win_hi_exp - 3 + (int)(k - 6);
    } else {
        uint32_t k = (r >> 12) & 15u;
        if (k == 0) return __uint_as_float(sign);                       // +-0
        if (k == 1) return __uint_as_float(sign | 0x7F800000u);         // +-inf
        if (k == 2) return __uint_as_float(sign | 0x7FC00000u | m);     // NaN
        if (k == 3) return __uint_as_float(sign | (m | 1u));            // denormal
        if (k == 4) return __uint_as_float(sign | 0x7F7FFFFFu);         // FLT_MAX
        if (k == 5) return __uint_as_float(sign | 0x00800000u | m);     // smallest normals
        return __uint_as_float(g.next());
    }
    return __uint_as_float(sign | ((uint32_t)(e + 127) << 23) | m);
}

__device__ __forceinline__ bool same(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

__global__ void __launch_bounds__(256) k_selftest_math(uint64_t seed, uint32_t iters, unsigned long long *mismatch) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = gid >> 6;
    Rng g{mix64(seed ^ ((uint64_t)gid << 20))};
    unsigned long long bad[RXR_MATH_KINDS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t mode = (wave + it) & 3u;
        // ---- divisions sharing a denominator
        float n0 = gen(g, mode, -40, 40), n1 = gen(g, mode, -40, 40), n2 = gen(g, mode, -40, 40), d = gen(g, mode, -40, 40);
        {
            float a, b;
            rxm::div2(n0, n1, d, a, b);
            if (!same(a, n0 / d) || !same(b, n1 / d)) bad[0]++;
            rxm::div2_pre(n0, n1, d, rxm::denominator_part(d), a, b);  // (the denominator's part precomputed: row mode)
            if (!same(a, n0 / d) || !same(b, n1 / d)) bad[0]++;
        }
        {
            float a, b, c;
            rxm::div3(n0, n1, n2, d, a, b, c);
            if (!same(a, n0 / d) || !same(b, n1 / d) || !same(c, n2 / d)) bad[1]++;
        }
        {
            float a, b, c, e;
            rxm::div3_self(n0, n1, n2, d, a, b, c, e);
            if (!same(a, n0 / d) || !same(b, n1 / d) || !same(c, n2 / d) || !same(e, d / d)) bad[2]++;
        }
        // ---- normalisation: components up to 2^38 keep the squared magnitude inside its window in mode 0
        {
            float x = gen(g, mode, -40, 39), y = gen(g, mode, -40, 39), z = gen(g, mode, -40, 39);
            if (mode == 0 && (it & 1u)) {  // comparable magnitudes: the common case
                y = x * (0.25f + (float)(g.next() & 1023u) / 256.0f);
                z = x * (0.25f + (float)(g.next() & 1023u) / 256.0f);
            }
            float ox, oy, oz, mg;
            rxm::normalize3(x, y, z, ox, oy, oz, mg);
            float m = sqrtf((x * x + y * y) + z * z);
            if (!same(mg, m) || !same(ox, x / m) || !same(oy, y / m) || !same(oz, z / m)) bad[3]++;
        }
        // ---- square root: window 2^-96 .. inf
        {
            float x = fabsf(gen(g, mode, -96, 127));
            if (mode == 3 && (it & 3u) == 0) x = -x;
            if (!same(rxm::sqrt_exact(x), sqrtf(x))) bad[4]++;
        }
        // ---- pow through exp2(k * log2(x)): x in (0, 1] mostly, as n.h is
        {
            float x = fabsf(gen(g, mode, -126, 1));
            if (mode == 0) x = (float)(g.next() >> 8) / 16777216.0f;
            float k = (it & 1u) ? 6.0f : gen(g, 0, -2, 11);
            if (!same(rxm::pow_exp2_log2(x, k), exp2f(k * log2f(x)))) bad[5]++;
        }
        // ---- one numerator, caller-supplied knowledge
        {
            bool ok = rxm::in_window(n0) && rxm::in_window(d);
            if (!same(rxm::div1_known(n0, d, ok), n0 / d)) bad[6]++;
        }
        // ---- the statically-known cases of the kernel: pixel centre / frame size, byte / 255
        {
            float px = (float)(g.next() & 32767u) + 0.5f, w = (float)((g.next() & 32767u) + 1u);
            float byte = (float)(g.next() & 255u);
            if (!same(rxm::div1_known(px, w, true), px / w) || !same(rxm::div1_known(byte, 255.0f, true), byte / 255.0f)) bad[7]++;
        }
        // ---- normalisation of vectors with exactly-zero components (axis-aligned normals); every pattern of zeros,
        //      both signs of zero, and (modes 2, 3) tiny non-zero components that must take the compiler's path
        {
            float x = gen(g, mode, -40, 39), y = gen(g, mode, -40, 39), z = gen(g, mode, -40, 39);
            const uint32_t zr = g.next();
            if (zr & 1u) x = __uint_as_float(zr & 0x80000000u);
            if (zr & 2u) y = __uint_as_float((zr << 1) & 0x80000000u);
            if (zr & 4u) z = __uint_as_float((zr << 2) & 0x80000000u);
            if (mode == 0 && (zr & 8u)) {  // unit-length-ish normals
                x = (float)((int)(g.next() & 2047u) - 1024) / 1024.0f;
                y = (zr & 16u) ? 0.0f : (float)((int)(g.next() & 2047u) - 1024) / 1024.0f;
                z = (zr & 32u) ? -0.0f : (float)((int)(g.next() & 2047u) - 1024) / 1024.0f;
            }
            float ox, oy, oz;
            rxm::normalize3_z(x, y, z, ox, oy, oz);
            float m = sqrtf((x * x + y * y) + z * z);
            if (!same(ox, x / m) || !same(oy, y / m) || !same(oz, z / m)) bad[8]++;
        }
        // ---- square root: strided sweep of the whole window [2^-96, inf) -- the short sequence is not the compiler's
        //      (tools/microbench/sqrt_variants.hip checks every operand; this re-checks a sample in every run)
        {
            const uint64_t span = 0x7f800000ull - 0x0f800000ull;
            const uint64_t idx = (uint64_t)gid * iters + it;
            const uint32_t bits = 0x0f800000u + (uint32_t)((idx * 2654435761ull + (seed & 0xFFFFFu)) % span);
            const float x = __uint_as_float(bits);
            if (!same(rxm::sqrt_exact(x), sqrtf(x))) bad[9]++;
        }
        // ---- the saturating float -> u32 conversion (Rust's `as` casts) named as one instruction: a strided sweep of ALL bit
        //      patterns (every exponent, both signs, NaNs, infinities, denormals), plus the neighbourhoods of the clamp points
        {
            const uint64_t idx = (uint64_t)gid * iters + it;
            uint32_t bits = (uint32_t)((idx * 2654435761ull + (seed & 0xFFFFFFu)) & 0xFFFFFFFFull);
            if ((it & 7u) == 0u) {
                const uint32_t anchors[8] = {0x00000000u, 0x80000000u, 0x3F800000u, 0x437F0000u /* 255 */, 0x4F800000u /* 2^32 */, 0x4F7FFFFFu, 0x7F800000u, 0x7FC00000u};
                bits = anchors[(it >> 3) & 7u] + ((g.next() & 15u) - 8u);
            }
            const float x = __uint_as_float(bits);
            if (rxm::sat_u32(x) != rxm::sat_u32_ref(x)) bad[10]++;
        }
    }
    // ---- the DPP wave maximum of non-negative floats against the shuffle butterfly (outside the loop above: every lane of the wave
    //      must be active, and a lane's loop may end early in no build of this kernel -- but the reduction must not depend on that).
    //      Magnitudes of every exponent, +0, denormals, +inf; the maximum placed in every lane position over the tuples.
    for (uint32_t it = 0; it < (iters < 64u ? iters : 64u); ++it) {
        const uint32_t mode = (wave + it) & 3u;
        float x = __builtin_fabsf(gen(g, mode, -120, 120));
        if (x != x) x = __builtin_huge_valf();
        if ((threadIdx.x & 63u) == ((it * 7u + wave) & 63u)) x = x * 4.0f + 1.0f;   // (a likely winner that walks through the lanes)
        float ref = x;
        for (int d = 32; d >= 1; d >>= 1) ref = fmaxf(ref, __shfl_xor(ref, d, 64));
        if (!same(rxm::wave_max_nonneg(x), ref)) bad[11]++;
        // the DPP inclusive prefix sum against the shuffle ladder: small counts (as the row-mode areas are), zeros, and full 32-bit words
        // (the sum wraps the same way in both)
        uint32_t c = g.next();
        if (mode == 0) c &= 0xFFu;
        else if (mode == 1) c = (c & 1u) ? 0u : (c >> 20);
        uint32_t inc = c;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if ((threadIdx.x & 63u) >= d) inc += o;
        }
        if (rxm::wave_inclusive_add(c) != inc) bad[12]++;
    }
    for (int k = 0; k < RXR_MATH_KINDS; ++k)
        if (bad[k]) atomicAdd(&mismatch[k], bad[k]);
}

}  // namespace

// runs `blocks` x 256 threads x `iters` tuples per operation kind; mismatch must point at 8 zeroed
// device words
extern "C" void rxr_launch_selftest_math(uint64_t seed, uint32_t blocks, uint32_t iters, unsigned long long *mismatch, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_math, dim3(blocks), dim3(256), 0, s, seed, iters, mismatch);
}
