// rxr_exact_math.h -- cheaper instruction sequences for IEEE-correct f32 division, square root and
// log2/exp2 on gfx950 that are BIT-IDENTICAL to what hipcc emits for `/`, sqrtf, log2f and exp2f.
//
// Why: the raster kernel is fp32-VALU bound and (measured, tools/microbench/valu_rates.hip) a
// correctly rounded division costs 36 SIMD cycles per wave (2 v_div_scale + v_rcp + 6 fma/mul +
// v_div_fmas + v_div_fixup), a square root 56, log2f/exp2f 26 each -- together about 40 % of the
// kernel's VALU time.  hipcc's expansions are written for every possible operand; most of their
// instructions only deal with operands near the ends of the exponent range:
//   * v_div_scale_f32 returns its operand unchanged and leaves VCC clear unless the denominator, its
//     reciprocal, the numerator or the quotient comes near the denormal range or the quotient near
//     FLT_MAX; v_div_fmas_f32 is then a plain fma and v_div_fixup_f32 returns its first operand
//     for finite non-zero operands.  What remains is rcp + 2 fma (functions of the denominator only)
//     and mul + 4 fma per numerator -- so divisions that share a denominator (normalisation, the
//     perspective divide) also share the reciprocal.
//   * sqrtf: v_sqrt_f32 followed by two one-ulp correction probes; the rest scales arguments below
//     2^-96 and passes 0 / inf through.  (sqrt_core below goes one step further: a shorter sequence that
//     is not the compiler's, proven equal by exhaustion over the whole window.)
//   * log2f / exp2f: v_log_f32 / v_exp_f32 plus scaling for arguments below 2^-126 / -126.
// Every helper here tests, for the whole wave, that all active lanes' operands are inside a window in
// which those extra instructions are provably no-ops, runs the short sequence if so and the
// compiler's own expansion otherwise (wave-uniform branch).  In the window the short sequence IS the
// compiler's sequence with the no-op instructions removed, so the result is the same float; outside
// the window it is the compiler's code.  tests/test_gpu_exact_math.py checks bit equality over
// billions of operand pairs, including the window edges and every special value.
//
// RXR_EXACT_FAST=0 compiles the plain operators instead (A/B measurements).
#pragma once
#ifndef RXR_JIT
#include <hip/hip_runtime.h>
#endif

#ifndef RXR_EXACT_FAST
#define RXR_EXACT_FAST 1
#endif

namespace rxm {

// operand window for divisions: with 2^-40 <= |n|, |d| <= 2^40 the quotient lies in [2^-80, 2^80],
// no v_div_scale_f32 case applies (exponent difference < 96, nothing denormal, numerator exponent > 23)
// and every intermediate of the chain (the smallest is the residual, >= 2^-24 * 2^-40 * 2^-24) is a
// normal number.
constexpr float WIN_LO = 0x1p-40f;
constexpr float WIN_HI = 0x1p40f;

__device__ __forceinline__ bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }

__device__ __forceinline__ bool in_window(float x) {
    float a = __builtin_fabsf(x);
    return a >= WIN_LO && a <= WIN_HI;  // false for NaN
}

// squares of window values, for magnitudes: 2^-80 <= x <= 2^80 by one unsigned compare on the bits
// (negative, NaN and inf fail)
__device__ __forceinline__ bool sq_in_window(float x) {
    return (__float_as_uint(x) - 0x17800000u /* 2^-80 */) <= (0x67800000u /* 2^80 */ - 0x17800000u);
}

// rcp + one Newton step: the part of the division chain that depends on the denominator only
__device__ __forceinline__ float rcp_refined(float d) {
    float r0 = __builtin_amdgcn_rcpf(d);
    float e0 = fmaf(-d, r0, 1.0f);
    return fmaf(e0, r0, r0);
}
// the per-numerator part: q0, residual, correction, residual, correction (= v_div_fmas with VCC clear)
__device__ __forceinline__ float div_chain(float n, float d, float r) {
    float q0 = n * r;
    float e1 = fmaf(-d, q0, n);
    float q1 = fmaf(e1, r, q0);
    float e2 = fmaf(-d, q1, n);
    return fmaf(e2, r, q1);
}

// sqrt core, valid for 2^-96 <= x < inf: y = v_rsq_f32(x), s = x*y corrected once with h = y/2 (19 issue cycles; hipcc's
// own expansion -- v_sqrt_f32 and two one-ulp probes -- costs 36 without its scaling prologue).  This is NOT the
// compiler's sequence with no-ops removed: it is a different sequence whose result was compared with sqrtf for EVERY
// float of the window on the device (tools/microbench/sqrt_variants.hip, all 1 879 048 192 operands, 0 differences;
// rxr_selftest_math re-checks a strided sample of the whole window in every test run).
__device__ __forceinline__ float sqrt_core(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}

// ---- public helpers (same value as the plain operator for EVERY input) --------------------------

// n / d for one numerator.  Only cheaper than `/` when the window test is cheap for the caller, so it
// takes the caller's knowledge: `ok` must imply in_window(n) && in_window(d) for this lane.
__device__ __forceinline__ float div1_known(float n, float d, bool ok) {
#if RXR_EXACT_FAST
    if (wave_all(ok)) return div_chain(n, d, rcp_refined(d));
#endif
    return n / d;
}

__device__ __forceinline__ void div2(float n0, float n1, float d, float &q0, float &q1) {
#if RXR_EXACT_FAST
    float lo = fminf(__builtin_fabsf(n0), __builtin_fabsf(n1)), hi = fmaxf(__builtin_fabsf(n0), __builtin_fabsf(n1));
    if (wave_all(in_window(d) && lo >= WIN_LO && hi <= WIN_HI)) {
        float r = rcp_refined(d);
        q0 = div_chain(n0, d, r);
        q1 = div_chain(n1, d, r);
        return;
    }
#endif
    q0 = n0 / d;
    q1 = n1 / d;
}

// div2 with the denominator's share of the work done once elsewhere (a triangle's area, by the thread that prepares its record for a
// row-mode round): r = denominator_part(d).  The same instructions on the same operands: the same quotients.
__device__ __forceinline__ float denominator_part(float d) { return in_window(d) ? rcp_refined(d) : 0.0f; }  // (never 0 inside the window)
__device__ __forceinline__ void div2_pre(float n0, float n1, float d, float r, float &q0, float &q1) {
#if RXR_EXACT_FAST
    float lo = fminf(__builtin_fabsf(n0), __builtin_fabsf(n1)), hi = fmaxf(__builtin_fabsf(n0), __builtin_fabsf(n1));
    if (wave_all(r != 0.0f && lo >= WIN_LO && hi <= WIN_HI)) {
        q0 = div_chain(n0, d, r);
        q1 = div_chain(n1, d, r);
        return;
    }
#endif
    q0 = n0 / d;
    q1 = n1 / d;
}

__device__ __forceinline__ void div3(float n0, float n1, float n2, float d, float &q0, float &q1, float &q2) {
#if RXR_EXACT_FAST
    float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(n0), __builtin_fabsf(n1)), __builtin_fabsf(n2));
    float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n0), __builtin_fabsf(n1)), __builtin_fabsf(n2));
    if (wave_all(in_window(d) && lo >= WIN_LO && hi <= WIN_HI)) {
        float r = rcp_refined(d);
        q0 = div_chain(n0, d, r);
        q1 = div_chain(n1, d, r);
        q2 = div_chain(n2, d, r);
        return;
    }
#endif
    q0 = n0 / d;
    q1 = n1 / d;
    q2 = n2 / d;
}

// (n0, n1, n2, d) / d: the perspective divide of a Vec4 by its own w (d / d is exactly 1 in the window)
__device__ __forceinline__ void div3_self(float n0, float n1, float n2, float d, float &q0, float &q1, float &q2, float &qd) {
#if RXR_EXACT_FAST
    float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(n0), __builtin_fabsf(n1)), __builtin_fabsf(n2));
    float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n0), __builtin_fabsf(n1)), __builtin_fabsf(n2));
    if (wave_all(in_window(d) && lo >= WIN_LO && hi <= WIN_HI)) {
        float r = rcp_refined(d);
        q0 = div_chain(n0, d, r);
        q1 = div_chain(n1, d, r);
        q2 = div_chain(n2, d, r);
        qd = 1.0f;
        return;
    }
#endif
    q0 = n0 / d;
    q1 = n1 / d;
    q2 = n2 / d;
    qd = d / d;
}

__device__ __forceinline__ float sqrt_exact(float x) {
#if RXR_EXACT_FAST
    // 2^-96 <= x < inf  (bits in [0x0f800000, 0x7f800000))
    if (wave_all((__float_as_uint(x) - 0x0f800000u) < (0x7f800000u - 0x0f800000u))) return sqrt_core(x);
#endif
    return sqrtf(x);
}

// v / |v| with |v| = sqrt((x*x + y*y) + z*z), vek's `normalized`; also returns the magnitude.
// One window test on the squared magnitude covers the square root and the denominator
// (2^-80 <= m2 <= 2^80  =>  2^-40 <= m <= 2^40); the numerators satisfy |n| <= m by monotonicity of
// rounding, so only their lower bound is tested.
__device__ __forceinline__ void normalize3(float x, float y, float z, float &ox, float &oy, float &oz, float &mag) {
    float m2 = (x * x + y * y) + z * z;
#if RXR_EXACT_FAST
    float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    if (wave_all(sq_in_window(m2) && lo >= WIN_LO)) {
        float m = sqrt_core(m2);
        float r = rcp_refined(m);
        ox = div_chain(x, m, r);
        oy = div_chain(y, m, r);
        oz = div_chain(z, m, r);
        mag = m;
        return;
    }
#endif
    float m = sqrtf(m2);
    ox = x / m;
    oy = y / m;
    oz = z / m;
    mag = m;
}

// RELAXED form for the light loop of lit 3D fragments (north_star: "within 1 per channel for float-interpolated 3D/lit paths"):
// v * rsq(|v|^2) and |v|^2 * rsq(|v|^2) -- every component and the magnitude within 2 ulp of the correctly rounded values, six
// instructions instead of twenty-five.  The same window vote as the exact form: zero, denormal, infinite and NaN magnitudes take
// the exact path, so the special cases of both forms are the same values.
__device__ __forceinline__ void normalize3_relaxed(float x, float y, float z, float &ox, float &oy, float &oz, float &mag) {
    float m2 = (x * x + y * y) + z * z;
    if (wave_all(sq_in_window(m2))) {
        float inv = __builtin_amdgcn_rsqf(m2);
        ox = x * inv;
        oy = y * inv;
        oz = z * inv;
        mag = m2 * inv;
        return;
    }
    float m = sqrtf(m2);
    ox = x / m;
    oy = y / m;
    oz = z / m;
    mag = m;
}

// The same for vectors that often have components that are exactly zero (axis-aligned surface normals): a zero
// numerator would lose its sign in the correction steps of the chain (fma(+0, r, -0) = +0), so each quotient goes
// through v_div_fixup_f32 -- the compiler's own last instruction, which returns a correctly signed zero for a zero
// numerator and its first operand otherwise.  Components must be exactly zero or at least 2^-40 in magnitude.
__device__ __forceinline__ uint32_t zero_or_window_key(float x) { return (__float_as_uint(x) & 0x7FFFFFFFu) - 1u; }  // 0 -> 0xFFFFFFFF
__device__ __forceinline__ void normalize3_z(float x, float y, float z, float &ox, float &oy, float &oz) {
    float m2 = (x * x + y * y) + z * z;
#if RXR_EXACT_FAST
    uint32_t k = min(min(zero_or_window_key(x), zero_or_window_key(y)), zero_or_window_key(z));
    if (wave_all(sq_in_window(m2) && k >= 0x2B800000u /* 2^-40 */ - 1u)) {
        float m = sqrt_core(m2);
        float r = rcp_refined(m);
        ox = __builtin_amdgcn_div_fixupf(div_chain(x, m, r), m, x);
        oy = __builtin_amdgcn_div_fixupf(div_chain(y, m, r), m, y);
        oz = __builtin_amdgcn_div_fixupf(div_chain(z, m, r), m, z);
        return;
    }
#endif
    float m = sqrtf(m2);
    ox = x / m;
    oy = y / m;
    oz = z / m;
}

// Rust's `x as u32` / `x as usize` / `x as u8` (saturating, NaN -> 0) in one instruction: v_cvt_u32_f32
// truncates toward zero, clamps to [0, 2^32 - 1] and turns NaN into 0 -- exactly the cast.  C's `(uint32_t)x` is undefined out
// of range, so the compiler may not be asked for it; the instruction is named directly.  Checked against the compare-and-
// select form over a strided sweep of ALL float bit patterns in every test run (rxr_selftest_math, kind 10).
// RXR_HW_SAT_CVT=0 compiles the compare-and-select form (A-B runs).
#ifndef RXR_HW_SAT_CVT
#define RXR_HW_SAT_CVT 1
#endif
__device__ __forceinline__ uint32_t sat_u32_ref(float x) {
    if (!(x > 0.0f)) return 0u;  // NaN, negatives, zero
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
__device__ __forceinline__ uint32_t sat_u32(float x) {
#if RXR_HW_SAT_CVT
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return sat_u32_ref(x);
#endif
}

// Maximum over the 64 lanes of a wave of NON-NEGATIVE floats (+0, denormals, normals, +inf: their order is the order of their bit
// patterns as unsigned integers), returned in a scalar register.  Six v_max_u32 with DPP operands -- inside each quad, across the
// quads of a row of 16, then rows 0 -> 1 and 2 -> 3 (row_bcast:15) and rows 0..1 -> 2..3 (row_bcast:31): lane 63 holds the wave's
// maximum -- and one v_readlane_b32, against six dependent ds_bpermute_b32 round trips through the LDS crossbar (with the index
// arithmetic and the canonicalising v_max_f32 pairs of fmaxf: 36 VALU instructions) for the __shfl_xor butterfly.  EVERY lane of the
// wave must be active.  A NaN or a negative operand makes the result meaningless, not undefined.  (Lanes a DPP pattern does not
// write read 0, the identity of the unsigned maximum: the compiler can then fold the move into v_max_u32_dpp.)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = (int)__builtin_bit_cast(unsigned int, v);
#define RXR_DPP_MAX_U32(ctrl, row_mask) \
    x = (int)__builtin_elementwise_max((unsigned int)x, (unsigned int)__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xf, false))
    RXR_DPP_MAX_U32(0xB1, 0xf);   // quad_perm:[1,0,3,2]
    RXR_DPP_MAX_U32(0x4E, 0xf);   // quad_perm:[2,3,0,1]
    RXR_DPP_MAX_U32(0x141, 0xf);  // row_half_mirror
    RXR_DPP_MAX_U32(0x140, 0xf);  // row_mirror: every lane of a row holds the row's maximum
    RXR_DPP_MAX_U32(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    RXR_DPP_MAX_U32(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef RXR_DPP_MAX_U32
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 63));
}

// Inclusive prefix sum of a 32-bit integer over the 64 lanes of a wave (EVERY lane active): four row_shr steps inside each row of 16
// (lanes the shift leaves without a source add 0), then row 0's total into row 1 and row 2's into row 3 (row_bcast:15), then the
// total of rows 0..1 into rows 2..3 (row_bcast:31) -- six v_add_u32 with DPP operands instead of the six dependent ds_bpermute_b32
// round trips of the __shfl_up ladder.
__device__ __forceinline__ unsigned int wave_inclusive_add(unsigned int v) {
    int x = (int)v;
#define RXR_DPP_ADD_U32(ctrl, row_mask) x += __builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xf, false)
    RXR_DPP_ADD_U32(0x111, 0xf);  // row_shr:1
    RXR_DPP_ADD_U32(0x112, 0xf);  // row_shr:2
    RXR_DPP_ADD_U32(0x114, 0xf);  // row_shr:4
    RXR_DPP_ADD_U32(0x118, 0xf);  // row_shr:8
    RXR_DPP_ADD_U32(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    RXR_DPP_ADD_U32(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef RXR_DPP_ADD_U32
    return (unsigned int)x;
}

// exp2f(k * log2f(x)) as the reference's pow32_fast computes it (rasterizer.rs:1895-1901)
__device__ __forceinline__ float pow_exp2_log2(float x, float k) {
#if RXR_EXACT_FAST
    // log2f scales arguments below 2^-126, exp2f arguments below -126; NaN takes neither branch
    float lg = __builtin_amdgcn_logf(x);
    float y = k * lg;
    if (wave_all(!(x < 0x1p-126f) && !(y < -126.0f))) return __builtin_amdgcn_exp2f(y);
#endif
    return exp2f(k * log2f(x));
}

}  // namespace rxm
