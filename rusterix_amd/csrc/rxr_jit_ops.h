// rxr_jit_ops.h -- the operations of the Rusteria interpreter (rxr_vm.h) as device function templates, one instantiation per
// opcode, for the straight-line code that rxr_jit.hip generates from a program set's jump code and compiles at run time.
//
// Every expression below is the expression of the interpreter's handler for that opcode, copied character for character (the
// handler is named by its `case`): a compiled program performs the same float operations in the same order as the interpreter,
// and both are checked against the CPU oracle and against each other (tests/test_gpu_shader_jit.py: every exact opcode, random
// programs, the configuration-C5 program).  The libm-backed opcodes call the interpreter's own out-of-line functions
// (slow_unary / slow_binary).  Included by rxr_vm.h in RXR_JIT mode only.
#pragma once

namespace rxvm {

template <uint32_t OP>
__device__ __forceinline__ v3 jit_un(v3 a) {
    if constexpr (OP == RXR_NODE_LENGTH) return splat(sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z));
    else if constexpr (OP == RXR_NODE_LENGTH2) return mk(sqrtf(a.x * a.x + a.y * a.y), 0.0f, 0.0f);
    else if constexpr (OP == RXR_NODE_LENGTH3) return mk(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), 0.0f, 0.0f);
    else if constexpr (OP == RXR_NODE_ABS) return mk(fabsf(a.x), fabsf(a.y), fabsf(a.z));
    else if constexpr (OP == RXR_NODE_SIN || OP == RXR_NODE_SIN1 || OP == RXR_NODE_SIN2 || OP == RXR_NODE_COS || OP == RXR_NODE_COS1 ||
                       OP == RXR_NODE_COS2 || OP == RXR_NODE_TAN || OP == RXR_NODE_ATAN || OP == RXR_NODE_LOG)
        return slow_unary(OP, a);
    else if constexpr (OP == RXR_NODE_NORMALIZE) {  // :345-353
        float len = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
        return len > 0.0f ? mk(a.x / len, a.y / len, a.z / len) : a;
    } else if constexpr (OP == RXR_NODE_FLOOR) return mk(floorf(a.x), floorf(a.y), floorf(a.z));
    else if constexpr (OP == RXR_NODE_CEIL) return mk(ceilf(a.x), ceilf(a.y), ceilf(a.z));
    else if constexpr (OP == RXR_NODE_ROUND) return mk(roundf(a.x), roundf(a.y), roundf(a.z));
    else if constexpr (OP == RXR_NODE_FRACT) return mk(a.x - floorf(a.x), a.y - floorf(a.y), a.z - floorf(a.z));
    else if constexpr (OP == RXR_NODE_RADIANS)
        return mk(a.x * (3.14159265358979323846f / 180.0f), a.y * (3.14159265358979323846f / 180.0f), a.z * (3.14159265358979323846f / 180.0f));
    else if constexpr (OP == RXR_NODE_DEGREES)
        return mk(a.x * 57.2957795130823208767981548141051703f, a.y * 57.2957795130823208767981548141051703f, a.z * 57.2957795130823208767981548141051703f);
    else if constexpr (OP == RXR_NODE_SQRT) return mk(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z));
    else if constexpr (OP == RXR_NODE_NOT) return splat(a.x == 0.0f ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_NEG) return mk(-a.x, -a.y, -a.z);
    else static_assert(OP != OP, "not a unary opcode");
}

template <uint32_t OP>
__device__ __forceinline__ v3 jit_bin(v3 a, v3 b) {
    if constexpr (OP == RXR_NODE_ADD) return mk(a.x + b.x, a.y + b.y, a.z + b.z);
    else if constexpr (OP == RXR_NODE_SUB) return mk(a.x - b.x, a.y - b.y, a.z - b.z);
    else if constexpr (OP == RXR_NODE_MUL) return mk(a.x * b.x, a.y * b.y, a.z * b.z);
    else if constexpr (OP == RXR_NODE_DIV) return mk(a.x / b.x, a.y / b.y, a.z / b.z);
    else if constexpr (OP == RXR_NODE_ATAN2 || OP == RXR_NODE_POW || OP == RXR_NODE_ROTATE2D) return slow_binary(OP, a, b);
    else if constexpr (OP == RXR_NODE_DOT) return splat((a.x * b.x + a.y * b.y) + a.z * b.z);
    else if constexpr (OP == RXR_NODE_DOT2) return mk(a.x * b.x + a.y * b.y, 0.0f, 0.0f);
    else if constexpr (OP == RXR_NODE_DOT3) return mk(a.x * b.x + a.y * b.y + a.z * b.z, 0.0f, 0.0f);
    else if constexpr (OP == RXR_NODE_CROSS) return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    else if constexpr (OP == RXR_NODE_MOD) return mk(a.x - b.x * floorf(a.x / b.x), a.y - b.y * floorf(a.y / b.y), a.z - b.z * floorf(a.z / b.z));
    else if constexpr (OP == RXR_NODE_MIN) return mk(rust_min(a.x, b.x), rust_min(a.y, b.y), rust_min(a.z, b.z));
    else if constexpr (OP == RXR_NODE_MAX) return mk(rust_max(a.x, b.x), rust_max(a.y, b.y), rust_max(a.z, b.z));
    else if constexpr (OP == RXR_NODE_STEP) return mk(b.x >= a.x ? 1.0f : 0.0f, b.y >= a.y ? 1.0f : 0.0f, b.z >= a.z ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_EQ) return splat(a.x == b.x ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_NE) return splat(a.x != b.x ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_LT) return splat(a.x < b.x ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_LE) return splat(a.x <= b.x ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_GT) return splat(a.x > b.x ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_GE) return splat(a.x >= b.x ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_AND) return splat(((a.x != 0.0f) & (b.x != 0.0f)) ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_OR) return splat(((a.x != 0.0f) | (b.x != 0.0f)) ? 1.0f : 0.0f);
    else if constexpr (OP == RXR_NODE_PACK2) return mk(a.x, b.x, 0.0f);
    else static_assert(OP != OP, "not a binary opcode");
}

template <uint32_t OP>
__device__ __forceinline__ v3 jit_ter(v3 a, v3 b, v3 c) {
    if constexpr (OP == RXR_NODE_PACK3) return mk(a.x, b.x, c.x);
    else if constexpr (OP == RXR_NODE_MIX) return mk(a.x + (b.x - a.x) * c.x, a.y + (b.y - a.y) * c.y, a.z + (b.z - a.z) * c.z);
    else if constexpr (OP == RXR_NODE_SMOOTHSTEP) {  // :456-474: a = edge0, b = edge1, c = x
        float denom = b.x - a.x;
        float t = denom != 0.0f ? (c.x - a.x) / denom : 0.0f;
        if (t < 0.0f) t = 0.0f;
        else if (t > 1.0f) t = 1.0f;
        return splat(t * t * (3.0f - 2.0f * t));
    } else if constexpr (OP == RXR_NODE_CLAMP)  // (the bounds have been checked: jit_clamp_ok)
        return mk(rclampf(a.x, b.x, c.x), rclampf(a.y, b.y, c.y), rclampf(a.z, b.z, c.z));
    else static_assert(OP != OP, "not a ternary opcode");
}
// f32::clamp panics unless min <= max: a = x, b = lo, c = hi
__device__ __forceinline__ bool jit_clamp_ok(v3 b, v3 c) { return (b.x <= c.x) && (b.y <= c.y) && (b.z <= c.z); }

// "Push c; op" fused by rxr_set_shaders (VM_BINC)
template <uint32_t WHICH>
__device__ __forceinline__ v3 jit_binc(v3 a, v3 b) {
    if constexpr (WHICH == VM_BINC_ADD) return mk(a.x + b.x, a.y + b.y, a.z + b.z);
    else if constexpr (WHICH == VM_BINC_SUB) return mk(a.x - b.x, a.y - b.y, a.z - b.z);
    else if constexpr (WHICH == VM_BINC_MUL) return mk(a.x * b.x, a.y * b.y, a.z * b.z);
    else if constexpr (WHICH == VM_BINC_DIV) return mk(a.x / b.x, a.y / b.y, a.z / b.z);
    else if constexpr (WHICH == VM_BINC_MIN) return mk(rust_min(a.x, b.x), rust_min(a.y, b.y), rust_min(a.z, b.z));
    else if constexpr (WHICH == VM_BINC_MAX) return mk(rust_max(a.x, b.x), rust_max(a.y, b.y), rust_max(a.z, b.z));
    else if constexpr (WHICH == VM_BINC_MOD) return mk(a.x - b.x * floorf(a.x / b.x), a.y - b.y * floorf(a.y / b.y), a.z - b.z * floorf(a.z / b.z));
    else if constexpr (WHICH == VM_BINC_LT) return splat(a.x < b.x ? 1.0f : 0.0f);
    else if constexpr (WHICH == VM_BINC_LE) return splat(a.x <= b.x ? 1.0f : 0.0f);
    else if constexpr (WHICH == VM_BINC_GT) return splat(a.x > b.x ? 1.0f : 0.0f);
    else if constexpr (WHICH == VM_BINC_GE) return splat(a.x >= b.x ? 1.0f : 0.0f);
    else if constexpr (WHICH == VM_BINC_EQ) return splat(a.x == b.x ? 1.0f : 0.0f);
    else return splat(a.x != b.x ? 1.0f : 0.0f);  // VM_BINC_NE (and, as in the interpreter, anything beyond it)
}

// VM_GETC, execution.rs:134-157 (`enc` is a compile-time constant at every call site: the loop folds)
__device__ __forceinline__ v3 jit_getc(uint32_t enc, v3 v) {
    const uint32_t n = enc & 15u;
    float r[3] = {0.0f, 0.0f, 0.0f};
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t c = (enc >> (4u + 2u * i)) & 3u;
        if (c == 3u) continue;
        float f = c == 0u ? v.x : (c == 1u ? v.y : v.z);
        if (k == 0u) r[0] = f;
        else if (k == 1u) r[1] = f;
        else if (k == 2u) r[2] = f;
        ++k;
    }
    return k == 1u ? splat(r[0]) : (k == 2u ? mk(r[0], r[1], 0.0f) : (k == 3u ? mk(r[0], r[1], r[2]) : splat(0.0f)));
}
// VM_SETC, :158-183
__device__ __forceinline__ v3 jit_setc(uint32_t enc, v3 target, v3 value) {
    const uint32_t n = enc & 15u;
    const uint32_t nc = (n >= 1u && n <= 3u) ? n : 0u;
    for (uint32_t i = 0; i < nc; ++i) {
        uint32_t c = (enc >> (4u + 2u * i)) & 3u;
        float f = i == 0u ? value.x : (i == 1u ? value.y : value.z);
        if (c == 0u) target.x = f;
        else if (c == 1u) target.y = f;
        else if (c == 2u) target.z = f;
    }
    return target;
}
// SetNormal: .normalized()
__device__ __forceinline__ v3 jit_set_normal(v3 a) {
    float len = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
    return mk(a.x / len, a.y / len, a.z / len);
}
// Sample / SampleNormal, :570-594: a = uv, b = pattern id
__device__ __forceinline__ v3 jit_sample(const RasterParams &P, v3 a, v3 b) {
    const uint32_t id = as_usize_sat(b.x);
    return id < P.n_patterns ? pattern_sample(P, P.patterns[id], a) : splat(0.0f);
}
__device__ __forceinline__ v3 jit_sample_normal(const RasterParams &P, v3 a, v3 b) {
    const uint32_t id = as_usize_sat(b.x);
    v3 o = splat(0.0f);
    if (id < P.n_normal_patterns) {
        v3 nm = pattern_sample(P, P.patterns[P.n_patterns + id], a);
        o = mk(nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, nm.z * 2.0f - 1.0f);
    }
    return o;
}

}  // namespace rxvm
