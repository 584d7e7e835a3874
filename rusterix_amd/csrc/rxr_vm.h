// rxr_vm.h -- the Rusteria stack VM on the device: Execution::shade
// (rusteria/src/node/execution.rs:741-749) and Execution::execute (:109-727) over the jump code that
// rxr_set_shaders flattens the reference's NodeOp trees into (rxr_device.h, VM_*).
//
// One invocation per fragment, state private to the lane (stack, locals, globals, frames live in scratch
// memory: dynamically indexed).  The reference keeps ONE Execution per tile and lets state leak from one
// fragment into the next; rxr_set_shaders only accepts programs for which that cannot matter (see
// include/rxr.h), and every invocation here starts from Execution::new's values plus the fields the raster
// loops assign before each call.
//
// Where the reference panics the lane stops and reports a VMF_* code through RasterParams.vm_fault;
// rxr_synchronize turns it into RXR_ERR_INVALID.  Loops are bounded by RXR_VM_MAX_STEPS instructions per
// invocation so that every wave reaches the end of the kernel.
//
// sin / cos / tan / atan / atan2 / pow / ln come from the device math library and differ from the host's
// libm by a few ulp: programs that use them are compared at +-1 per 8-bit channel.
#pragma once
#include <hip/hip_runtime.h>

#include "rxr_device.h"

namespace rxvm {

struct v3 {
    float x, y, z;
};
__device__ __forceinline__ v3 mk(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 splat(float x) { return v3{x, x, x}; }

// the Execution fields the raster loops exchange with a program (execution.rs:28-56)
struct IO {
    v3 uv, color, roughness, metallic, emissive, opacity, bump, normal, hitpoint, time;
};
__device__ __forceinline__ void io_defaults(IO &io) {  // Execution::new, :59-79
    io.uv = io.color = io.metallic = io.emissive = io.opacity = io.bump = io.normal = io.hitpoint = io.time = splat(0.0f);
    io.roughness = splat(0.5f);
}

__device__ __forceinline__ uint32_t as_usize_sat(float x) {  // `x as usize`, clamped to 32 bits (only compared with small counts)
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
__device__ __forceinline__ int floor_as_i32(float x) {  // `x.floor() as i32`
    float f = floorf(x);
    if (!(f == f)) return 0;
    if (f <= -2147483648.0f) return (int)0x80000000;
    if (f >= 2147483648.0f) return 0x7FFFFFFF;
    return (int)f;
}
__device__ __forceinline__ int rem_i32(int a, int m) {
    int r = a % m;
    return r < 0 ? r + m : r;
}
// TexStorage::sample (rusteria/src/textures/mod.rs:28-31, :125-141)
__device__ __forceinline__ v3 pattern_sample(const RasterParams &P, const DevPattern &t, v3 uv) {
    float u = uv.x, v = uv.y;
    u = u - floorf(u);
    v = v - floorf(v);
    int x = floor_as_i32(u * (float)t.w);
    int y = floor_as_i32(v * (float)t.h);
    x = rem_i32(x, (int)t.w);
    y = rem_i32(y, (int)t.h);
    const float *p = P.pattern_data + t.offset + 3u * ((size_t)y * t.w + (size_t)x);
    return mk(p[0], p[1], p[2]);
}
__device__ __forceinline__ float rclampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

#define VM_FAIL(code)  \
    do {               \
        fault = (code); \
        goto done;     \
    } while (0)
#define VM_POP(dst)                                  \
    do {                                             \
        if (sp == 0u) VM_FAIL(VMF_STACK_UNDERFLOW);  \
        dst = stack[--sp];                           \
    } while (0)
#define VM_PUSH(val)                                            \
    do {                                                        \
        if (sp >= RXR_VM_STACK) VM_FAIL(VMF_STACK_OVERFLOW);    \
        stack[sp++] = (val);                                    \
    } while (0)
#define VM_UN(expr)      \
    {                    \
        v3 a;            \
        VM_POP(a);       \
        VM_PUSH(expr);   \
        break;           \
    }
#define VM_BIN(expr)     \
    {                    \
        v3 a, b;         \
        VM_POP(b);       \
        VM_POP(a);       \
        VM_PUSH(expr);   \
        break;           \
    }

// Execution::shade on program `pi`.  Returns 0 or a VMF_* code.
__device__ __noinline__ uint32_t shade(const RasterParams &P, uint32_t pi, IO &io) {
    const DevProgram prog = P.programs[pi];
    if (prog.shade_entry == 0xFFFFFFFFu) return 0u;  // shade_index None: nothing runs (:1291, :771, :1651)
    const uint32_t *code = P.vm_code;
    v3 stack[RXR_VM_STACK];
    v3 locals[RXR_VM_LOCALS];
    v3 globals[RXR_VM_GLOBALS];
    uint32_t fr_pc[RXR_VM_FRAMES], fr_base[RXR_VM_FRAMES], fr_lbase[RXR_VM_FRAMES], fr_llen[RXR_VM_FRAMES];
    uint32_t loop_base[RXR_VM_LOOPS];
    uint32_t sp = 0, nframes = 0, nloops = 0, lbase = 0, llen = prog.shade_locals, pc = prog.shade_entry, fault = 0;
    bool has_ret = false;
    v3 ret = splat(0.0f);
    if (llen > RXR_VM_LOCALS) return VMF_LOCALS_OVERFLOW;
    for (uint32_t i = 0; i < llen; ++i) locals[i] = splat(0.0f);
    for (uint32_t i = 0; i < RXR_VM_GLOBALS; ++i) globals[i] = splat(0.0f);

    for (uint32_t step = 0;; ++step) {
        if (step >= RXR_VM_MAX_STEPS) VM_FAIL(VMF_STEP_LIMIT);
        const uint32_t w = code[pc++];
        switch (w & 0xFFu) {
            case RXR_NODE_LOAD_GLOBAL: {
                uint32_t i = code[pc++];
                if (i >= prog.n_globals) VM_FAIL(VMF_GLOBAL_INDEX);
                VM_PUSH(globals[i]);
                break;
            }
            case RXR_NODE_STORE_GLOBAL: {
                uint32_t i = code[pc++];
                if (i >= prog.n_globals) VM_FAIL(VMF_GLOBAL_INDEX);
                VM_POP(globals[i]);
                break;
            }
            case RXR_NODE_LOAD_LOCAL: {
                uint32_t i = code[pc++];
                if (i >= llen) VM_FAIL(VMF_LOCAL_INDEX);
                VM_PUSH(locals[lbase + i]);
                break;
            }
            case RXR_NODE_STORE_LOCAL: {
                uint32_t i = code[pc++];
                if (i >= llen) VM_FAIL(VMF_LOCAL_INDEX);
                VM_POP(locals[lbase + i]);
                break;
            }
            case RXR_NODE_SWAP: {
                v3 a, b;
                VM_POP(b);
                VM_POP(a);
                VM_PUSH(b);
                VM_PUSH(a);
                break;
            }
            case VM_GETC: {  // execution.rs:134-157
                uint32_t enc = code[pc++], n = enc & 15u, k = 0;
                v3 v;
                VM_POP(v);
                float r[3] = {0.0f, 0.0f, 0.0f};
                for (uint32_t i = 0; i < n; ++i) {
                    uint32_t c = (enc >> (4u + 2u * i)) & 3u;
                    if (c == 3u) continue;
                    float f = c == 0u ? v.x : (c == 1u ? v.y : v.z);
                    if (k < 3u) r[k] = f;
                    ++k;
                }
                v3 o = k == 1u ? splat(r[0]) : (k == 2u ? mk(r[0], r[1], 0.0f) : (k == 3u ? mk(r[0], r[1], r[2]) : splat(0.0f)));
                VM_PUSH(o);
                break;
            }
            case VM_SETC: {  // :158-183
                uint32_t enc = code[pc++], n = enc & 15u;
                v3 value, target;
                VM_POP(value);
                VM_POP(target);
                const uint32_t nc = (n >= 1u && n <= 3u) ? n : 0u;
                for (uint32_t i = 0; i < nc; ++i) {
                    uint32_t c = (enc >> (4u + 2u * i)) & 3u;
                    float f = i == 0u ? value.x : (i == 1u ? value.y : value.z);
                    if (c == 0u) target.x = f;
                    else if (c == 1u) target.y = f;
                    else if (c == 2u) target.z = f;
                }
                VM_PUSH(target);
                break;
            }
            case RXR_NODE_PUSH: {
                v3 v = mk(__uint_as_float(code[pc]), __uint_as_float(code[pc + 1]), __uint_as_float(code[pc + 2]));
                pc += 3;
                VM_PUSH(v);
                break;
            }
            case RXR_NODE_CLEAR:
                if (sp) --sp;
                break;
            case RXR_NODE_DUP:
                if (sp) {
                    v3 t = stack[sp - 1];
                    VM_PUSH(t);
                }
                break;
            case RXR_NODE_PACK2: {
                v3 x, y;
                VM_POP(y);
                VM_POP(x);
                VM_PUSH(mk(x.x, y.x, 0.0f));
                break;
            }
            case RXR_NODE_PACK3: {
                v3 x, y, z;
                VM_POP(z);
                VM_POP(y);
                VM_POP(x);
                VM_PUSH(mk(x.x, y.x, z.x));
                break;
            }
            // ---- control flow (flattened If / For / FunctionCall / Return)
            case VM_JMP: pc = code[pc]; break;
            case VM_JZ: {
                v3 c;
                VM_POP(c);
                pc = (c.x != 0.0f) ? pc + 1 : code[pc];
                break;
            }
            case VM_FOR_ENTER:
                if (nloops >= RXR_VM_LOOPS) VM_FAIL(VMF_LOOP_DEPTH);
                loop_base[nloops++] = sp;
                break;
            case VM_FOR_TRUNC:
                if (sp > loop_base[nloops - 1]) sp = loop_base[nloops - 1];
                break;
            case VM_FOR_COND: {
                v3 z;
                VM_POP(z);
                pc = (z.x == 0.0f) ? code[pc] : pc + 1;
                break;
            }
            case VM_FOR_EXIT: --nloops; break;
            case VM_CALL: {  // :186-223
                const uint32_t arity = code[pc], total = code[pc + 1], target = code[pc + 2];
                pc += 3;
                if (nframes >= RXR_VM_FRAMES) VM_FAIL(VMF_CALL_DEPTH);
                const uint32_t nb = lbase + llen;
                if (nb + total > RXR_VM_LOCALS) VM_FAIL(VMF_LOCALS_OVERFLOW);
                for (uint32_t i = 0; i < total; ++i) locals[nb + i] = splat(0.0f);
                for (uint32_t i = arity; i-- > 0u;) {
                    if (sp) {
                        if (i >= total) VM_FAIL(VMF_LOCAL_INDEX);
                        locals[nb + i] = stack[--sp];
                    }
                }
                fr_pc[nframes] = pc;
                fr_base[nframes] = sp;
                fr_lbase[nframes] = lbase;
                fr_llen[nframes] = llen;
                ++nframes;
                lbase = nb;
                llen = total;
                pc = target;
                break;
            }
            case VM_RETURN: {  // :224-234
                v3 v;
                if (sp) v = stack[--sp];
                else if (has_ret) v = ret;
                else v = splat(0.0f);
                ret = v;
                has_ret = true;
                pc = code[pc];  // the function's VM_ENDFN
                break;
            }
            case VM_ENDFN: {
                if (nframes == 0u) goto done;  // end of `shade`
                --nframes;
                const uint32_t base = fr_base[nframes];
                v3 r;
                if (has_ret) {
                    r = ret;
                    has_ret = false;
                } else if (sp > base) {
                    r = stack[--sp];
                } else {
                    r = splat(0.0f);
                }
                if (sp > base) sp = base;
                lbase = fr_lbase[nframes];
                llen = fr_llen[nframes];
                pc = fr_pc[nframes];
                VM_PUSH(r);
                break;
            }
            case VM_FAULT: VM_FAIL(code[pc]);
            // ---- arithmetic
            case RXR_NODE_ADD: VM_BIN(mk(a.x + b.x, a.y + b.y, a.z + b.z))
            case RXR_NODE_SUB: VM_BIN(mk(a.x - b.x, a.y - b.y, a.z - b.z))
            case RXR_NODE_MUL: VM_BIN(mk(a.x * b.x, a.y * b.y, a.z * b.z))
            case RXR_NODE_DIV: VM_BIN(mk(a.x / b.x, a.y / b.y, a.z / b.z))
            case RXR_NODE_LENGTH: VM_UN(splat(sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z)))
            case RXR_NODE_LENGTH2: VM_UN(mk(sqrtf(a.x * a.x + a.y * a.y), 0.0f, 0.0f))
            case RXR_NODE_LENGTH3: VM_UN(mk(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), 0.0f, 0.0f))
            case RXR_NODE_ABS: VM_UN(mk(fabsf(a.x), fabsf(a.y), fabsf(a.z)))
            case RXR_NODE_SIN: VM_UN(mk(sinf(a.x), sinf(a.y), sinf(a.z)))
            case RXR_NODE_SIN1: VM_UN(mk(sinf(a.x), 0.0f, 0.0f))
            case RXR_NODE_SIN2: VM_UN(mk(sinf(a.x), sinf(a.y), 0.0f))
            case RXR_NODE_COS: VM_UN(mk(cosf(a.x), cosf(a.y), cosf(a.z)))
            case RXR_NODE_COS1: VM_UN(mk(sinf(a.x), 0.0f, 0.0f))        // :337-344: the reference computes the sine
            case RXR_NODE_COS2: VM_UN(mk(sinf(a.x), sinf(a.y), 0.0f))
            case RXR_NODE_TAN: VM_UN(mk(tanf(a.x), tanf(a.y), tanf(a.z)))
            case RXR_NODE_ATAN: VM_UN(mk(atanf(a.x), atanf(a.y), atanf(a.z)))
            case RXR_NODE_ATAN2: VM_BIN(mk(atan2f(a.x, b.x), atan2f(a.y, b.y), atan2f(a.z, b.z)))
            case RXR_NODE_NORMALIZE: {  // :345-353
                v3 a;
                VM_POP(a);
                float len = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
                if (len > 0.0f) a = mk(a.x / len, a.y / len, a.z / len);
                VM_PUSH(a);
                break;
            }
            case RXR_NODE_ROTATE2D: {  // :367-374
                v3 angle, v;
                VM_POP(angle);
                VM_POP(v);
                float rad = angle.x * (3.14159265358979323846f / 180.0f);
                float s = sinf(rad), c = cosf(rad);
                VM_PUSH(mk(v.x * c - v.y * s, v.x * s + v.y * c, v.z));
                break;
            }
            case RXR_NODE_DOT: VM_BIN(splat((a.x * b.x + a.y * b.y) + a.z * b.z))
            case RXR_NODE_DOT2: VM_BIN(mk(a.x * b.x + a.y * b.y, 0.0f, 0.0f))
            case RXR_NODE_DOT3: VM_BIN(mk(a.x * b.x + a.y * b.y + a.z * b.z, 0.0f, 0.0f))
            case RXR_NODE_CROSS: VM_BIN(mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x))
            case RXR_NODE_FLOOR: VM_UN(mk(floorf(a.x), floorf(a.y), floorf(a.z)))
            case RXR_NODE_CEIL: VM_UN(mk(ceilf(a.x), ceilf(a.y), ceilf(a.z)))
            case RXR_NODE_ROUND: VM_UN(mk(roundf(a.x), roundf(a.y), roundf(a.z)))
            case RXR_NODE_FRACT: VM_UN(mk(a.x - floorf(a.x), a.y - floorf(a.y), a.z - floorf(a.z)))
            case RXR_NODE_MOD: VM_BIN(mk(a.x - b.x * floorf(a.x / b.x), a.y - b.y * floorf(a.y / b.y), a.z - b.z * floorf(a.z / b.z)))
            case RXR_NODE_RADIANS: {
                const float k = 3.14159265358979323846f / 180.0f;
                VM_UN(mk(a.x * k, a.y * k, a.z * k))
            }
            case RXR_NODE_DEGREES: {
                const float k = 57.2957795130823208767981548141051703f;
                VM_UN(mk(a.x * k, a.y * k, a.z * k))
            }
            case RXR_NODE_MIN: VM_BIN(mk(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)))
            case RXR_NODE_MAX: VM_BIN(mk(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)))
            case RXR_NODE_MIX: {  // a + (b - a) * c
                v3 a, b, c;
                VM_POP(c);
                VM_POP(b);
                VM_POP(a);
                VM_PUSH(mk(a.x + (b.x - a.x) * c.x, a.y + (b.y - a.y) * c.y, a.z + (b.z - a.z) * c.z));
                break;
            }
            case RXR_NODE_SMOOTHSTEP: {  // :456-474
                v3 a, b, c;
                VM_POP(c);
                VM_POP(b);
                VM_POP(a);
                float denom = b.x - a.x;
                float t = denom != 0.0f ? (c.x - a.x) / denom : 0.0f;
                if (t < 0.0f) t = 0.0f;
                else if (t > 1.0f) t = 1.0f;
                VM_PUSH(splat(t * t * (3.0f - 2.0f * t)));
                break;
            }
            case RXR_NODE_STEP: VM_BIN(mk(b.x >= a.x ? 1.0f : 0.0f, b.y >= a.y ? 1.0f : 0.0f, b.z >= a.z ? 1.0f : 0.0f))
            case RXR_NODE_CLAMP: {  // f32::clamp panics unless min <= max
                v3 a, b, c;
                VM_POP(c);
                VM_POP(b);
                VM_POP(a);
                if (!(b.x <= c.x) || !(b.y <= c.y) || !(b.z <= c.z)) VM_FAIL(VMF_CLAMP_BOUNDS);
                VM_PUSH(mk(rclampf(a.x, b.x, c.x), rclampf(a.y, b.y, c.y), rclampf(a.z, b.z, c.z)));
                break;
            }
            case RXR_NODE_SQRT: VM_UN(mk(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)))
            case RXR_NODE_LOG: VM_UN(mk(logf(a.x), logf(a.y), logf(a.z)))
            case RXR_NODE_POW: VM_BIN(mk(powf(a.x, b.x), powf(a.y, b.y), powf(a.z, b.z)))
            case RXR_NODE_EQ: VM_BIN(splat(a.x == b.x ? 1.0f : 0.0f))
            case RXR_NODE_NE: VM_BIN(splat(a.x != b.x ? 1.0f : 0.0f))
            case RXR_NODE_LT: VM_BIN(splat(a.x < b.x ? 1.0f : 0.0f))
            case RXR_NODE_LE: VM_BIN(splat(a.x <= b.x ? 1.0f : 0.0f))
            case RXR_NODE_GT: VM_BIN(splat(a.x > b.x ? 1.0f : 0.0f))
            case RXR_NODE_GE: VM_BIN(splat(a.x >= b.x ? 1.0f : 0.0f))
            case RXR_NODE_AND: VM_BIN(splat(((a.x != 0.0f) & (b.x != 0.0f)) ? 1.0f : 0.0f))
            case RXR_NODE_OR: VM_BIN(splat(((a.x != 0.0f) | (b.x != 0.0f)) ? 1.0f : 0.0f))
            case RXR_NODE_NOT: VM_UN(splat(a.x == 0.0f ? 1.0f : 0.0f))
            case RXR_NODE_NEG: VM_UN(mk(-a.x, -a.y, -a.z))
            case RXR_NODE_PRINT: {  // println! only
                v3 a;
                VM_POP(a);
                break;
            }
            // ---- the fragment's fields
            case RXR_NODE_UV: VM_PUSH(io.uv); break;
            case RXR_NODE_SET_UV: VM_POP(io.uv); break;
            case RXR_NODE_NORMAL: VM_PUSH(io.normal); break;
            case RXR_NODE_SET_NORMAL: {  // .normalized()
                v3 a;
                VM_POP(a);
                float len = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
                io.normal = mk(a.x / len, a.y / len, a.z / len);
                break;
            }
            case RXR_NODE_HITPOINT: VM_PUSH(io.hitpoint); break;
            case RXR_NODE_TIME: VM_PUSH(io.time); break;
            case RXR_NODE_COLOR: VM_PUSH(io.color); break;
            case RXR_NODE_SET_COLOR: VM_POP(io.color); break;
            case RXR_NODE_ROUGHNESS: VM_PUSH(io.roughness); break;
            case RXR_NODE_SET_ROUGHNESS: VM_POP(io.roughness); break;
            case RXR_NODE_METALLIC: VM_PUSH(io.metallic); break;
            case RXR_NODE_SET_METALLIC: VM_POP(io.metallic); break;
            case RXR_NODE_EMISSIVE: VM_PUSH(io.emissive); break;
            case RXR_NODE_OPACITY: VM_PUSH(io.opacity); break;
            case RXR_NODE_SET_OPACITY: VM_POP(io.opacity); break;
            case RXR_NODE_BUMP: VM_PUSH(io.bump); break;
            case RXR_NODE_SET_BUMP: VM_POP(io.bump); break;
            case RXR_NODE_SAMPLE: {  // :570-578
                v3 a, b;
                VM_POP(b);
                VM_POP(a);
                uint32_t id = as_usize_sat(b.x);
                v3 o = splat(0.0f);
                if (id < P.n_patterns) o = pattern_sample(P, P.patterns[id], a);
                VM_PUSH(o);
                break;
            }
            case RXR_NODE_SAMPLE_NORMAL: {  // :579-594
                v3 a, b;
                VM_POP(b);
                VM_POP(a);
                uint32_t id = as_usize_sat(b.x);
                v3 o = splat(0.0f);
                if (id < P.n_normal_patterns) {
                    v3 nm = pattern_sample(P, P.patterns[P.n_patterns + id], a);
                    o = mk(nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, nm.z * 2.0f - 1.0f);
                }
                VM_PUSH(o);
                break;
            }
            case RXR_NODE_PALETTE_INDEX: {  // :694-701: pushes nothing for a missing / empty slot
                v3 a;
                VM_POP(a);
                uint32_t id = as_usize_sat(a.x);
                if (id < P.n_palette && P.palette[4u * id + 3u] != 0.0f) {
                    v3 c = mk(P.palette[4u * id], P.palette[4u * id + 1u], P.palette[4u * id + 2u]);
                    VM_PUSH(c);
                }
                break;
            }
            default: VM_FAIL(VMF_BAD_OPCODE);
        }
    }
done:
    if (fault) *P.vm_fault = fault;
    return fault;
}

#undef VM_FAIL
#undef VM_POP
#undef VM_PUSH
#undef VM_UN
#undef VM_BIN

}  // namespace rxvm
