// rusteria_vm.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rusterix_oracle.hpp).
//
// CPU restatement of the Rusteria stack VM as the raster loops use it: `Execution`
// (rusteria/src/node/execution.rs:8-107), `Execution::execute` (:109-727) over the reference's NodeOp
// TREE (rusteria/src/node/nodeop.rs:12-103) -- nested blocks executed recursively exactly like the
// reference, NOT the flattened jump code the device runs --, `shade` (:741-749), `TexStorage::sample`
// (rusteria/src/textures/mod.rs:28-31, :125-141).  Where the reference panics (stack underflow, index
// out of range, clamp with min > max, a For loop past 10 000 000 iterations) this throws vm::Fault,
// which the raster loops turn into RXR_ERR_INVALID.
//
// PINNED BY THE REFERENCE'S OWN TWO TESTS, otherwise unpinned: `addition` and `fib` (rusteria/src/lib.rs:274-296) are the only
// assertions the reference holds for this VM -- tests/test_oracle_reference_tests.py runs the NodeOp lists its compiler emits for
// them and requires 4.0 and fib(27) = 196418.0 (Push, globals, locals, Le, If / else, FunctionCall, Return, Add, Sub, recursion 27
// deep).  Every other opcode is pinned only by the cited source text and the per-opcode known answers in tests/test_oracle_vm.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/rusterix_vek.hpp"
#include "../include/rxr.h"

namespace orc {
// f32::min / f32::max: a NaN operand is dropped; for operands that compare EQUAL (+0.0 and -0.0) Rust documents "either may be
// returned".  What rustc's x86-64 back end returns is fixed, though: llvm.minnum / llvm.maxnum lower to MINSS / MAXSS with the operands
// swapped plus a select on the first operand's NaN-ness ("Max = Op1 > Op0 ? Op1 : Op0", X86ISelLowering.cpp combineFMinNumFMaxNum):
// ties return `self`.  glibc's fmin / fmax return the OTHER operand there, the GPU's v_max_f32 always +0 -- the sign of a zero shows
// as soon as a program divides by it (found by the fuzz sweep, seed 52871: Max(0.0, -0.0) as a divisor).
inline float rmin(float a, float b) { return a != a ? b : (b < a ? b : a); }
inline float rmax(float a, float b) { return a != a ? b : (b > a ? b : a); }
namespace vm {

using rvek::Vec3;

struct Fault {
    const char *what;
};

// rusteria/src/node/nodeop.rs:12-103
struct NodeOp {
    int op = RXR_NODE_CLEAR;
    uint32_t a = 0, b = 0, c = 0;      // index | (arity, total_locals, function index)
    Vec3 value{0, 0, 0};               // Push
    std::vector<uint8_t> comps;        // Get/SetComponents swizzle
    bool has_else = false;             // If
    std::vector<NodeOp> blk[4];        // If: then, else; For: init, cond, incr, body
};

// rusteria/src/node/program.rs:7-29 (fields the raster path reads)
struct Program {
    size_t globals = 0;
    std::vector<std::vector<NodeOp>> user_functions;
    int shade_index = -1;
    size_t shade_locals = 0;
};

// rusteria/src/textures/mod.rs:10-15
struct TexStorage {
    size_t width = 0, height = 0;
    std::vector<Vec3> data;
    static int rem_i32(int a, int m) {  // :113-117
        int r = a % m;
        return r < 0 ? r + m : r;
    }
    static int floor_to_i32(float x) {  // `x.floor() as i32`: saturating, NaN -> 0
        float f = std::floor(x);
        if (!(f == f)) return 0;
        if (f <= -2147483648.0f) return INT32_MIN;
        if (f >= 2147483648.0f) return INT32_MAX;
        return (int)f;
    }
    Vec3 sample(Vec3 uv) const {  // :28-31 with sample_index :125-141
        float u = uv.x, v = uv.y;
        u = u - std::floor(u);
        v = v - std::floor(v);
        int x = floor_to_i32(u * (float)width);
        int y = floor_to_i32(v * (float)height);
        x = rem_i32(x, (int)width);
        y = rem_i32(y, (int)height);
        return data[(size_t)y * width + (size_t)x];
    }
};

// what the VM reads besides the program: the global pattern banks (rusteria/src/textures/patterns.rs)
// and assets.palette (ThePalette.colors: Vec<Option<TheColor>>)
struct Env {
    std::vector<TexStorage> patterns, patterns_normal;
    std::vector<Vec3> palette_rgb;
    std::vector<uint8_t> palette_present;
};

inline size_t as_usize(float x) {  // `x as usize`
    if (!(x == x) || x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return (size_t)UINT64_MAX;
    return (size_t)x;
}

// rusteria/src/node/execution.rs:8-56
struct Execution {
    std::vector<Vec3> globals, locals, stack;
    std::vector<std::vector<Vec3>> locals_stack;
    bool has_return = false;
    Vec3 return_value{0, 0, 0};
    Vec3 uv{0, 0, 0}, color{0, 0, 0}, roughness{0.5f, 0.5f, 0.5f}, metallic{0, 0, 0}, emissive{0, 0, 0}, opacity{0, 0, 0},
        bump{0, 0, 0}, normal{0, 0, 0}, hitpoint{0, 0, 0}, time{0, 0, 0};

    explicit Execution(size_t var_size = 0) : globals(var_size, Vec3{0, 0, 0}) {}  // :59-79

    void reset(size_t var_size) {  // :102-107
        if (var_size != globals.size()) globals.resize(var_size, Vec3{0, 0, 0});
    }

    Vec3 pop() {  // self.stack.pop().unwrap()
        if (stack.empty()) throw Fault{"stack underflow"};
        Vec3 v = stack.back();
        stack.pop_back();
        return v;
    }
    void push(Vec3 v) { stack.push_back(v); }
    static Vec3 splat(float x) { return Vec3{x, x, x}; }
    template <class F>
    static Vec3 map(Vec3 a, F f) {
        return Vec3{f(a.x), f(a.y), f(a.z)};
    }

    // :109-727
    void execute(const std::vector<NodeOp> &code, const Program &program, const Env &env) {
        for (const NodeOp &op : code) {
            if (has_return) break;  // :112-114
            switch (op.op) {
                case RXR_NODE_LOAD_GLOBAL:
                    if (op.a >= globals.size()) throw Fault{"global index"};
                    push(globals[op.a]);
                    break;
                case RXR_NODE_STORE_GLOBAL: {
                    if (op.a >= globals.size()) throw Fault{"global index"};
                    globals[op.a] = pop();
                    break;
                }
                case RXR_NODE_LOAD_LOCAL:
                    if (op.a >= locals.size()) throw Fault{"local index"};
                    push(locals[op.a]);
                    break;
                case RXR_NODE_STORE_LOCAL: {
                    if (op.a >= locals.size()) throw Fault{"local index"};
                    locals[op.a] = pop();
                    break;
                }
                case RXR_NODE_SWAP: {
                    Vec3 b = pop(), a = pop();
                    push(b);
                    push(a);
                    break;
                }
                case RXR_NODE_GET_COMPONENTS: {  // :134-157
                    Vec3 v = pop();
                    float result[8];
                    size_t n = 0;
                    bool too_many = false;
                    for (uint8_t index : op.comps) {
                        float f;
                        if (index == 0) f = v.x;
                        else if (index == 1) f = v.y;
                        else if (index == 2) f = v.z;
                        else continue;
                        if (n < 8) result[n] = f;
                        else too_many = true;
                        ++n;
                    }
                    (void)too_many;
                    if (n == 1) push(splat(result[0]));
                    else if (n == 2) push(Vec3{result[0], result[1], 0.0f});
                    else if (n == 3) push(Vec3{result[0], result[1], result[2]});
                    else push(splat(0.0f));
                    break;
                }
                case RXR_NODE_SET_COMPONENTS: {  // :158-183
                    Vec3 value = pop();
                    Vec3 target = pop();
                    float components[3] = {value.x, value.y, value.z};
                    size_t nc = op.comps.size() >= 1 && op.comps.size() <= 3 ? op.comps.size() : 0;
                    for (size_t i = 0; i < op.comps.size(); ++i) {
                        if (i >= nc) break;
                        if (op.comps[i] == 0) target.x = components[i];
                        else if (op.comps[i] == 1) target.y = components[i];
                        else if (op.comps[i] == 2) target.z = components[i];
                    }
                    push(target);
                    break;
                }
                case RXR_NODE_PUSH: push(op.value); break;
                case RXR_NODE_CLEAR:
                    if (!stack.empty()) stack.pop_back();
                    break;
                case RXR_NODE_FUNCTION_CALL: {  // :186-223
                    locals_stack.push_back(locals);
                    locals.assign(op.b, Vec3{0, 0, 0});
                    for (size_t index = op.a; index-- > 0;) {
                        if (!stack.empty()) {
                            Vec3 arg = stack.back();
                            stack.pop_back();
                            if (index >= locals.size()) throw Fault{"argument index"};
                            locals[index] = arg;
                        }
                    }
                    size_t stack_base = stack.size();
                    if (op.c >= program.user_functions.size()) throw Fault{"function index"};
                    execute(program.user_functions[op.c], program, env);
                    Vec3 ret;
                    if (has_return) {
                        ret = return_value;
                        has_return = false;
                    } else if (stack.size() > stack_base) {
                        ret = pop();
                    } else {
                        ret = Vec3{0, 0, 0};
                    }
                    if (stack.size() > stack_base) stack.resize(stack_base);
                    if (!locals_stack.empty()) {
                        locals = locals_stack.back();
                        locals_stack.pop_back();
                    }
                    push(ret);
                    break;
                }
                case RXR_NODE_RETURN: {  // :224-234
                    Vec3 v;
                    if (!stack.empty()) {
                        v = stack.back();
                        stack.pop_back();
                    } else if (has_return) {
                        v = return_value;
                    } else {
                        v = Vec3{0, 0, 0};
                    }
                    return_value = v;
                    has_return = true;
                    return;  // break out of the op loop
                }
                case RXR_NODE_PACK2: {
                    Vec3 y = pop(), x = pop();
                    push(Vec3{x.x, y.x, 0.0f});
                    break;
                }
                case RXR_NODE_PACK3: {
                    Vec3 z = pop(), y = pop(), x = pop();
                    push(Vec3{x.x, y.x, z.x});
                    break;
                }
                case RXR_NODE_DUP:
                    if (!stack.empty()) push(stack.back());
                    break;
                case RXR_NODE_FOR: {  // :251-278
                    size_t base = stack.size();
                    size_t iter = 0;
                    execute(op.blk[0], program, env);
                    if (stack.size() > base) stack.resize(base);
                    for (;;) {
                        execute(op.blk[1], program, env);
                        Vec3 z = pop();
                        if (z.x == 0.0f) break;
                        if (stack.size() > base) stack.resize(base);
                        execute(op.blk[3], program, env);
                        if (stack.size() > base) stack.resize(base);
                        execute(op.blk[2], program, env);
                        if (stack.size() > base) stack.resize(base);
                        iter += 1;
                        if (iter > 10000000) throw Fault{"Inifinite for loop detected"};
                    }
                    break;
                }
                case RXR_NODE_IF: {  // :279-286
                    bool value = pop().x != 0.0f;
                    if (value) execute(op.blk[0], program, env);
                    else if (op.has_else) execute(op.blk[1], program, env);
                    break;
                }
                case RXR_NODE_ADD: { Vec3 b = pop(), a = pop(); push(a + b); break; }
                case RXR_NODE_SUB: { Vec3 b = pop(), a = pop(); push(a - b); break; }
                case RXR_NODE_MUL: { Vec3 b = pop(), a = pop(); push(Vec3{a.x * b.x, a.y * b.y, a.z * b.z}); break; }
                case RXR_NODE_DIV: { Vec3 b = pop(), a = pop(); push(Vec3{a.x / b.x, a.y / b.y, a.z / b.z}); break; }
                case RXR_NODE_LENGTH: { Vec3 a = pop(); push(splat(rvek::magnitude(a))); break; }
                case RXR_NODE_LENGTH2: { Vec3 a = pop(); push(Vec3{std::sqrt(a.x * a.x + a.y * a.y), 0.0f, 0.0f}); break; }
                case RXR_NODE_LENGTH3: { Vec3 a = pop(); push(Vec3{std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z), 0.0f, 0.0f}); break; }
                case RXR_NODE_ABS: { Vec3 a = pop(); push(map(a, [](float x) { return std::fabs(x); })); break; }
                case RXR_NODE_SIN: { Vec3 a = pop(); push(map(a, [](float x) { return std::sin(x); })); break; }
                case RXR_NODE_SIN1: { Vec3 a = pop(); push(Vec3{std::sin(a.x), 0.0f, 0.0f}); break; }
                case RXR_NODE_SIN2: { Vec3 a = pop(); push(Vec3{std::sin(a.x), std::sin(a.y), 0.0f}); break; }
                case RXR_NODE_COS: { Vec3 a = pop(); push(map(a, [](float x) { return std::cos(x); })); break; }
                // :337-344: Cos1 / Cos2 compute the SINE in the reference
                case RXR_NODE_COS1: { Vec3 a = pop(); push(Vec3{std::sin(a.x), 0.0f, 0.0f}); break; }
                case RXR_NODE_COS2: { Vec3 a = pop(); push(Vec3{std::sin(a.x), std::sin(a.y), 0.0f}); break; }
                case RXR_NODE_NORMALIZE: {  // :345-353
                    Vec3 a = pop();
                    float len = rvek::magnitude(a);
                    push(len > 0.0f ? Vec3{a.x / len, a.y / len, a.z / len} : a);
                    break;
                }
                case RXR_NODE_TAN: { Vec3 a = pop(); push(map(a, [](float x) { return std::tan(x); })); break; }
                case RXR_NODE_ATAN: { Vec3 a = pop(); push(map(a, [](float x) { return std::atan(x); })); break; }
                case RXR_NODE_ATAN2: {
                    Vec3 b = pop(), a = pop();
                    push(Vec3{std::atan2(a.x, b.x), std::atan2(a.y, b.y), std::atan2(a.z, b.z)});
                    break;
                }
                case RXR_NODE_ROTATE2D: {  // :367-374
                    Vec3 angle = pop(), v = pop();
                    float rad = angle.x * (3.14159265358979323846f / 180.0f);  // f32::to_radians
                    float s = std::sin(rad), c = std::cos(rad);
                    push(Vec3{v.x * c - v.y * s, v.x * s + v.y * c, v.z});
                    break;
                }
                case RXR_NODE_DOT: { Vec3 b = pop(), a = pop(); push(splat(rvek::dot(a, b))); break; }
                case RXR_NODE_DOT2: { Vec3 b = pop(), a = pop(); push(Vec3{a.x * b.x + a.y * b.y, 0.0f, 0.0f}); break; }
                case RXR_NODE_DOT3: { Vec3 b = pop(), a = pop(); push(Vec3{a.x * b.x + a.y * b.y + a.z * b.z, 0.0f, 0.0f}); break; }
                case RXR_NODE_CROSS: {  // vek Vec3::cross
                    Vec3 b = pop(), a = pop();
                    push(Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x});
                    break;
                }
                case RXR_NODE_FLOOR: { Vec3 a = pop(); push(map(a, [](float x) { return std::floor(x); })); break; }
                case RXR_NODE_CEIL: { Vec3 a = pop(); push(map(a, [](float x) { return std::ceil(x); })); break; }
                case RXR_NODE_ROUND: { Vec3 a = pop(); push(map(a, [](float x) { return std::round(x); })); break; }
                case RXR_NODE_FRACT: { Vec3 a = pop(); push(map(a, [](float x) { return x - std::floor(x); })); break; }
                case RXR_NODE_MOD: {  // :421-428
                    Vec3 b = pop(), a = pop();
                    push(Vec3{a.x - b.x * std::floor(a.x / b.x), a.y - b.y * std::floor(a.y / b.y), a.z - b.z * std::floor(a.z / b.z)});
                    break;
                }
                case RXR_NODE_RADIANS: { Vec3 a = pop(); push(map(a, [](float x) { return x * (3.14159265358979323846f / 180.0f); })); break; }
                case RXR_NODE_DEGREES: { Vec3 a = pop(); push(map(a, [](float x) { return x * 57.2957795130823208767981548141051703f; })); break; }
                case RXR_NODE_MIN: { Vec3 b = pop(), a = pop(); push(Vec3{rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z)}); break; }   // (execution.rs:439-444: a.x.min(b.x); ties return a, rusterix_oracle.hpp)
                case RXR_NODE_MAX: { Vec3 b = pop(), a = pop(); push(Vec3{rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z)}); break; }
                case RXR_NODE_MIX: {  // :449-455: a + (b - a) * c
                    Vec3 c = pop(), b = pop(), a = pop();
                    push(Vec3{a.x + (b.x - a.x) * c.x, a.y + (b.y - a.y) * c.y, a.z + (b.z - a.z) * c.z});
                    break;
                }
                case RXR_NODE_SMOOTHSTEP: {  // :456-474
                    Vec3 c = pop(), b = pop(), a = pop();
                    float denom = b.x - a.x;
                    float t = denom != 0.0f ? (c.x - a.x) / denom : 0.0f;
                    if (t < 0.0f) t = 0.0f;
                    else if (t > 1.0f) t = 1.0f;
                    push(splat(t * t * (3.0f - 2.0f * t)));
                    break;
                }
                case RXR_NODE_STEP: {
                    Vec3 b = pop(), a = pop();
                    push(Vec3{b.x >= a.x ? 1.0f : 0.0f, b.y >= a.y ? 1.0f : 0.0f, b.z >= a.z ? 1.0f : 0.0f});
                    break;
                }
                case RXR_NODE_CLAMP: {  // f32::clamp asserts min <= max
                    Vec3 c = pop(), b = pop(), a = pop();
                    if (!(b.x <= c.x) || !(b.y <= c.y) || !(b.z <= c.z)) throw Fault{"clamp: min > max"};
                    push(Vec3{rvek::rclamp(a.x, b.x, c.x), rvek::rclamp(a.y, b.y, c.y), rvek::rclamp(a.z, b.z, c.z)});
                    break;
                }
                case RXR_NODE_SQRT: { Vec3 a = pop(); push(map(a, [](float x) { return std::sqrt(x); })); break; }
                case RXR_NODE_LOG: { Vec3 a = pop(); push(map(a, [](float x) { return std::log(x); })); break; }
                case RXR_NODE_POW: {
                    Vec3 b = pop(), a = pop();
                    push(Vec3{std::pow(a.x, b.x), std::pow(a.y, b.y), std::pow(a.z, b.z)});
                    break;
                }
                case RXR_NODE_EQ: { Vec3 b = pop(), a = pop(); push(splat(a.x == b.x ? 1.0f : 0.0f)); break; }
                case RXR_NODE_NE: { Vec3 b = pop(), a = pop(); push(splat(a.x != b.x ? 1.0f : 0.0f)); break; }
                case RXR_NODE_LT: { Vec3 b = pop(), a = pop(); push(splat(a.x < b.x ? 1.0f : 0.0f)); break; }
                case RXR_NODE_LE: { Vec3 b = pop(), a = pop(); push(splat(a.x <= b.x ? 1.0f : 0.0f)); break; }
                case RXR_NODE_GT: { Vec3 b = pop(), a = pop(); push(splat(a.x > b.x ? 1.0f : 0.0f)); break; }
                case RXR_NODE_GE: { Vec3 b = pop(), a = pop(); push(splat(a.x >= b.x ? 1.0f : 0.0f)); break; }
                case RXR_NODE_AND: { Vec3 b = pop(), a = pop(); push(splat(((a.x != 0.0f) & (b.x != 0.0f)) ? 1.0f : 0.0f)); break; }
                case RXR_NODE_OR: { Vec3 b = pop(), a = pop(); push(splat(((a.x != 0.0f) | (b.x != 0.0f)) ? 1.0f : 0.0f)); break; }
                case RXR_NODE_NOT: { Vec3 a = pop(); push(splat(a.x == 0.0f ? 1.0f : 0.0f)); break; }
                case RXR_NODE_NEG: { Vec3 a = pop(); push(-a); break; }
                case RXR_NODE_PRINT: (void)pop(); break;  // println! only
                case RXR_NODE_UV: push(uv); break;
                case RXR_NODE_SET_UV: uv = pop(); break;
                case RXR_NODE_NORMAL: push(normal); break;
                case RXR_NODE_SET_NORMAL: normal = rvek::normalized(pop()); break;
                case RXR_NODE_HITPOINT: push(hitpoint); break;
                case RXR_NODE_TIME: push(time); break;
                case RXR_NODE_COLOR: push(color); break;
                case RXR_NODE_SET_COLOR: color = pop(); break;
                case RXR_NODE_ROUGHNESS: push(roughness); break;
                case RXR_NODE_SET_ROUGHNESS: roughness = pop(); break;
                case RXR_NODE_METALLIC: push(metallic); break;
                case RXR_NODE_SET_METALLIC: metallic = pop(); break;
                case RXR_NODE_EMISSIVE: push(emissive); break;
                case RXR_NODE_SET_EMISSIVE: emissive = pop(); break;
                case RXR_NODE_OPACITY: push(opacity); break;
                case RXR_NODE_SET_OPACITY: opacity = pop(); break;
                case RXR_NODE_BUMP: push(bump); break;
                case RXR_NODE_SET_BUMP: bump = pop(); break;
                case RXR_NODE_SAMPLE: {  // :570-578
                    Vec3 b = pop(), a = pop();
                    size_t id = as_usize(b.x);
                    if (id < env.patterns.size()) push(env.patterns[id].sample(a));
                    else push(Vec3{0, 0, 0});
                    break;
                }
                case RXR_NODE_SAMPLE_NORMAL: {  // :579-594
                    Vec3 b = pop(), a = pop();
                    size_t id = as_usize(b.x);
                    if (id < env.patterns_normal.size()) {
                        Vec3 nm = env.patterns_normal[id].sample(a);
                        push(Vec3{nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, nm.z * 2.0f - 1.0f});
                    } else {
                        push(Vec3{0, 0, 0});
                    }
                    break;
                }
                case RXR_NODE_PALETTE_INDEX: {  // :694-701: pushes NOTHING when the slot is missing or None
                    Vec3 a = pop();
                    size_t id = as_usize(a.x);
                    if (id < env.palette_rgb.size() && env.palette_present[id]) push(env.palette_rgb[id]);
                    break;
                }
                case RXR_NODE_ALLOC:
                case RXR_NODE_ITERATE:
                case RXR_NODE_SAVE:
                    // texture baking (Rusteria::shade) -- not part of per-fragment shading
                    throw Fault{"Alloc / Iterate / Save are outside the raster path"};
                default: throw Fault{"unknown opcode"};
            }
        }
    }

    // :741-749
    void shade(size_t index, const Program &program, const Env &env) {
        stack.clear();
        has_return = false;
        locals.resize(program.shade_locals, Vec3{0, 0, 0});
        if (index >= program.user_functions.size()) throw Fault{"shade index"};
        execute(program.user_functions[index], program, env);
    }
};

// the NodeOp tree from its word serialisation (include/rxr.h, "Rusteria shader programs"); false if malformed
inline bool parse_block(const uint32_t *w, size_t n, std::vector<NodeOp> &out, int depth = 0) {
    if (depth > 64) return false;
    size_t i = 0;
    auto need = [&](size_t k) { return i + k <= n; };
    while (i < n) {
        NodeOp op;
        op.op = (int)w[i++];
        if (op.op < 0 || op.op >= RXR_NODE_COUNT) return false;
        switch (op.op) {
            case RXR_NODE_LOAD_GLOBAL:
            case RXR_NODE_STORE_GLOBAL:
            case RXR_NODE_LOAD_LOCAL:
            case RXR_NODE_STORE_LOCAL:
                if (!need(1)) return false;
                op.a = w[i++];
                break;
            case RXR_NODE_GET_COMPONENTS:
            case RXR_NODE_SET_COMPONENTS: {
                if (!need(1)) return false;
                uint32_t k = w[i++];
                if (k > 64 || !need(k)) return false;
                for (uint32_t j = 0; j < k; ++j) {
                    uint32_t c = w[i++];
                    op.comps.push_back((uint8_t)(c > 255u ? 255u : c));
                }
                break;
            }
            case RXR_NODE_IF: {
                if (!need(3)) return false;
                uint32_t tl = w[i], he = w[i + 1], el = w[i + 2];
                i += 3;
                if (!need((size_t)tl + el)) return false;
                op.has_else = he != 0;
                if (!parse_block(w + i, tl, op.blk[0], depth + 1)) return false;
                i += tl;
                if (!parse_block(w + i, el, op.blk[1], depth + 1)) return false;
                i += el;
                break;
            }
            case RXR_NODE_FOR: {
                if (!need(4)) return false;
                uint32_t l[4] = {w[i], w[i + 1], w[i + 2], w[i + 3]};
                i += 4;
                if (!need((size_t)l[0] + l[1] + l[2] + l[3])) return false;
                for (int k = 0; k < 4; ++k) {
                    if (!parse_block(w + i, l[k], op.blk[k], depth + 1)) return false;
                    i += l[k];
                }
                break;
            }
            case RXR_NODE_PUSH: {
                if (!need(3)) return false;
                float f[3];
                std::memcpy(f, w + i, 12);
                i += 3;
                op.value = Vec3{f[0], f[1], f[2]};
                break;
            }
            case RXR_NODE_FUNCTION_CALL:
                if (!need(3)) return false;
                op.a = w[i];
                op.b = w[i + 1];
                op.c = w[i + 2];
                i += 3;
                break;
            default: break;
        }
        out.push_back(std::move(op));
    }
    return true;
}

}  // namespace vm
}  // namespace orc
