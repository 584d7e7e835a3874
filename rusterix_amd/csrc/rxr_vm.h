// rxr_vm.h -- the Rusteria stack VM on the device: Execution::shade
// (rusteria/src/node/execution.rs:741-749) and Execution::execute (:109-727) over the jump code that
// rxr_set_shaders flattens the reference's NodeOp trees into (rxr_device.h, VM_*).
//
// One invocation per fragment, state private to the lane; see "the interpreter" below for how a wave runs them.  The reference keeps ONE Execution per tile and lets state leak from one
// fragment into the next; rxr_set_shaders only accepts programs for which that cannot matter (see
// include/rxr.h), and every invocation here starts from Execution::new's values plus the fields the raster
// loops assign before each call.
//
// Where the reference panics the lane stops and reports a VMF_* code through RasterParams.vm_fault;
// rxr_synchronize turns it into RXR_ERR_INVALID.  An invocation is bounded by RXR_VM_MAX_STEPS backward jumps and
// calls (the only ways to execute an instruction twice) so that every wave reaches the end of the kernel.
//
// sin / cos / tan / atan / atan2 / pow / ln come from the device math library and differ from the host's
// libm by a few ulp: programs that use them are compared at +-1 per 8-bit channel.
#pragma once
#ifndef RXR_JIT
#include <hip/hip_runtime.h>
#endif

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
// f32::min / f32::max as rustc's x86-64 back end compiles them (oracle/rusterix_oracle.hpp rmin / rmax): a NaN operand is dropped and
// operands that compare equal -- +0.0 and -0.0 -- return the FIRST one.  v_min_f32 / v_max_f32 order the zeros instead; a program that
// divides by the result sees the difference.
#ifdef RXR_TEST_PLAIN_MINMAX   // (a build that undoes the rule: tests/test_gpu_shader_edge_values.py must notice)
__device__ __forceinline__ float rust_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float rust_max(float a, float b) { return fmaxf(a, b); }
#else
__device__ __forceinline__ float rust_min(float a, float b) { return a != a ? b : (b < a ? b : a); }
__device__ __forceinline__ float rust_max(float a, float b) { return a != a ? b : (b > a ? b : a); }
#endif

#ifndef RXR_JIT
// ---- the interpreter ------------------------------------------------------------------------------
// All lanes of a wave that need a program run call shade() together.  Instruction fetch and decode are
// wave-uniform: every step executes the instruction at the SMALLEST program counter among the lanes still
// running ("min-pc scheduling") on exactly the lanes that are at that address, so lanes that diverged at an
// If / For / different programs re-join as soon as their paths meet, and the opcode switch is a scalar branch.
// The value stack lives in LDS (RXR_VM_LDS_STACK entries per lane, [slot][component][lane]: conflict-free for any
// per-lane depth) with its top entry cached in registers, so an operation touches LDS at most once; deeper
// stacks, locals, globals and call frames are in scratch memory.
#ifndef RXR_VM_LDS_STACK
#define RXR_VM_LDS_STACK 2  // sized for 6 workgroups per CU: see RXR_VM_WAVES_PER_SIMD in rxr_kernels.hip
#endif

// (the LDS block is passed to every access instead of being stored here: a pointer kept in a struct that itself lives in
// scratch would become a generic pointer)
// the LDS part is addressed through an explicit LDS pointer type: an access to it can then never be merged with an access to
// the scratch part into one access through a generic pointer (which this compiler also fails to select: "Illegal
// instruction detected ... $src_private_base")
typedef float __attribute__((address_space(3))) lds_float;
struct Stack {
    // The slots beyond the LDS part live in a SEPARATE array in scratch memory, handed to every access (`deep`).  As a member
    // of this struct its run-time index kept the whole struct -- the top of the stack included -- in scratch: every VM
    // instruction started with a scratch load of `tos` and ended with a scratch store (seen in the ISA).
    v3 tos;                                  // entry sp - 1
    uint32_t sp;
    __device__ __forceinline__ v3 load(const float *lds, const v3 *deep, uint32_t slot) const {
        if (slot < RXR_VM_LDS_STACK) {
            const lds_float *p = (const lds_float *)lds + (slot * 3u) * RXR_TILE_THREADS + threadIdx.x;
            return mk(p[0], p[RXR_TILE_THREADS], p[2 * RXR_TILE_THREADS]);
        }
        return deep[slot - RXR_VM_LDS_STACK];
    }
    __device__ __forceinline__ void store(float *lds, v3 *deep, uint32_t slot, v3 v) {
        if (slot < RXR_VM_LDS_STACK) {
            lds_float *p = (lds_float *)lds + (slot * 3u) * RXR_TILE_THREADS + threadIdx.x;
            p[0] = v.x;
            p[RXR_TILE_THREADS] = v.y;
            p[2 * RXR_TILE_THREADS] = v.z;
        } else {
            deep[slot - RXR_VM_LDS_STACK] = v;
        }
    }
    // callers check the depth first
    __device__ __forceinline__ void push(float *lds, v3 *deep, v3 v) {
        if (sp) store(lds, deep, sp - 1u, tos);
        tos = v;
        ++sp;
    }
    __device__ __forceinline__ v3 pop(const float *lds, const v3 *deep) {
        v3 r = tos;
        --sp;
        if (sp) tos = load(lds, deep, sp - 1u);
        return r;
    }
    __device__ __forceinline__ void truncate(const float *lds, const v3 *deep, uint32_t n) {  // Vec::truncate: only ever shrinks
        if (sp > n) {
            sp = n;
            if (sp) tos = load(lds, deep, sp - 1u);
        }
    }
};

// A lane that has finished (or faulted) parks its program counter here: "still running", "at the current instruction" and
// "behind the candidate" are then single compares of the pc, and the interpreter has one state variable less to carry.
#define VM_PC_STOPPED 0xFFFFFFFFu

// smallest pc among the running lanes of the calling wave.  Ballots and shuffles FROM active lanes only: the lanes that
// did not enter shade() (or left it) are masked off and must not be read.  One round when the wave has not diverged.
__device__ __forceinline__ uint32_t wave_min_pc(uint32_t pc, unsigned long long running_mask) {
    uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)pc, __ffsll((long long)running_mask) - 1);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(pc < cand) == 0ull, 1)) return cand;   // nobody behind the first running lane: the usual case, kept out of the loop's bookkeeping
    for (;;) {
        const unsigned long long less = __builtin_amdgcn_ballot_w64(pc < cand);   // (a stopped lane sits at VM_PC_STOPPED)
        if (!less) return cand;
        cand = (uint32_t)__builtin_amdgcn_readlane((int)pc, __ffsll((long long)less) - 1);
    }
}

#endif  // !RXR_JIT
// the libm-backed opcodes, kept out of line: they are rare and their inlined bodies would set the register
// budget of the whole interpreter
__device__ __noinline__ v3 slow_unary(uint32_t op, v3 a) {
    switch (op) {
        case RXR_NODE_SIN: return mk(sinf(a.x), sinf(a.y), sinf(a.z));
        case RXR_NODE_SIN1: return mk(sinf(a.x), 0.0f, 0.0f);
        case RXR_NODE_SIN2: return mk(sinf(a.x), sinf(a.y), 0.0f);
        case RXR_NODE_COS: return mk(cosf(a.x), cosf(a.y), cosf(a.z));
        case RXR_NODE_COS1: return mk(sinf(a.x), 0.0f, 0.0f);  // execution.rs:337-344: the reference computes the sine
        case RXR_NODE_COS2: return mk(sinf(a.x), sinf(a.y), 0.0f);
        case RXR_NODE_TAN: return mk(tanf(a.x), tanf(a.y), tanf(a.z));
        case RXR_NODE_ATAN: return mk(atanf(a.x), atanf(a.y), atanf(a.z));
        default: return mk(logf(a.x), logf(a.y), logf(a.z));  // RXR_NODE_LOG
    }
}
__device__ __noinline__ v3 slow_binary(uint32_t op, v3 a, v3 b) {
    switch (op) {
        case RXR_NODE_ATAN2: return mk(atan2f(a.x, b.x), atan2f(a.y, b.y), atan2f(a.z, b.z));
        case RXR_NODE_POW: return mk(powf(a.x, b.x), powf(a.y, b.y), powf(a.z, b.z));
        default: {  // RXR_NODE_ROTATE2D, :367-374: a = v, b = angle
            float rad = b.x * (3.14159265358979323846f / 180.0f);
            float s = sinf(rad), c = cosf(rad);
            return mk(a.x * c - a.y * s, a.x * s + a.y * c, a.z);
        }
    }
}

#ifdef RXR_JIT
// ---- a run-time compiled program set (rxr_jit.hip): no interpreter -- every program of the set is a straight-line device
// function generated from its jump code (rxr_jit_programs.h, handed to hiprtc as an in-memory header), built from the operation
// templates of rxr_jit_ops.h, and the three entry points of the interpreter forward to it
}  // namespace rxvm
#include "rxr_jit_ops.h"
#include "rxr_jit_programs.h"
namespace rxvm {
template <bool SSP>
__device__ __forceinline__ uint32_t shade_inline(const RasterParams &P, uint32_t pi, IO &io, float *) { return rxr_jit_shade(P, pi, io); }
__device__ __forceinline__ float *stack_block() { return nullptr; }
template <bool SSP>
__device__ __forceinline__ uint32_t shade_call(const RasterParams &P, uint32_t pi, IO &io) { return rxr_jit_shade(P, pi, io); }
#else
// calls f(integral_constant<K>) for K == op, LO <= op < HI, through a balanced tree of comparisons (see shade_inline)
template <uint32_t K>
struct OpConst { static constexpr uint32_t value = K; };
// How often an opcode is expected at run time (a guess from what shader programs are made of: arithmetic, pushes, field
// access and locals are hot, libm calls and stack shuffles are not).  A VM instruction costs about as many SCALAR issue slots
// as it has scalar instructions (measured: ~33 of them, 4 cycles each, against ~20 VALU), and every level of the tree is a
// compare and a branch -- so the tree splits each range where the WEIGHT halves, not the count: hot opcodes sit 2-3 levels up.
constexpr uint32_t op_weight(uint32_t op) {
    switch (op) {
        case RXR_NODE_PUSH: case RXR_NODE_ADD: case RXR_NODE_SUB: case RXR_NODE_MUL: case RXR_NODE_DIV: case RXR_NODE_LOAD_LOCAL:
        case RXR_NODE_STORE_LOCAL: case RXR_NODE_UV: case RXR_NODE_COLOR: case RXR_NODE_SET_COLOR: case RXR_NODE_HITPOINT:
        case VM_BINC: case VM_GETC: case VM_JZ: case VM_JMP: case VM_ENDFN:
            return 32u;
        case RXR_NODE_DOT: case RXR_NODE_LENGTH: case RXR_NODE_FRACT: case RXR_NODE_FLOOR: case RXR_NODE_MIX: case RXR_NODE_SMOOTHSTEP:
        case RXR_NODE_CLAMP: case RXR_NODE_MIN: case RXR_NODE_MAX: case RXR_NODE_ABS: case RXR_NODE_NORMAL: case RXR_NODE_SAMPLE:
        case RXR_NODE_PACK2: case RXR_NODE_PACK3: case RXR_NODE_STEP: case RXR_NODE_MOD: case RXR_NODE_NORMALIZE: case RXR_NODE_LT:
        case RXR_NODE_GT: case RXR_NODE_SET_ROUGHNESS: case RXR_NODE_SET_METALLIC: case RXR_NODE_NEG: case RXR_NODE_SQRT:
        case VM_SETC: case VM_CALL: case VM_RETURN: case VM_FOR_COND: case VM_FOR_ENTER: case VM_FOR_TRUNC: case VM_FOR_EXIT:
            return 8u;
        default: return 1u;
    }
}
constexpr uint32_t op_weight_sum(uint32_t lo, uint32_t hi) {
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += op_weight(i);
    return s;
}
// the split point of [lo, hi): the first m with weight(lo..m) >= half, kept strictly inside the range
constexpr uint32_t op_split(uint32_t lo, uint32_t hi) {
    const uint32_t total = op_weight_sum(lo, hi);
    uint32_t acc = 0, best = lo + 1u, best_d = 0xFFFFFFFFu;
    for (uint32_t m = lo + 1u; m < hi; ++m) {
        acc += op_weight(m - 1u);
        const uint32_t d = 2u * acc > total ? 2u * acc - total : total - 2u * acc;
        if (d < best_d) {
            best_d = d;
            best = m;
        }
    }
    return best;
}
// (the same with halves of equal size)
template <uint32_t LO, uint32_t HI, class F>
__device__ __forceinline__ void vm_dispatch_even(uint32_t op, F &f) {
    if constexpr (HI - LO == 1u) {
        f(OpConst<LO>{});
    } else {
        constexpr uint32_t MID = (LO + HI) / 2u;
        if (op < MID) vm_dispatch_even<LO, MID>(op, f);
        else vm_dispatch_even<MID, HI>(op, f);
    }
}
template <uint32_t LO, uint32_t HI, class F>
__device__ __forceinline__ void vm_dispatch(uint32_t op, F &f) {
    if constexpr (HI - LO == 1u) {
        f(OpConst<LO>{});
    } else {
        constexpr uint32_t MID = op_split(LO, HI);
        if (op < MID) vm_dispatch<LO, MID>(op, f);
        else vm_dispatch<MID, HI>(op, f);
    }
}

#define VM_FAIL(code)       \
    {                       \
        fault = (code);     \
        pc = VM_PC_STOPPED; \
        break;              \
    }
// (with static depths the host has proved operands and room for every instruction -- tag_static_depths in rxr_api.hip
// leaves a program dynamic otherwise -- so the checks are compiled out and most handlers leave `fault` alone)
#define VM_NEED(n)        \
    if constexpr (!SSP) { \
        if (st.sp < (n)) VM_FAIL(VMF_STACK_UNDERFLOW) \
    }
#define VM_ROOM           \
    if constexpr (!SSP) { \
        if (st.sp >= RXR_VM_STACK) VM_FAIL(VMF_STACK_OVERFLOW) \
    }
// one value in, one out: the top of the stack is rewritten in registers
#define VM_UN(expr)        \
    if (on) {              \
        VM_NEED(1u)        \
        const v3 a = st.tos; \
        st.tos = (expr);   \
        pc = upc + 1u;     \
    }                      \
    break;
// two in, one out: one LDS read
#define VM_BIN(expr)                      \
    if (on) {                             \
        VM_NEED(2u)                       \
        const v3 b = st.tos;              \
        const v3 a = st.load(vm_lds, deep, st.sp - 2u); \
        st.tos = (expr);                  \
        --st.sp;                          \
        pc = upc + 1u;                    \
    }                                     \
    break;
#define VM_TER(expr)                      \
    if (on) {                             \
        VM_NEED(3u)                       \
        const v3 c = st.tos;              \
        const v3 b = st.load(vm_lds, deep, st.sp - 2u); \
        const v3 a = st.load(vm_lds, deep, st.sp - 3u); \
        st.tos = (expr);                  \
        st.sp -= 2u;                      \
        pc = upc + 1u;                    \
    }                                     \
    break;
#define VM_GET(field)      \
    if (on) {              \
        VM_ROOM            \
        st.push(vm_lds, deep, field); \
        pc = upc + 1u;     \
    }                      \
    break;
#define VM_SET(field)      \
    if (on) {              \
        VM_NEED(1u)        \
        field = st.pop(vm_lds, deep); \
        pc = upc + 1u;     \
    }                      \
    break;

// Execution::shade on program `pi` for every calling lane.  Returns 0 or this lane's VMF_* code.
// Inlined form: used where most fragments pass (the opaque pass).  A real call would make the interpreter save and
// restore ~100 callee-saved VGPRs per invocation -- measured: 26 000 cycles per wave for an EMPTY program, the spill
// traffic of all resident waves going through HBM.
// SSP ("static stack pointer"): every program of the set carries the stack depth before each instruction in bits 16..23 of
// its opcode word (rxr_api.hip, tag_static_depths).  The word comes through the scalar cache, so `st.sp` is then a
// wave-uniform value: slot addresses, the LDS / scratch split of the stack and the operand checks become scalar code and the
// per-lane depth register disappears (a lane that is not at the current pc does not use its depth until it is, and then
// it is re-derived).  The handlers below are unchanged: their updates of st.sp are simply dead in this instantiation.
template <bool SSP>
__device__ __forceinline__ uint32_t shade_inline(const RasterParams &P, uint32_t pi, IO &io, float *vm_lds) {
    const DevProgram prog = P.programs[pi];
    if (prog.shade_entry == 0xFFFFFFFFu) return 0u;  // shade_index None: nothing runs (:1291, :771, :1651)
    // the jump code is read-only for the whole launch and every fetch address is wave-uniform: through the constant
    // address space the fetches become scalar loads (scalar cache) instead of vector loads that each cost an L2 round trip
    typedef const uint32_t __attribute__((address_space(4))) *code_ptr;
    const code_ptr code = (code_ptr)P.vm_code;
    Stack st;
    st.sp = 0;
    st.tos = splat(0.0f);
    v3 deep[RXR_VM_STACK - RXR_VM_LDS_STACK];  // stack slots beyond the LDS part (scratch memory)
    v3 locals[RXR_VM_LOCALS];
    v3 globals[RXR_VM_GLOBALS];
    uint32_t fr_pc[RXR_VM_FRAMES], fr_base[RXR_VM_FRAMES], fr_lbase[RXR_VM_FRAMES], fr_llen[RXR_VM_FRAMES];
    uint32_t loop_base[RXR_VM_LOOPS];
    uint32_t nframes = 0, nloops = 0, lbase = 0, llen = prog.shade_locals, pc = prog.shade_entry, fault = 0, steps = 0;
    bool has_ret = false;
    v3 ret = splat(0.0f);
    if (llen > RXR_VM_LOCALS) {
        fault = VMF_LOCALS_OVERFLOW;
        pc = VM_PC_STOPPED;
        llen = 0;
    }
    for (uint32_t i = 0; i < llen; ++i) locals[i] = splat(0.0f);
    for (uint32_t i = 0; i < prog.n_globals && i < RXR_VM_GLOBALS; ++i) globals[i] = splat(0.0f);

    for (;;) {
        const unsigned long long running_mask = __builtin_amdgcn_ballot_w64(pc != VM_PC_STOPPED);
        if (running_mask == 0ull) break;
        // the instruction every lane at the smallest program counter executes now
        const uint32_t upc = wave_min_pc(pc, running_mask);
        const bool on = pc == upc;
        // the whole instruction (at most four words) in one go: an immediate fetched inside a handler would be a second
        // scalar-memory round trip per instruction (the stream is padded so that upc + 3 is always readable)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const u32x4 __attribute__((address_space(4), aligned(4))) *code4_ptr;
        const u32x4 insn = *(code4_ptr)(code + upc);   // one s_load_dwordx4
        const uint32_t w = insn.x, imm0 = insn.y, imm1 = insn.z, imm2 = insn.w;
        if constexpr (SSP) st.sp = (w >> 16) & 0xFFu;
        const uint32_t op = w & 0xFFu;
        // One instruction, for the opcode K known at COMPILE time: the switch below folds to a single case in each of its
        // instantiations, and vm_dispatch() reaches the right one through a binary tree of wave-uniform comparisons.  Why not
        // a plain `switch (op)`: its lowered form is one region with a hundred conditional branches and a few divergent ones
        // inside the handlers, which the compiler's control-flow structurizer linearises as a whole -- every handler then
        // hands the complete interpreter state (30-odd VGPRs) through "flow" blocks, 30 to 60 register moves per VM
        // instruction (two thirds of its VALU work).  In the tree every node is a region with ONE conditional branch, which
        // the structurizer leaves alone, and a handler touches only the registers it changes.
        auto handler = [&](auto K) __attribute__((always_inline)) {
        switch (decltype(K)::value) {
            case RXR_NODE_LOAD_GLOBAL:
                if (on) {
                    if (imm0 >= prog.n_globals) VM_FAIL(VMF_GLOBAL_INDEX)
                    VM_ROOM
                    st.push(vm_lds, deep, globals[imm0]);
                    pc = upc + 2u;
                }
                break;
            case RXR_NODE_STORE_GLOBAL:
                if (on) {
                    if (imm0 >= prog.n_globals) VM_FAIL(VMF_GLOBAL_INDEX)
                    VM_NEED(1u)
                    globals[imm0] = st.pop(vm_lds, deep);
                    pc = upc + 2u;
                }
                break;
            case RXR_NODE_LOAD_LOCAL:
                if (on) {
                    if (imm0 >= llen) VM_FAIL(VMF_LOCAL_INDEX)
                    VM_ROOM
                    st.push(vm_lds, deep, locals[lbase + imm0]);
                    pc = upc + 2u;
                }
                break;
            case RXR_NODE_STORE_LOCAL:
                if (on) {
                    if (imm0 >= llen) VM_FAIL(VMF_LOCAL_INDEX)
                    VM_NEED(1u)
                    locals[lbase + imm0] = st.pop(vm_lds, deep);
                    pc = upc + 2u;
                }
                break;
            case RXR_NODE_SWAP:
                if (on) {
                    VM_NEED(2u)
                    const v3 b = st.tos, a = st.load(vm_lds, deep, st.sp - 2u);
                    st.store(vm_lds, deep, st.sp - 2u, b);
                    st.tos = a;
                    pc = upc + 1u;
                }
                break;
            case VM_GETC:  // execution.rs:134-157
                if (on) {
                    VM_NEED(1u)
                    const uint32_t enc = imm0, n = enc & 15u;
                    const v3 v = st.tos;
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
                    st.tos = k == 1u ? splat(r[0]) : (k == 2u ? mk(r[0], r[1], 0.0f) : (k == 3u ? mk(r[0], r[1], r[2]) : splat(0.0f)));
                    pc = upc + 2u;
                }
                break;
            case VM_SETC:  // :158-183
                if (on) {
                    VM_NEED(2u)
                    const uint32_t enc = imm0, n = enc & 15u;
                    const v3 value = st.tos;
                    v3 target = st.load(vm_lds, deep, st.sp - 2u);
                    const uint32_t nc = (n >= 1u && n <= 3u) ? n : 0u;
                    for (uint32_t i = 0; i < nc; ++i) {
                        uint32_t c = (enc >> (4u + 2u * i)) & 3u;
                        float f = i == 0u ? value.x : (i == 1u ? value.y : value.z);
                        if (c == 0u) target.x = f;
                        else if (c == 1u) target.y = f;
                        else if (c == 2u) target.z = f;
                    }
                    st.tos = target;
                    --st.sp;
                    pc = upc + 2u;
                }
                break;
            case RXR_NODE_PUSH:
                if (on) {
                    VM_ROOM
                    st.push(vm_lds, deep, mk(__uint_as_float(imm0), __uint_as_float(imm1), __uint_as_float(imm2)));
                    pc = upc + 4u;
                }
                break;
            case VM_BINC: {  // "Push c; op" fused by rxr_set_shaders: the same values, no stack traffic
                // which operation: a second tree of uniform branches, OUTSIDE the divergent part (inside `if (on)` the
                // structurizer would turn a switch into a chain of thirteen masked blocks)
                if (on) {
                    VM_NEED(1u)
                }
                const v3 a = st.tos, b = mk(__uint_as_float(imm0), __uint_as_float(imm1), __uint_as_float(imm2));
                v3 r = a;
                auto fused = [&](auto J) __attribute__((always_inline)) {
                    switch (decltype(J)::value) {
                        case VM_BINC_ADD: r = mk(a.x + b.x, a.y + b.y, a.z + b.z); break;
                        case VM_BINC_SUB: r = mk(a.x - b.x, a.y - b.y, a.z - b.z); break;
                        case VM_BINC_MUL: r = mk(a.x * b.x, a.y * b.y, a.z * b.z); break;
                        case VM_BINC_DIV: r = mk(a.x / b.x, a.y / b.y, a.z / b.z); break;
                        case VM_BINC_MIN: r = mk(rust_min(a.x, b.x), rust_min(a.y, b.y), rust_min(a.z, b.z)); break;
                        case VM_BINC_MAX: r = mk(rust_max(a.x, b.x), rust_max(a.y, b.y), rust_max(a.z, b.z)); break;
                        case VM_BINC_MOD: r = mk(a.x - b.x * floorf(a.x / b.x), a.y - b.y * floorf(a.y / b.y), a.z - b.z * floorf(a.z / b.z)); break;
                        case VM_BINC_LT: r = splat(a.x < b.x ? 1.0f : 0.0f); break;
                        case VM_BINC_LE: r = splat(a.x <= b.x ? 1.0f : 0.0f); break;
                        case VM_BINC_GT: r = splat(a.x > b.x ? 1.0f : 0.0f); break;
                        case VM_BINC_GE: r = splat(a.x >= b.x ? 1.0f : 0.0f); break;
                        case VM_BINC_EQ: r = splat(a.x == b.x ? 1.0f : 0.0f); break;
                        default: r = splat(a.x != b.x ? 1.0f : 0.0f); break;  // VM_BINC_NE
                    }
                };
                const uint32_t which = (w >> 8) & 0xFFu;
                vm_dispatch_even<0u, (uint32_t)VM_BINC_COUNT>(which < (uint32_t)VM_BINC_COUNT ? which : (uint32_t)VM_BINC_NE, fused);
                if (on) {
                    st.tos = r;
                    pc = upc + 4u;
                }
                break;
            }
            case RXR_NODE_CLEAR:
                if (on) {
                    if (st.sp) (void)st.pop(vm_lds, deep);
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_DUP:
                if (on) {
                    if (st.sp) {
                        VM_ROOM
                        st.push(vm_lds, deep, st.tos);
                    }
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_PACK2: VM_BIN(mk(a.x, b.x, 0.0f))
            case RXR_NODE_PACK3: VM_TER(mk(a.x, b.x, c.x))
            // ---- control flow (flattened If / For / FunctionCall / Return)
            case VM_JMP:
                if (on) {
                    if (imm0 <= upc) {  // the only way back in `shade`'s own code: a For loop's closing jump
                        if (++steps > RXR_VM_MAX_STEPS) VM_FAIL(VMF_STEP_LIMIT)
                    }
                    pc = imm0;
                }
                break;
            case VM_JZ:
                if (on) {
                    VM_NEED(1u)
                    const v3 c = st.pop(vm_lds, deep);
                    pc = (c.x != 0.0f) ? upc + 2u : imm0;
                }
                break;
            case VM_FOR_ENTER:
                if (on) {
                    if (nloops >= RXR_VM_LOOPS) VM_FAIL(VMF_LOOP_DEPTH)
                    loop_base[nloops++] = st.sp;
                    pc = upc + 1u;
                }
                break;
            case VM_FOR_TRUNC:
                if (on) {
                    st.truncate(vm_lds, deep, loop_base[nloops - 1u]);
                    pc = upc + 1u;
                }
                break;
            case VM_FOR_COND:
                if (on) {
                    VM_NEED(1u)
                    const v3 z = st.pop(vm_lds, deep);
                    pc = (z.x == 0.0f) ? imm0 : upc + 2u;
                }
                break;
            case VM_FOR_EXIT:
                if (on) {
                    --nloops;
                    pc = upc + 1u;
                }
                break;
            case VM_CALL:  // :186-223
                if (on) {
                    const uint32_t arity = imm0, total = imm1, target = imm2;
                    if (nframes >= RXR_VM_FRAMES) VM_FAIL(VMF_CALL_DEPTH)
                    if (++steps > RXR_VM_MAX_STEPS) VM_FAIL(VMF_STEP_LIMIT)
                    const uint32_t nb = lbase + llen;
                    if (nb + total > RXR_VM_LOCALS) VM_FAIL(VMF_LOCALS_OVERFLOW)
                    for (uint32_t i = 0; i < total; ++i) locals[nb + i] = splat(0.0f);
                    bool bad = false;
                    for (uint32_t i = arity; i-- > 0u;) {
                        if (st.sp) {
                            if (i >= total) {
                                bad = true;
                                break;
                            }
                            locals[nb + i] = st.pop(vm_lds, deep);
                        }
                    }
                    if (bad) VM_FAIL(VMF_LOCAL_INDEX)
                    fr_pc[nframes] = upc + 4u;
                    fr_base[nframes] = st.sp;
                    fr_lbase[nframes] = lbase;
                    fr_llen[nframes] = llen;
                    ++nframes;
                    lbase = nb;
                    llen = total;
                    pc = target;
                }
                break;
            case VM_RETURN:  // :224-234
                if (on) {
                    v3 v;
                    if (st.sp) v = st.pop(vm_lds, deep);
                    else if (has_ret) v = ret;
                    else v = splat(0.0f);
                    ret = v;
                    has_ret = true;
                    pc = imm0;  // the function's VM_ENDFN
                }
                break;
            case VM_ENDFN:
                if (on) {
                    if (nframes == 0u) {  // end of `shade`
                        pc = VM_PC_STOPPED;
                        break;
                    }
                    --nframes;
                    const uint32_t base = fr_base[nframes];
                    v3 r;
                    if (has_ret) {
                        r = ret;
                        has_ret = false;
                    } else if (st.sp > base) {
                        r = st.pop(vm_lds, deep);
                    } else {
                        r = splat(0.0f);
                    }
                    st.truncate(vm_lds, deep, base);
                    lbase = fr_lbase[nframes];
                    llen = fr_llen[nframes];
                    pc = fr_pc[nframes];
                    VM_ROOM
                    st.push(vm_lds, deep, r);
                }
                break;
            case VM_FAULT:
                if (on) VM_FAIL(imm0)
                break;
            // ---- arithmetic
            case RXR_NODE_ADD: VM_BIN(mk(a.x + b.x, a.y + b.y, a.z + b.z))
            case RXR_NODE_SUB: VM_BIN(mk(a.x - b.x, a.y - b.y, a.z - b.z))
            case RXR_NODE_MUL: VM_BIN(mk(a.x * b.x, a.y * b.y, a.z * b.z))
            case RXR_NODE_DIV: VM_BIN(mk(a.x / b.x, a.y / b.y, a.z / b.z))
            case RXR_NODE_LENGTH: VM_UN(splat(sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z)))
            case RXR_NODE_LENGTH2: VM_UN(mk(sqrtf(a.x * a.x + a.y * a.y), 0.0f, 0.0f))
            case RXR_NODE_LENGTH3: VM_UN(mk(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), 0.0f, 0.0f))
            case RXR_NODE_ABS: VM_UN(mk(fabsf(a.x), fabsf(a.y), fabsf(a.z)))
            case RXR_NODE_SIN:
            case RXR_NODE_SIN1:
            case RXR_NODE_SIN2:
            case RXR_NODE_COS:
            case RXR_NODE_COS1:
            case RXR_NODE_COS2:
            case RXR_NODE_TAN:
            case RXR_NODE_ATAN:
            case RXR_NODE_LOG: VM_UN(slow_unary(w & 0xFFu, a))
            case RXR_NODE_ATAN2:
            case RXR_NODE_POW:
            case RXR_NODE_ROTATE2D: VM_BIN(slow_binary(w & 0xFFu, a, b))
            case RXR_NODE_NORMALIZE:  // :345-353
                if (on) {
                    VM_NEED(1u)
                    const v3 a = st.tos;
                    float len = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
                    if (len > 0.0f) st.tos = mk(a.x / len, a.y / len, a.z / len);
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_DOT: VM_BIN(splat((a.x * b.x + a.y * b.y) + a.z * b.z))
            case RXR_NODE_DOT2: VM_BIN(mk(a.x * b.x + a.y * b.y, 0.0f, 0.0f))
            case RXR_NODE_DOT3: VM_BIN(mk(a.x * b.x + a.y * b.y + a.z * b.z, 0.0f, 0.0f))
            case RXR_NODE_CROSS: VM_BIN(mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x))
            case RXR_NODE_FLOOR: VM_UN(mk(floorf(a.x), floorf(a.y), floorf(a.z)))
            case RXR_NODE_CEIL: VM_UN(mk(ceilf(a.x), ceilf(a.y), ceilf(a.z)))
            case RXR_NODE_ROUND: VM_UN(mk(roundf(a.x), roundf(a.y), roundf(a.z)))
            case RXR_NODE_FRACT: VM_UN(mk(a.x - floorf(a.x), a.y - floorf(a.y), a.z - floorf(a.z)))
            case RXR_NODE_MOD: VM_BIN(mk(a.x - b.x * floorf(a.x / b.x), a.y - b.y * floorf(a.y / b.y), a.z - b.z * floorf(a.z / b.z)))
            case RXR_NODE_RADIANS: VM_UN(mk(a.x * (3.14159265358979323846f / 180.0f), a.y * (3.14159265358979323846f / 180.0f), a.z * (3.14159265358979323846f / 180.0f)))
            case RXR_NODE_DEGREES: VM_UN(mk(a.x * 57.2957795130823208767981548141051703f, a.y * 57.2957795130823208767981548141051703f, a.z * 57.2957795130823208767981548141051703f))
            case RXR_NODE_MIN: VM_BIN(mk(rust_min(a.x, b.x), rust_min(a.y, b.y), rust_min(a.z, b.z)))
            case RXR_NODE_MAX: VM_BIN(mk(rust_max(a.x, b.x), rust_max(a.y, b.y), rust_max(a.z, b.z)))
            case RXR_NODE_MIX: VM_TER(mk(a.x + (b.x - a.x) * c.x, a.y + (b.y - a.y) * c.y, a.z + (b.z - a.z) * c.z))
            case RXR_NODE_SMOOTHSTEP:  // :456-474: a = edge0, b = edge1, c = x
                if (on) {
                    VM_NEED(3u)
                    const v3 c = st.tos, b = st.load(vm_lds, deep, st.sp - 2u), a = st.load(vm_lds, deep, st.sp - 3u);
                    float denom = b.x - a.x;
                    float t = denom != 0.0f ? (c.x - a.x) / denom : 0.0f;
                    if (t < 0.0f) t = 0.0f;
                    else if (t > 1.0f) t = 1.0f;
                    st.tos = splat(t * t * (3.0f - 2.0f * t));
                    st.sp -= 2u;
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_STEP: VM_BIN(mk(b.x >= a.x ? 1.0f : 0.0f, b.y >= a.y ? 1.0f : 0.0f, b.z >= a.z ? 1.0f : 0.0f))
            case RXR_NODE_CLAMP:  // f32::clamp panics unless min <= max: a = x, b = lo, c = hi
                if (on) {
                    VM_NEED(3u)
                    const v3 c = st.tos, b = st.load(vm_lds, deep, st.sp - 2u), a = st.load(vm_lds, deep, st.sp - 3u);
                    if (!(b.x <= c.x) || !(b.y <= c.y) || !(b.z <= c.z)) VM_FAIL(VMF_CLAMP_BOUNDS)
                    st.tos = mk(rclampf(a.x, b.x, c.x), rclampf(a.y, b.y, c.y), rclampf(a.z, b.z, c.z));
                    st.sp -= 2u;
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_SQRT: VM_UN(mk(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)))
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
            case RXR_NODE_PRINT:  // println! only
                if (on) {
                    VM_NEED(1u)
                    (void)st.pop(vm_lds, deep);
                    pc = upc + 1u;
                }
                break;
            // ---- the fragment's fields
            case RXR_NODE_UV: VM_GET(io.uv)
            case RXR_NODE_SET_UV: VM_SET(io.uv)
            case RXR_NODE_NORMAL: VM_GET(io.normal)
            case RXR_NODE_SET_NORMAL:  // .normalized()
                if (on) {
                    VM_NEED(1u)
                    const v3 a = st.pop(vm_lds, deep);
                    float len = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
                    io.normal = mk(a.x / len, a.y / len, a.z / len);
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_HITPOINT: VM_GET(io.hitpoint)
            case RXR_NODE_TIME: VM_GET(io.time)
            case RXR_NODE_COLOR: VM_GET(io.color)
            case RXR_NODE_SET_COLOR: VM_SET(io.color)
            case RXR_NODE_ROUGHNESS: VM_GET(io.roughness)
            case RXR_NODE_SET_ROUGHNESS: VM_SET(io.roughness)
            case RXR_NODE_METALLIC: VM_GET(io.metallic)
            case RXR_NODE_SET_METALLIC: VM_SET(io.metallic)
            case RXR_NODE_EMISSIVE: VM_GET(io.emissive)
            case RXR_NODE_SET_EMISSIVE: VM_SET(io.emissive)
            case RXR_NODE_OPACITY: VM_GET(io.opacity)
            case RXR_NODE_SET_OPACITY: VM_SET(io.opacity)
            case RXR_NODE_BUMP: VM_GET(io.bump)
            case RXR_NODE_SET_BUMP: VM_SET(io.bump)
            case RXR_NODE_SAMPLE:  // :570-578: a = uv, b = pattern id
                if (on) {
                    VM_NEED(2u)
                    const v3 b = st.tos, a = st.load(vm_lds, deep, st.sp - 2u);
                    const uint32_t id = as_usize_sat(b.x);
                    st.tos = id < P.n_patterns ? pattern_sample(P, P.patterns[id], a) : splat(0.0f);
                    --st.sp;
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_SAMPLE_NORMAL:  // :579-594
                if (on) {
                    VM_NEED(2u)
                    const v3 b = st.tos, a = st.load(vm_lds, deep, st.sp - 2u);
                    const uint32_t id = as_usize_sat(b.x);
                    v3 o = splat(0.0f);
                    if (id < P.n_normal_patterns) {
                        v3 nm = pattern_sample(P, P.patterns[P.n_patterns + id], a);
                        o = mk(nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, nm.z * 2.0f - 1.0f);
                    }
                    st.tos = o;
                    --st.sp;
                    pc = upc + 1u;
                }
                break;
            case RXR_NODE_PALETTE_INDEX:  // :694-701: pushes nothing for a missing / empty slot
                if (on) {
                    VM_NEED(1u)
                    const v3 a = st.pop(vm_lds, deep);
                    const uint32_t id = as_usize_sat(a.x);
                    if (id < P.n_palette && P.palette[4u * id + 3u] != 0.0f)
                        st.push(vm_lds, deep, mk(P.palette[4u * id], P.palette[4u * id + 1u], P.palette[4u * id + 2u]));  // (room: one was just popped)
                    pc = upc + 1u;
                }
                break;
            default:
                if (on) VM_FAIL(VMF_BAD_OPCODE)
                break;
        }
        };
        if (op < (uint32_t)RXR_NODE_COUNT) vm_dispatch<0u, (uint32_t)RXR_NODE_COUNT>(op, handler);
        else if (op >= (uint32_t)VM_JMP && op <= (uint32_t)VM_BINC) vm_dispatch<(uint32_t)VM_JMP, (uint32_t)VM_BINC + 1u>(op, handler);
        else if (on) {
            fault = VMF_BAD_OPCODE;
            pc = VM_PC_STOPPED;
        }
    }
    if (fault) *P.vm_fault = fault;
    return fault;
}

// this workgroup's LDS block for the value stacks
__device__ __forceinline__ float *stack_block() {
    __shared__ float vm_lds[RXR_VM_LDS_STACK * 3 * RXR_TILE_THREADS];
    return vm_lds;
}

// out-of-line form for the rarer call sites (opacity pass, 2D pass, the visibility loop's alpha test): one shared copy
// of the interpreter instead of one per site
template <bool SSP>
__device__ __noinline__ uint32_t shade_call(const RasterParams &P, uint32_t pi, IO &io_caller) {
    IO io = io_caller;  // the caller's copy sits in scratch (its address crosses the call): work on registers, write back once
    const uint32_t fault = shade_inline<SSP>(P, pi, io, stack_block());
    io_caller = io;
    return fault;
}

#undef VM_FAIL
#undef VM_NEED
#undef VM_ROOM
#undef VM_UN
#undef VM_BIN
#undef VM_TER
#undef VM_GET
#undef VM_SET
#endif  // RXR_JIT

}  // namespace rxvm
