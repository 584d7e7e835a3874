// rxr_jit.hip -- Rusteria program sets compiled at run time (opt-in: RXR_SHADER_JIT=1).
//
// Why: the interpreter (rxr_vm.h) costs about a hundred SIMD cycles per VM instruction and wave -- fetch, a tree of scalar
// branches, the value stack in LDS -- and its state bounds the program kernels to 80 VGPRs with spills; the 1 M-triangle grid with
// the per-batch program of BASELINE.json's configuration C5 spends more time interpreting sixteen instructions per fragment than
// rasterising.  For a set whose stack depths are static (no calls, no PaletteIndex: what rxr_set_shaders proves with
// tag_static_depths, the common case of colour / material programs) the jump code IS a register program: the depth before every
// instruction is known, so stack slot k is the variable s<k>, locals and globals are variables, If / For are gotos, and a lane
// that diverges is the compiler's business.  This file turns the SAME jump code the interpreter would run into that C++
// (rxr_jit_generate), compiles the raster kernel around it with hiprtc (the kernel sources are embedded in the library:
// rxr_jit_embedded.inc, written by __graft_entry__.build()), loads the code object and launches it in place of k_raster_vm*.
// The operations are the interpreter's own expressions (rxr_jit_ops.h); parity with the interpreter and the CPU oracle is
// tests/test_gpu_shader_jit.py.  Everything else -- validation, purity analysis, flattening, refusals -- is unchanged and runs
// first; a set that is not eligible, or a compiler failure, leaves the interpreter in charge (the reason is kept in the context).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "rxr_ctx.h"

#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

extern char **environ;

namespace {

#include "rxr_jit_embedded.inc"  // rxr_jit_header_names[], rxr_jit_header_sources[], rxr_jit_n_headers

// hiprtc is looked up when the first set is compiled, not linked: the library must load (and interpret programs) on an
// installation without the run-time compiler
struct Rtc {
    decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
    decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
    decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
    decltype(&hiprtcGetCode) GetCode = nullptr;
    decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
    decltype(&hiprtcGetErrorString) GetErrorString = nullptr;
    decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
    decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
    std::string why;
    bool ok = false;
};
const Rtc &rtc() {
    static const Rtc r = [] {
        Rtc x;
        void *h = nullptr;
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) {
            x.why = "libhiprtc.so not found";
            return x;
        }
        auto sym = [&](auto &fn, const char *name) {
            fn = reinterpret_cast<typename std::remove_reference<decltype(fn)>::type>(dlsym(h, name));
            return fn != nullptr;
        };
        x.ok = sym(x.CreateProgram, "hiprtcCreateProgram") && sym(x.CompileProgram, "hiprtcCompileProgram") && sym(x.DestroyProgram, "hiprtcDestroyProgram") &&
               sym(x.GetCode, "hiprtcGetCode") && sym(x.GetCodeSize, "hiprtcGetCodeSize") && sym(x.GetErrorString, "hiprtcGetErrorString") &&
               sym(x.GetProgramLog, "hiprtcGetProgramLog") && sym(x.GetProgramLogSize, "hiprtcGetProgramLogSize");
        if (!x.ok) x.why = "libhiprtc.so lacks an entry point";
        return x;
    }();
    return r;
}

uint32_t length_of(uint32_t op) {
    switch (op) {
        case RXR_NODE_LOAD_GLOBAL: case RXR_NODE_STORE_GLOBAL: case RXR_NODE_LOAD_LOCAL: case RXR_NODE_STORE_LOCAL:
        case VM_GETC: case VM_SETC: case VM_JMP: case VM_JZ: case VM_FOR_COND: case VM_RETURN: case VM_FAULT: return 2;
        case RXR_NODE_PUSH: case VM_BINC: case VM_CALL: return 4;
        default: return 1;
    }
}

std::string slot(int k) { return "s" + std::to_string(k); }
std::string hexw(uint32_t w) {
    char b[16];
    snprintf(b, sizeof b, "0x%08xu", w);
    return b;
}
std::string imm3(const std::vector<uint32_t> &code, uint32_t pc) {
    return "mk(__uint_as_float(" + hexw(code[pc + 1]) + "), __uint_as_float(" + hexw(code[pc + 2]) + "), __uint_as_float(" + hexw(code[pc + 3]) + "))";
}

enum Arity { A_NONE, A_UN, A_BIN, A_TER };
Arity pure_arity(uint32_t op) {
    switch (op) {
        case RXR_NODE_LENGTH: case RXR_NODE_LENGTH2: case RXR_NODE_LENGTH3: case RXR_NODE_ABS: case RXR_NODE_SIN: case RXR_NODE_SIN1:
        case RXR_NODE_SIN2: case RXR_NODE_COS: case RXR_NODE_COS1: case RXR_NODE_COS2: case RXR_NODE_TAN: case RXR_NODE_ATAN:
        case RXR_NODE_NORMALIZE: case RXR_NODE_FLOOR: case RXR_NODE_CEIL: case RXR_NODE_ROUND: case RXR_NODE_FRACT:
        case RXR_NODE_DEGREES: case RXR_NODE_RADIANS: case RXR_NODE_SQRT: case RXR_NODE_LOG: case RXR_NODE_NOT: case RXR_NODE_NEG:
            return A_UN;
        case RXR_NODE_PACK2: case RXR_NODE_ADD: case RXR_NODE_SUB: case RXR_NODE_MUL: case RXR_NODE_DIV: case RXR_NODE_ATAN2:
        case RXR_NODE_ROTATE2D: case RXR_NODE_DOT: case RXR_NODE_DOT2: case RXR_NODE_DOT3: case RXR_NODE_CROSS: case RXR_NODE_MOD:
        case RXR_NODE_MIN: case RXR_NODE_MAX: case RXR_NODE_STEP: case RXR_NODE_POW: case RXR_NODE_EQ: case RXR_NODE_NE: case RXR_NODE_LT:
        case RXR_NODE_LE: case RXR_NODE_GT: case RXR_NODE_GE: case RXR_NODE_AND: case RXR_NODE_OR:
            return A_BIN;
        case RXR_NODE_PACK3: case RXR_NODE_MIX: case RXR_NODE_SMOOTHSTEP: case RXR_NODE_CLAMP:
            return A_TER;
        default: return A_NONE;
    }
}
const char *field_of(uint32_t op, bool &is_set) {
    is_set = false;
    switch (op) {
        case RXR_NODE_UV: return "io.uv";
        case RXR_NODE_SET_UV: is_set = true; return "io.uv";
        case RXR_NODE_NORMAL: return "io.normal";
        case RXR_NODE_HITPOINT: return "io.hitpoint";
        case RXR_NODE_TIME: return "io.time";
        case RXR_NODE_COLOR: return "io.color";
        case RXR_NODE_SET_COLOR: is_set = true; return "io.color";
        case RXR_NODE_ROUGHNESS: return "io.roughness";
        case RXR_NODE_SET_ROUGHNESS: is_set = true; return "io.roughness";
        case RXR_NODE_METALLIC: return "io.metallic";
        case RXR_NODE_SET_METALLIC: is_set = true; return "io.metallic";
        case RXR_NODE_EMISSIVE: return "io.emissive";
        case RXR_NODE_SET_EMISSIVE: is_set = true; return "io.emissive";
        case RXR_NODE_OPACITY: return "io.opacity";
        case RXR_NODE_SET_OPACITY: is_set = true; return "io.opacity";
        case RXR_NODE_BUMP: return "io.bump";
        case RXR_NODE_SET_BUMP: is_set = true; return "io.bump";
        default: return nullptr;
    }
}

// ---- analysis: the stack depth before every instruction of one function, RELATIVE to the depth at its entry --------------------
// The same abstract interpretation as tag_static_depths (rxr_api.hip), with calls: a call needs its `arity` arguments on the
// caller's own part of the stack, pops them and leaves one value.  A function that could touch the stack BELOW its entry depth --
// Dup / Clear / Return at relative depth 0 (they look at the absolute depth: the reference shares one stack between caller and
// callee) -- is left to the interpreter.  (Recursion: generate_program_levels.)  Depths live in a side table: the code words are not touched.
struct Call {
    uint32_t pc, target, arity, total;
};
struct Fn {
    uint32_t entry = 0, arity = 0, total = 0;   // (`shade`: arity 0, total = shade_locals)
    bool is_shade = false;
    int level = -1;                             // recursive programs (generate_program_levels): the call depth this copy runs at, else -1
    std::vector<int> depth;                     // per pc, -1: unreachable
    int max_depth = 0;
    std::vector<Call> calls;
};

bool analyse(const std::vector<uint32_t> &code, Fn &f, std::string &why) {
    struct State {
        int depth;
        std::vector<int> loops;
        bool operator==(const State &o) const { return depth == o.depth && loops == o.loops; }
    };
    f.depth.assign(code.size(), -1);
    std::vector<State> states(code.size());
    std::vector<std::pair<uint32_t, State>> work{{f.entry, State{0, {}}}};
    while (!work.empty()) {
        auto [pc, st] = work.back();
        work.pop_back();
        for (;;) {
            if (pc >= code.size()) { why = "jump code runs past its end"; return false; }
            const uint32_t op = code[pc] & 0xFFu;
            if (f.depth[pc] >= 0) {
                if (!(states[pc] == st)) { why = "two paths reach an instruction with different stacks"; return false; }
                break;
            }
            if (st.depth < 0 || st.depth > 200) { why = "stack depth out of range"; return false; }
            f.depth[pc] = st.depth;
            states[pc] = st;
            f.max_depth = std::max(f.max_depth, st.depth + 1);
            if (op == VM_ENDFN) break;
            const uint32_t len = length_of(op);
            if (pc + len > code.size()) { why = "truncated instruction"; return false; }
            int need = 0, delta = 0;
            bool is_set = false;
            const Arity ar = pure_arity(op);
            if (ar == A_UN) need = 1;
            else if (ar == A_BIN) { need = 2; delta = -1; }
            else if (ar == A_TER) { need = 3; delta = -2; }
            else if (field_of(op, is_set)) { need = is_set ? 1 : 0; delta = is_set ? -1 : 1; }
            else switch (op) {
                case RXR_NODE_LOAD_GLOBAL: case RXR_NODE_LOAD_LOCAL: case RXR_NODE_PUSH: delta = 1; break;
                case RXR_NODE_STORE_GLOBAL: case RXR_NODE_STORE_LOCAL: case RXR_NODE_PRINT: case RXR_NODE_SET_NORMAL: case VM_JZ: case VM_FOR_COND:
                    need = 1; delta = -1; break;
                case RXR_NODE_SWAP: need = 2; break;
                case VM_GETC: case VM_BINC: need = 1; break;
                case VM_SETC: case RXR_NODE_SAMPLE: case RXR_NODE_SAMPLE_NORMAL: need = 2; delta = -1; break;
                // PaletteIndex pops the index and pushes the colour -- unless the slot is missing or empty, when the reference pushes
                // nothing (execution.rs:742-749).  Compiled as the push; a fragment that meets the other case raises
                // VMF_JIT_PALETTE_MISS and the whole set goes back to the interpreter (rxr_synchronize), whose stack is dynamic.
                case RXR_NODE_PALETTE_INDEX: need = 1; break;
                case RXR_NODE_CLEAR:
                    if (st.depth == 0 && !f.is_shade) { why = "Clear on a callee's empty stack (it would pop the caller's value)"; return false; }
                    delta = st.depth > 0 ? -1 : 0;
                    break;
                case RXR_NODE_DUP:
                    if (st.depth == 0 && !f.is_shade) { why = "Dup on a callee's empty stack (it would copy the caller's value)"; return false; }
                    delta = st.depth > 0 ? 1 : 0;
                    break;
                case VM_RETURN:
                    if (st.depth == 0 && !f.is_shade) { why = "Return on a callee's empty stack (it would pop the caller's value)"; return false; }
                    break;
                case VM_CALL: {
                    const Call c{pc, code[pc + 3], code[pc + 1], code[pc + 2]};
                    need = (int)c.arity;
                    delta = 1 - (int)c.arity;
                    f.calls.push_back(c);
                    break;
                }
                case VM_JMP: case VM_FOR_ENTER: case VM_FOR_TRUNC: case VM_FOR_EXIT: case VM_FAULT: break;
                default: why = "opcode " + std::to_string(op) + " is not covered by the run-time compiler"; return false;  // PaletteIndex, unknown
            }
            if (st.depth < need) { why = "an instruction without its operands (the interpreter reports the underflow)"; return false; }
            State nx = st;
            nx.depth += delta;
            if (op == VM_FOR_ENTER) {
                if (nx.loops.size() >= RXR_VM_LOOPS) { why = "loops nested too deeply"; return false; }
                nx.loops.push_back(st.depth);
            } else if (op == VM_FOR_TRUNC) {
                if (nx.loops.empty()) { why = "loop bookkeeping"; return false; }
                nx.depth = std::min(nx.depth, nx.loops.back());
            } else if (op == VM_FOR_EXIT) {
                if (nx.loops.empty()) { why = "loop bookkeeping"; return false; }
                nx.loops.pop_back();
            }
            if (op == VM_FAULT || op == VM_RETURN) break;  // (Return: straight to the function's end, its value in hand)
            if (op == VM_JMP) { pc = code[pc + 1]; st = nx; continue; }
            if (op == VM_JZ || op == VM_FOR_COND) work.push_back({code[pc + 1], nx});
            pc += len;
            st = nx;
        }
    }
    return true;
}

std::string fn_name(uint32_t prog, const Fn &f) {
    return f.is_shade ? "rxr_jit_prog_" + std::to_string(prog)
                      : "rxr_jit_fn_" + std::to_string(prog) + "_" + std::to_string(f.entry) + "_" + std::to_string(f.arity) + "_" + std::to_string(f.total) +
                            (f.level >= 0 ? "_L" + std::to_string(f.level) : std::string());
}

// the body of one function: the reachable instructions in address order, each under its label
bool emit_body(const std::vector<uint32_t> &code, const DevProgram &p, uint32_t prog, const Fn &f, std::string &out, std::string &why) {
    const uint32_t n_locals = f.is_shade ? p.shade_locals : f.total;
    for (int k = 0; k <= f.max_depth; ++k) out += "    v3 " + slot(k) + " = splat(0.0f);\n";
    for (uint32_t k = 0; k < n_locals; ++k)
        out += "    v3 l" + std::to_string(k) + " = " + (!f.is_shade && k < f.arity ? "a" + std::to_string(k) : std::string("splat(0.0f)")) + ";\n";
    const std::string leave = f.is_shade ? "goto Lend;" : "return splat(0.0f);";
    auto fail = [&](uint32_t code_) { return "{ fault = " + std::to_string(code_) + "u; " + leave + " }"; };
    for (uint32_t pc = f.entry; pc < code.size(); ++pc) {
        if (f.depth[pc] < 0) continue;  // (dead code behind a jump, or the immediates of the instruction before)
        const uint32_t w = code[pc], op = w & 0xFFu;
        const int d = f.depth[pc];  // depth BEFORE the instruction, relative to the function's entry
        const uint32_t len = length_of(op);
        out += "L" + std::to_string(pc) + ": ";
        const std::string t0 = slot(d - 1), t1 = slot(d - 2), t2 = slot(d - 3), top = slot(d);
        bool is_set = false;
        const Arity ar = pure_arity(op);
        if (op == RXR_NODE_CLAMP) {
            out += "{ if (!jit_clamp_ok(" + t1 + ", " + t0 + ")) " + fail(VMF_CLAMP_BOUNDS) + " " + t2 + " = jit_ter<" + std::to_string(op) + "u>(" + t2 + ", " + t1 + ", " + t0 + "); }\n";
        } else if (ar == A_UN) {
            out += t0 + " = jit_un<" + std::to_string(op) + "u>(" + t0 + ");\n";
        } else if (ar == A_BIN) {
            out += t1 + " = jit_bin<" + std::to_string(op) + "u>(" + t1 + ", " + t0 + ");\n";
        } else if (ar == A_TER) {
            out += t2 + " = jit_ter<" + std::to_string(op) + "u>(" + t2 + ", " + t1 + ", " + t0 + ");\n";
        } else if (const char *fld = field_of(op, is_set)) {
            out += is_set ? std::string(fld) + " = " + t0 + ";\n" : top + " = " + fld + ";\n";
        } else {
            switch (op) {
                case RXR_NODE_PUSH: out += top + " = " + imm3(code, pc) + ";\n"; break;
                case VM_BINC: out += t0 + " = jit_binc<" + std::to_string((w >> 8) & 0xFFu) + "u>(" + t0 + ", " + imm3(code, pc) + ");\n"; break;
                case RXR_NODE_LOAD_LOCAL:
                    out += code[pc + 1] < n_locals ? top + " = l" + std::to_string(code[pc + 1]) + ";\n" : fail(VMF_LOCAL_INDEX) + "\n";
                    break;
                case RXR_NODE_STORE_LOCAL:
                    out += code[pc + 1] < n_locals ? "l" + std::to_string(code[pc + 1]) + " = " + t0 + ";\n" : fail(VMF_LOCAL_INDEX) + "\n";
                    break;
                case RXR_NODE_LOAD_GLOBAL:
                    out += code[pc + 1] < p.n_globals ? top + " = G.g" + std::to_string(code[pc + 1]) + ";\n" : fail(VMF_GLOBAL_INDEX) + "\n";
                    break;
                case RXR_NODE_STORE_GLOBAL:
                    out += code[pc + 1] < p.n_globals ? "G.g" + std::to_string(code[pc + 1]) + " = " + t0 + ";\n" : fail(VMF_GLOBAL_INDEX) + "\n";
                    break;
                case RXR_NODE_SWAP: out += "{ const v3 t = " + t0 + "; " + t0 + " = " + t1 + "; " + t1 + " = t; }\n"; break;
                case VM_GETC: out += t0 + " = jit_getc(" + hexw(code[pc + 1]) + ", " + t0 + ");\n"; break;
                case VM_SETC: out += t1 + " = jit_setc(" + hexw(code[pc + 1]) + ", " + t1 + ", " + t0 + ");\n"; break;
                case RXR_NODE_CLEAR: case RXR_NODE_PRINT: case VM_FOR_ENTER: case VM_FOR_TRUNC: case VM_FOR_EXIT: out += ";\n"; break;  // (static depths)
                case RXR_NODE_DUP: out += d > 0 ? top + " = " + t0 + ";\n" : std::string(";\n"); break;
                case RXR_NODE_SET_NORMAL: out += "io.normal = jit_set_normal(" + t0 + ");\n"; break;
                case RXR_NODE_SAMPLE: out += t1 + " = jit_sample(P, " + t1 + ", " + t0 + ");\n"; break;
                case RXR_NODE_SAMPLE_NORMAL: out += t1 + " = jit_sample_normal(P, " + t1 + ", " + t0 + ");\n"; break;
                case RXR_NODE_PALETTE_INDEX:
                    out += "{ const uint32_t id = as_usize_sat(" + t0 + ".x); if (!(id < P.n_palette && P.palette[4u * id + 3u] != 0.0f)) " + fail(VMF_JIT_PALETTE_MISS) + " " + t0 +
                           " = mk(P.palette[4u * id], P.palette[4u * id + 1u], P.palette[4u * id + 2u]); }\n";
                    break;
                case VM_JMP:
                    if (code[pc + 1] <= pc)  // the only way back: a For loop's closing jump
                        out += "{ if (++steps > " + std::to_string((uint32_t)RXR_VM_MAX_STEPS) + "u) " + fail(VMF_STEP_LIMIT) + " goto L" + std::to_string(code[pc + 1]) + "; }\n";
                    else
                        out += "goto L" + std::to_string(code[pc + 1]) + ";\n";
                    break;
                case VM_JZ: out += "if (!(" + t0 + ".x != 0.0f)) goto L" + std::to_string(code[pc + 1]) + ";\n"; break;
                case VM_FOR_COND: out += "if (" + t0 + ".x == 0.0f) goto L" + std::to_string(code[pc + 1]) + ";\n"; break;
                case VM_RETURN: out += f.is_shade ? std::string("goto Lend;\n") : "return " + t0 + ";\n"; break;  // (a callee: relative depth >= 1, analyse())
                case VM_ENDFN:  // falling off the end: the value on top of the function's own stack, if any
                    out += f.is_shade ? std::string("goto Lend;\n") : "return " + (d > 0 ? t0 : std::string("splat(0.0f)")) + ";\n";
                    break;
                case VM_FAULT: out += fail(code[pc + 1]) + "\n"; break;
                case VM_CALL: {
                    const uint32_t arity = code[pc + 1], total = code[pc + 2], target = code[pc + 3];
                    if (arity > total) {  // an argument without a local to land in: the reference's index panic
                        out += fail(VMF_LOCAL_INDEX) + "\n";
                        break;
                    }
                    Fn callee;
                    callee.entry = target; callee.arity = arity; callee.total = total;
                    std::string args;
                    for (uint32_t i = 0; i < arity; ++i) args += ", " + slot(d - (int)arity + (int)i);  // local i = the i-th argument pushed
                    const std::string res = slot(d - (int)arity);
                    if (f.level >= 0) {
                        // a recursive program: this copy runs with f.level frames on the interpreter's frame stack and calls the copy
                        // one level up; the checks the acyclic form makes per call chain at generation time are the interpreter's
                        // own here, in its order (rxr_vm.h VM_CALL): frames, steps, locals
                        if (f.level >= (int)RXR_VM_FRAMES) {
                            out += fail(VMF_CALL_DEPTH) + "\n";
                            break;
                        }
                        callee.level = f.level + 1;
                        out += "{ if (++steps > " + std::to_string((uint32_t)RXR_VM_MAX_STEPS) + "u) " + fail(VMF_STEP_LIMIT) + " if (lbase + " + std::to_string(n_locals + total) +
                               "u > " + std::to_string((uint32_t)RXR_VM_LOCALS) + "u) " + fail(VMF_LOCALS_OVERFLOW) + " const v3 r = " + fn_name(prog, callee) +
                               "(P, io, G, fault, steps, lbase + " + std::to_string(n_locals) + "u" + args + "); if (fault) " + leave + " " + res + " = r; }\n";
                        break;
                    }
                    out += "{ if (++steps > " + std::to_string((uint32_t)RXR_VM_MAX_STEPS) + "u) " + fail(VMF_STEP_LIMIT) + " const v3 r = " + fn_name(prog, callee) +
                           "(P, io, G, fault, steps" + args + "); if (fault) " + leave + " " + res + " = r; }\n";
                    break;
                }
                default: why = "opcode " + std::to_string(op) + " is not covered by the run-time compiler"; return false;
            }
        }
        if (op == VM_ENDFN) break;  // (the end of this function; what follows belongs to other functions)
        pc += len - 1;
    }
    return true;
}

// A program whose functions call themselves (directly or around a cycle).  The interpreter bounds every call chain by its frame stack
// (RXR_VM_FRAMES), so recursion unrolls into a FINITE acyclic program: one copy of a function per call depth it can run at, a call
// in the copy of depth L going to the callee's copy of depth L + 1, and a call in a copy of the last depth being the fault the
// interpreter raises there (VMF_CALL_DEPTH).  No device-side recursion, no dynamic stack: the compiler sees an acyclic call graph
// and sizes registers and scratch as for any other program.  The locals a chain has in use depend on the path, so that check moves
// to run time (`lbase`, the interpreter's own test); the value stack is checked here against the deepest chain through every copy.
// The copies are real functions (noinline) when inlining them all would multiply the program (a body with two recursive call sites
// has 2^8 paths), inlined otherwise.
bool generate_program_levels(const std::vector<uint32_t> &code, const DevProgram &p, uint32_t index, std::string &out, std::string &why) {
    std::vector<Fn> fns;                       // [0] = shade (level 0), then copies in discovery order
    std::vector<int> stack_base;               // deepest value stack below the copy's own part, over every chain that reaches it
    std::vector<double> paths;                 // number of call chains that reach the copy (for the inlining decision)
    auto find = [&](uint32_t entry, uint32_t arity, uint32_t total, int level) {
        for (size_t i = 1; i < fns.size(); ++i)
            if (fns[i].entry == entry && fns[i].arity == arity && fns[i].total == total && fns[i].level == level) return (int)i;
        return -1;
    };
    Fn shade;
    shade.entry = p.shade_entry;
    shade.total = p.shade_locals;
    shade.is_shade = true;
    shade.level = 0;
    if (!analyse(code, shade, why)) return false;
    fns.push_back(shade);
    stack_base.push_back(0);
    paths.push_back(1.0);
    // level by level: every call goes exactly one level up, so when a level is processed the copies of that level are complete
    for (int level = 0; level < (int)RXR_VM_FRAMES; ++level) {
        const size_t n_now = fns.size();
        for (size_t i = 0; i < n_now; ++i) {
            if (fns[i].level != level) continue;
            const std::vector<Call> calls = fns[i].calls;
            for (const Call &c : calls) {
                if (c.arity > c.total) continue;  // (emitted as a static fault)
                int k = find(c.target, c.arity, c.total, level + 1);
                if (k < 0) {
                    Fn g;
                    g.entry = c.target; g.arity = c.arity; g.total = c.total; g.level = level + 1;
                    if (!analyse(code, g, why)) return false;
                    fns.push_back(g);
                    stack_base.push_back(0);
                    paths.push_back(0.0);
                    k = (int)fns.size() - 1;
                    if (fns.size() > 512u) { why = "a recursive program with more than 512 function copies"; return false; }
                }
                stack_base[(size_t)k] = std::max(stack_base[(size_t)k], stack_base[i] + fns[i].depth[c.pc] - (int)c.arity);
                paths[(size_t)k] += paths[i];
            }
        }
    }
    double expanded = 0.0;
    for (size_t i = 0; i < fns.size(); ++i) {
        if (stack_base[i] + fns[i].max_depth > (int)RXR_VM_STACK) { why = "a value stack deeper than the interpreter's"; return false; }
        uint32_t live = 0;
        for (int dpt : fns[i].depth) live += dpt >= 0 ? 1u : 0u;
        expanded += paths[i] * (double)live;
    }
    const bool inline_all = expanded <= 1024.0;  // (VM instructions after inlining every chain)
    out += "struct rxr_jit_globals_" + std::to_string(index) + " {\n";
    for (uint32_t k = 0; k < p.n_globals && k < RXR_VM_GLOBALS; ++k) out += "    rxvm::v3 g" + std::to_string(k) + ";\n";
    out += "    int unused;\n};\n";
    // callees first: the deepest level down to shade
    for (int level = (int)RXR_VM_FRAMES; level >= 0; --level)
        for (size_t i = 0; i < fns.size(); ++i) {
            const Fn &f = fns[i];
            if (f.level != level) continue;
            if (f.is_shade) {
                out += "__device__ __forceinline__ uint32_t " + fn_name(index, f) + "(const RasterParams &P, rxvm::IO &io) {\n    using namespace rxvm;\n    (void)P;\n";
                out += "    rxr_jit_globals_" + std::to_string(index) + " G;\n    G.unused = 0;\n";
                for (uint32_t g = 0; g < p.n_globals && g < RXR_VM_GLOBALS; ++g) out += "    G.g" + std::to_string(g) + " = splat(0.0f);\n";
                out += "    uint32_t fault = 0u, steps = 0u;\n    (void)steps;\n    const uint32_t lbase = 0u;\n    (void)lbase;\n";
                if (!emit_body(code, p, index, f, out, why)) return false;
                out += "Lend:\n    if (fault) *P.vm_fault = fault;\n    return fault;\n}\n";
            } else {
                out += std::string("__device__ ") + (inline_all ? "__forceinline__" : "__noinline__") + " rxvm::v3 " + fn_name(index, f) +
                       "(const RasterParams &P, rxvm::IO &io, rxr_jit_globals_" + std::to_string(index) + " &G, uint32_t &fault, uint32_t &steps, const uint32_t lbase";
                for (uint32_t a = 0; a < f.arity; ++a) out += ", rxvm::v3 a" + std::to_string(a);
                out += ") {\n    using namespace rxvm;\n    (void)P; (void)io; (void)G; (void)steps; (void)lbase;\n";
                if (!emit_body(code, p, index, f, out, why)) return false;
                out += "    return splat(0.0f);\n}\n";
            }
        }
    return true;
}

// one program: `shade` and every function it can reach (specialised per (entry, arity, total_locals) as the call sites name them),
// callees before callers.  Returns false for anything the generator does not cover; the interpreter then runs the set.
bool generate_program(const std::vector<uint32_t> &code, const DevProgram &p, uint32_t index, std::string &out, std::string &why) {
    const std::string sig = "(const RasterParams &P, rxvm::IO &io)";
    if (p.shade_entry == 0xFFFFFFFFu) {  // shade_index None: nothing runs
        out += "__device__ __forceinline__ uint32_t rxr_jit_prog_" + std::to_string(index) + sig + " {\n    (void)P;\n    (void)io;\n    return 0u;\n}\n";
        return true;
    }
    if (p.shade_locals > RXR_VM_LOCALS) {
        out += "__device__ __forceinline__ uint32_t rxr_jit_prog_" + std::to_string(index) + sig + " {\n    (void)io;\n    *P.vm_fault = " +
               std::to_string((uint32_t)VMF_LOCALS_OVERFLOW) + "u;\n    return " + std::to_string((uint32_t)VMF_LOCALS_OVERFLOW) + "u;\n}\n";
        return true;
    }
    // the functions, discovered from `shade` down; `order` ends up callees-first
    std::vector<Fn> fns;
    std::vector<int> order, state;  // state: 0 new, 1 on the walk's stack, 2 done
    auto find = [&](uint32_t entry, uint32_t arity, uint32_t total) {
        for (size_t i = 0; i < fns.size(); ++i)
            if (fns[i].entry == entry && fns[i].arity == arity && fns[i].total == total && !fns[i].is_shade) return (int)i;
        return -1;
    };
    Fn shade;
    shade.entry = p.shade_entry;
    shade.total = p.shade_locals;
    shade.is_shade = true;
    fns.push_back(shade);
    state.push_back(0);
    // depth-first: (function, chain depth, locals and stack in use above it)
    struct Walk { int fn; int frames; uint32_t locals; int stack; };
    std::vector<int> path;
    bool ok = true;
    std::function<void(const Walk &)> visit = [&](const Walk &wk) {
        if (!ok) return;
        Fn &f = fns[(size_t)wk.fn];
        if (state[(size_t)wk.fn] == 1) { ok = false; why = "recursion"; return; }
        if (state[(size_t)wk.fn] == 0) {
            if (!analyse(code, f, why)) { ok = false; return; }
        }
        // what the interpreter would refuse at run time must not be compiled away: call depth, locals, stack (checked per chain)
        if (wk.frames > (int)RXR_VM_FRAMES) { ok = false; why = "calls nested deeper than the interpreter's frame stack"; return; }
        if (wk.locals + (f.is_shade ? p.shade_locals : f.total) > RXR_VM_LOCALS) { ok = false; why = "more locals along a call chain than the interpreter holds"; return; }
        if (wk.stack + f.max_depth > (int)RXR_VM_STACK) { ok = false; why = "a value stack deeper than the interpreter's"; return; }
        const bool first = state[(size_t)wk.fn] == 0;
        state[(size_t)wk.fn] = 1;
        const std::vector<Call> calls = f.calls;  // (fns may grow below: no references into it across the loop)
        const uint32_t own_locals = f.is_shade ? p.shade_locals : f.total;
        for (const Call &c : calls) {
            if (c.arity > c.total) continue;  // (emitted as a static fault)
            int k = find(c.target, c.arity, c.total);
            if (k < 0) {
                Fn g;
                g.entry = c.target; g.arity = c.arity; g.total = c.total;
                fns.push_back(g);
                state.push_back(0);
                k = (int)fns.size() - 1;
            }
            const int depth_at_call = fns[(size_t)wk.fn].depth[c.pc] - (int)c.arity;
            visit(Walk{k, wk.frames + 1, wk.locals + own_locals, wk.stack + depth_at_call});
            if (!ok) return;
        }
        state[(size_t)wk.fn] = 2;
        if (first) order.push_back(wk.fn);
    };
    visit(Walk{0, 0, 0u, 0});
    if (!ok && why == "recursion") return generate_program_levels(code, p, index, out, why);
    if (!ok) return false;
    // globals of the program, shared by its functions
    out += "struct rxr_jit_globals_" + std::to_string(index) + " {\n";
    for (uint32_t k = 0; k < p.n_globals && k < RXR_VM_GLOBALS; ++k) out += "    rxvm::v3 g" + std::to_string(k) + ";\n";
    out += "    int unused;\n};\n";
    for (int k : order) {
        const Fn &f = fns[(size_t)k];
        if (f.is_shade) {
            out += "__device__ __forceinline__ uint32_t " + fn_name(index, f) + sig + " {\n    using namespace rxvm;\n    (void)P;\n";
            out += "    rxr_jit_globals_" + std::to_string(index) + " G;\n    G.unused = 0;\n";
            for (uint32_t g = 0; g < p.n_globals && g < RXR_VM_GLOBALS; ++g) out += "    G.g" + std::to_string(g) + " = splat(0.0f);\n";
            out += "    uint32_t fault = 0u, steps = 0u;\n    (void)steps;\n";
            if (!emit_body(code, p, index, f, out, why)) return false;
            out += "Lend:\n    if (fault) *P.vm_fault = fault;\n    return fault;\n}\n";
        } else {
            out += "__device__ __forceinline__ rxvm::v3 " + fn_name(index, f) + "(const RasterParams &P, rxvm::IO &io, rxr_jit_globals_" + std::to_string(index) +
                   " &G, uint32_t &fault, uint32_t &steps";
            for (uint32_t a = 0; a < f.arity; ++a) out += ", rxvm::v3 a" + std::to_string(a);
            out += ") {\n    using namespace rxvm;\n    (void)P; (void)io; (void)G; (void)steps;\n";
            if (!emit_body(code, p, index, f, out, why)) return false;
            out += "    return splat(0.0f);\n}\n";
        }
    }
    return true;
}

}  // namespace

// the header rxr_vm.h includes in RXR_JIT mode: one function per program and the dispatcher
bool rxr_jit_generate(const std::vector<uint32_t> &code, const std::vector<DevProgram> &progs, std::string &src, std::string &why) {
    src = "// generated by rxr_jit.hip from the jump code of one rxr_set_shaders call\n#pragma once\n";
    for (size_t i = 0; i < progs.size(); ++i)
        if (!generate_program(code, progs[i], (uint32_t)i, src, why)) {
            why = "program " + std::to_string(i) + ": " + why;
            return false;
        }
    src += "__device__ __forceinline__ uint32_t rxr_jit_shade(const RasterParams &P, uint32_t pi, rxvm::IO &io) {\n    switch (pi) {\n";
    for (size_t i = 0; i < progs.size(); ++i) src += "        case " + std::to_string(i) + "u: return rxr_jit_prog_" + std::to_string(i) + "(P, io);\n";
    src += "        default: return 0u;\n    }\n}\n";
    return true;
}

namespace {
std::mutex g_cache_mu;
std::map<std::string, std::vector<char>> g_code_objects;  // generated source -> code object (one compilation per set and process)
}  // namespace

namespace {
std::string cache_key(const std::string &arch, int level, const std::string &gen) {
    const char *flags_env = getenv("RXR_JIT_FLAGS");  // further compiler options, blank-separated (tuning runs: -DRXR_JIT_WAVES_PER_SIMD=6 ...)
    return arch + "\n" + std::to_string(level) + "\n" + (flags_env ? flags_env : "") + "\n" + gen;
}

// Background compilations (the default): a compilation in flight is a child process (rxr_jitc, next to this library) and two files
// in /tmp, registered process-wide by cache key -- the contexts of a multi-device group, or two contexts with the same set, share
// one child; at most RXR_JIT_MAX_CHILDREN run at a time (each is a few seconds of one core), the others wait their turn.
constexpr size_t RXR_JIT_MAX_CHILDREN = 2;
struct BgJob {
    std::string dir, src, out;   // a directory of its own under /tmp: the child's TMPDIR too (a killed compiler leaves its temporaries there)
    pid_t pid = 0;
    int refs = 0;
};
std::mutex g_bg_mu;
std::map<std::string, BgJob> g_bg;          // in flight
std::map<std::string, std::string> g_bg_failed;  // key -> why (a set that failed once is not tried again)

// Job directories live under ONE parent that only this user can write: $XDG_RUNTIME_DIR/rxr_jit when that variable names a
// directory of ours, else /tmp/rxr_jit-<uid>, created 0700 and checked with lstat (a real directory, ours, no group / other bits)
// before anything is put into or removed from it.  /tmp itself is world-writable: nothing directly under it is ever examined,
// opened or removed by name, so another user's entry (a symlink to somebody's directory, say) cannot steer the clean-up.
std::string job_parent(std::string &err) {
    std::string parent;
    if (const char *xdg = getenv("XDG_RUNTIME_DIR")) {
        struct stat st;
        if (xdg[0] == '/' && lstat(xdg, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 077) == 0) parent = std::string(xdg) + "/rxr_jit";
    }
    if (parent.empty()) parent = "/tmp/rxr_jit-" + std::to_string((unsigned long)geteuid());
    if (mkdir(parent.c_str(), 0700) != 0 && errno != EEXIST) {
        err = "the background compiler's directory " + parent + " cannot be created";
        return "";
    }
    struct stat st;
    if (lstat(parent.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 077) != 0) {
        err = parent + " is not a private directory of this user (a real directory, owned by it, mode 0700): background compilation is off";
        return "";
    }
    return parent;
}
// ---- the code objects on disk ----------------------------------------------------------------------------------------------
// A compilation takes seconds, during which the interpreter renders; the same set in the NEXT process would wait for the same seconds
// again.  Code objects are therefore also kept in a directory of this user -- RXR_JIT_CACHE_DIR (an absolute path), else
// $XDG_CACHE_HOME/rusterix_amd/jit, else $HOME/.cache/rusterix_amd/jit; RXR_JIT_CACHE=0 switches it off -- under the rules of the job
// directories: created 0700, a real directory owned by this user without group / other write bits (checked with lstat before every
// use), files opened through the directory's descriptor with O_NOFOLLOW, regular files of this user only, written under a temporary
// name and renamed.  A file holds the COMPLETE key material in front of the code object -- generated source, architecture, template
// level, extra flags, a hash of the kernel sources this library embeds and the HIP version it was built with and runs on -- and is
// only used when that material is byte-identical: the 128-bit file name merely finds it.  At most 64 files, the oldest go first.
uint64_t fnv1a64(const void *data, size_t n, uint64_t h) {
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001B3ull;
    }
    return h;
}
const std::string &disk_build_id() {
    static const std::string id = [] {
        uint64_t a = 0xCBF29CE484222325ull, b = 0x84222325CBF29CE4ull;
        for (int i = 0; i < rxr_jit_n_headers; ++i) {
            a = fnv1a64(rxr_jit_header_names[i], strlen(rxr_jit_header_names[i]) + 1, a);
            a = fnv1a64(rxr_jit_header_sources[i], strlen(rxr_jit_header_sources[i]) + 1, a);
            b = fnv1a64(rxr_jit_header_sources[i], strlen(rxr_jit_header_sources[i]) + 1, b * 31u + 7u);
        }
        int rt = 0;
        (void)hipRuntimeGetVersion(&rt);  // (no device needed)
        char buf[128];
        snprintf(buf, sizeof buf, "sources %016llx%016llx hip-built %d.%d.%d hip-runtime %d", (unsigned long long)a, (unsigned long long)b, HIP_VERSION_MAJOR,
                 HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt);
        return std::string(buf);
    }();
    return id;
}
bool private_dir(const std::string &path) {
    struct stat st;
    return lstat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 022) == 0;
}
// "" = no cache (switched off, no usable place)
std::string disk_cache_dir() {
    if (const char *off = getenv("RXR_JIT_CACHE")) {
        if (off[0] == '0') return "";
    }
    std::vector<std::string> chain;  // directories to create if missing, outermost first; the last one is the cache
    if (const char *d = getenv("RXR_JIT_CACHE_DIR")) {
        if (d[0] != '/') return "";
        chain = {d};
    } else {
        std::string base;
        const char *xdg = getenv("XDG_CACHE_HOME"), *home = getenv("HOME");
        if (xdg && xdg[0] == '/') base = xdg;
        else if (home && home[0] == '/') base = std::string(home) + "/.cache";
        else return "";
        chain = {base, base + "/rusterix_amd", base + "/rusterix_amd/jit"};
    }
    for (const std::string &c : chain)
        if (mkdir(c.c_str(), 0700) != 0 && errno != EEXIST) return "";
    return private_dir(chain.back()) ? chain.back() : "";
}
std::string disk_material(const std::string &key) { return "rxr-jit-cache-1\n" + disk_build_id() + "\n" + key; }
std::string disk_name(const std::string &material) {
    char buf[48];
    snprintf(buf, sizeof buf, "%016llx%016llx.rxrco", (unsigned long long)fnv1a64(material.data(), material.size(), 0xCBF29CE484222325ull),
             (unsigned long long)fnv1a64(material.data(), material.size(), 0x9E3779B97F4A7C15ull));
    return buf;
}
constexpr char DISK_MAGIC[8] = {'R', 'X', 'R', 'J', 'I', 'T', '0', '1'};
bool disk_get(const std::string &key, std::vector<char> &obj) {
    const std::string dir = disk_cache_dir();
    if (dir.empty()) return false;
    const std::string material = disk_material(key);
    const int dfd = open(dir.c_str(), O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
    if (dfd < 0) return false;
    const int fd = openat(dfd, disk_name(material).c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    bool ok = false;
    if (fd >= 0) {
        struct stat st;
        const size_t head = sizeof DISK_MAGIC + sizeof(uint64_t) + material.size();
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 022) == 0 && (size_t)st.st_size > head &&
            (size_t)st.st_size < head + (256u << 20)) {
            std::vector<char> all((size_t)st.st_size);
            size_t got = 0;
            while (got < all.size()) {
                const ssize_t n = read(fd, all.data() + got, all.size() - got);
                if (n <= 0) break;
                got += (size_t)n;
            }
            uint64_t mlen = 0;
            if (got == all.size() && memcmp(all.data(), DISK_MAGIC, sizeof DISK_MAGIC) == 0) {
                memcpy(&mlen, all.data() + sizeof DISK_MAGIC, sizeof mlen);
                if (mlen == material.size() && memcmp(all.data() + sizeof DISK_MAGIC + sizeof mlen, material.data(), material.size()) == 0) {
                    obj.assign(all.begin() + head, all.end());
                    ok = !obj.empty();
                    (void)futimens(fd, nullptr);  // (recently used: the pruning below removes the oldest)
                }
            }
        }
        close(fd);
    }
    close(dfd);
    return ok;
}
void disk_put(const std::string &key, const std::vector<char> &obj) {
    const std::string dir = disk_cache_dir();
    if (dir.empty() || obj.empty()) return;
    const std::string material = disk_material(key), name = disk_name(material);
    const int dfd = open(dir.c_str(), O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
    if (dfd < 0) return;
    static std::atomic<unsigned> counter{0};
    char tmp[96];
    snprintf(tmp, sizeof tmp, "%.40s.tmp-%ld-%u", name.c_str(), (long)getpid(), counter.fetch_add(1u));
    const int fd = openat(dfd, tmp, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd >= 0) {
        const uint64_t mlen = material.size();
        auto put = [&](const void *p, size_t n) {
            const char *c = (const char *)p;
            while (n) {
                const ssize_t w = write(fd, c, n);
                if (w <= 0) return false;
                c += w;
                n -= (size_t)w;
            }
            return true;
        };
        const bool ok = put(DISK_MAGIC, sizeof DISK_MAGIC) && put(&mlen, sizeof mlen) && put(material.data(), material.size()) && put(obj.data(), obj.size());
        close(fd);
        if (!ok || renameat(dfd, tmp, dfd, name.c_str()) != 0) (void)unlinkat(dfd, tmp, 0);
    }
    // at most 64 code objects (and no temporary of a writer that died an hour ago)
    if (DIR *d = fdopendir(dup(dfd))) {
        std::vector<std::pair<time_t, std::string>> files;
        const time_t now = time(nullptr);
        while (dirent *e = readdir(d)) {
            const size_t len = strlen(e->d_name);
            struct stat st;
            if (fstatat(dfd, e->d_name, &st, AT_SYMLINK_NOFOLLOW) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid()) continue;
            if (len > 6 && !strcmp(e->d_name + len - 6, ".rxrco")) files.push_back({st.st_mtime, e->d_name});
            else if (strstr(e->d_name, ".rxrco.tmp-") && now - st.st_mtime > 3600) (void)unlinkat(dfd, e->d_name, 0);
        }
        closedir(d);
        if (files.size() > 64u) {
            std::sort(files.begin(), files.end());
            for (size_t i = 0; i + 64u < files.size(); ++i) (void)unlinkat(dfd, files[i].second.c_str(), 0);
        }
    }
    close(dfd);
}
// memory first, then the disk (what is found there moves into memory)
bool cached_object(const std::string &key, std::vector<char> *obj) {
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_code_objects.find(key);
        if (it != g_code_objects.end()) {
            if (obj) *obj = it->second;
            return true;
        }
    }
    std::vector<char> from_disk;
    if (!disk_get(key, from_disk)) return false;
    if (obj) *obj = from_disk;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_code_objects[key].swap(from_disk);
    return true;
}

// this process's PID namespace (two containers that share a /tmp each have a process 1234: a pid says nothing across them)
unsigned long long pid_namespace() {
    struct stat st;
    return stat("/proc/self/ns/pid", &st) == 0 ? (unsigned long long)st.st_ino : 0ull;
}
// Everything below directory `name` of the open directory `at`, then the directory itself.  Walks by file descriptor
// (openat O_DIRECTORY | O_NOFOLLOW, unlinkat): a symbolic link is removed as a link, never followed; a killed compiler leaves
// comgr-<pid>-... directories of temporaries behind, hence the (bounded) recursion.
void remove_tree_at(int at, const char *name, int depth = 0) {
    const int fd = openat(at, name, O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
    if (fd >= 0) {
        if (DIR *d = fdopendir(fd)) {
            while (dirent *e = readdir(d)) {
                if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
                if (unlinkat(fd, e->d_name, 0) != 0 && depth < 4) remove_tree_at(fd, e->d_name, depth + 1);
            }
            closedir(d);  // (closes fd)
        } else {
            close(fd);
        }
    }
    (void)unlinkat(at, name, AT_REMOVEDIR);
}
// `dir` is a job directory this process made with mkdtemp inside job_parent()
void remove_job_dir(const std::string &dir) {
    const size_t slash = dir.rfind('/');
    if (slash == std::string::npos || slash == 0) return;
    const int pfd = open(dir.substr(0, slash).c_str(), O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
    if (pfd < 0) return;
    remove_tree_at(pfd, dir.c_str() + slash + 1);
    close(pfd);
}
// A process that was killed (or left through _exit) could not remove its jobs' directories: each directory names its owner
// (file "owner": pid and PID-namespace inode), and the first job of a process removes the directories -- real directories of this
// user inside the private parent only -- whose owner, in OUR namespace, no longer exists; a directory of another namespace or
// without a readable owner is stale once it is an hour old (a compilation takes seconds).
void sweep_stale_job_dirs(const std::string &parent) {
    const int pfd = open(parent.c_str(), O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
    if (pfd < 0) return;
    DIR *d = fdopendir(dup(pfd));
    if (!d) {
        close(pfd);
        return;
    }
    const unsigned long long my_ns = pid_namespace();
    std::vector<std::string> stale;
    while (dirent *e = readdir(d)) {
        if (strncmp(e->d_name, "job_", 4) != 0) continue;
        struct stat st;
        if (fstatat(pfd, e->d_name, &st, AT_SYMLINK_NOFOLLOW) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid()) continue;
        long pid = 0;
        unsigned long long ns = 0;
        const int jfd = openat(pfd, e->d_name, O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
        if (jfd >= 0) {
            const int ofd = openat(jfd, "owner", O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
            if (ofd >= 0) {
                char buf[64] = {0};
                const ssize_t n = read(ofd, buf, sizeof buf - 1);
                if (n <= 0 || sscanf(buf, "%ld %llu", &pid, &ns) < 1) pid = 0;
                close(ofd);
            }
            close(jfd);
        }
        const bool old = time(nullptr) - st.st_mtime > 3600;
        if (pid > 0 && ns != 0 && ns == my_ns) {
            if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) stale.push_back(e->d_name);
        } else if (old) {
            stale.push_back(e->d_name);
        }
    }
    closedir(d);
    for (const std::string &name : stale) remove_tree_at(pfd, name.c_str());
    close(pfd);
}

void finish_job_locked(std::map<std::string, BgJob>::iterator it, bool kill_it) {
    BgJob &j = it->second;
    if (kill_it && j.pid > 0) {
        (void)kill(j.pid, SIGKILL);
        int st = 0;
        (void)waitpid(j.pid, &st, 0);
    }
    if (!j.dir.empty()) remove_job_dir(j.dir);
    g_bg.erase(it);
}
// library unload (process exit): compilations still in flight are killed and their files removed -- plain system calls, unlike
// joining a compiler thread
struct BgReaper {
    ~BgReaper() {
        std::lock_guard<std::mutex> lk(g_bg_mu);
        while (!g_bg.empty()) finish_job_locked(g_bg.begin(), true);
    }
} g_bg_reaper;
void detach(rxr_ctx *ctx, int slot) {
    if (ctx->jit_wait_key[slot].empty()) return;
    std::lock_guard<std::mutex> lk(g_bg_mu);
    auto it = g_bg.find(ctx->jit_wait_key[slot]);
    if (it != g_bg.end() && --it->second.refs <= 0) finish_job_locked(it, true);  // nobody waits for it any more
    ctx->jit_wait_key[slot].clear();
}
std::string sibling_path(const char *name, std::string *self = nullptr) {
    Dl_info info;
    if (!dladdr((const void *)&rxr_jit_generate, &info) || !info.dli_fname) return "";
    std::string lib = info.dli_fname;
    if (self) *self = lib;
    const size_t slash = lib.rfind('/');
    return (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/" + name;
}
// The child's environment.  rxr_jitc compiles (hiprtc needs no device: the architecture is an argument) and must never touch the
// GPU -- on the pool a second process on the card counts against the box's limit, and under a profiler the parent's environment
// preloads a tool library that initialises the GPU in every process it reaches.  So: ours, minus whatever would load code into the
// child or point a ROCm tool at it (LD_PRELOAD, LD_AUDIT, HSA_TOOLS_*, ROCP* / ROCPROFILER* / ROCTRACER* / ROCTX*, RPD*), with every
// device hidden from both runtimes, and TMPDIR pointing into the job's directory.
static bool child_env_dropped(const char *e) {
    static const char *const prefixes[] = {"TMPDIR=", "LD_PRELOAD=", "LD_AUDIT=", "HSA_TOOLS_", "ROCP", "ROCTRACER", "ROCTX", "RPD",
                                           "HIP_VISIBLE_DEVICES=", "ROCR_VISIBLE_DEVICES=", "CUDA_VISIBLE_DEVICES=", "GPU_DEVICE_ORDINAL="};
    for (const char *p : prefixes)
        if (strncmp(e, p, strlen(p)) == 0) return true;
    return false;
}
static std::vector<std::string> child_environment(const std::string &tmpdir) {
    std::vector<std::string> env;
    for (char **e = environ; e && *e; ++e)
        if (!child_env_dropped(*e)) env.push_back(*e);
    env.push_back("TMPDIR=" + tmpdir);
    env.push_back("HIP_VISIBLE_DEVICES=");   // (an empty list: no device for the HIP runtime ...
    env.push_back("ROCR_VISIBLE_DEVICES=");  //  ... nor for the HSA runtime underneath)
    return env;
}
bool start_child_locked(const std::string &key, const std::string &gen, const std::string &arch, int level, std::string &err) {
    std::string lib;
    const std::string exe = sibling_path("rxr_jitc", &lib);
    if (exe.empty() || access(exe.c_str(), X_OK) != 0) {
        err = "the background compiler rxr_jitc is not next to the library";
        return false;
    }
    const std::string parent = job_parent(err);
    if (parent.empty()) return false;
    static bool swept = false;  // (under g_bg_mu)
    if (!swept) {
        swept = true;
        sweep_stale_job_dirs(parent);
    }
    std::string dir_t = parent + "/job_XXXXXX";
    if (!mkdtemp(&dir_t[0])) {
        err = "no temporary directory";
        return false;
    }
    const std::string dir = dir_t, src_t = dir + "/set.h", out_t = dir + "/set.co";
    if (FILE *f = fopen((dir + "/owner").c_str(), "w")) {
        fprintf(f, "%ld %llu\n", (long)getpid(), pid_namespace());
        fclose(f);
    }
    bool wrote = false;
    if (FILE *f = fopen(src_t.c_str(), "wb")) {
        wrote = fwrite(gen.data(), 1, gen.size(), f) == gen.size();
        fclose(f);
    }
    const std::string lvl = std::to_string(level);
    char *const argv[] = {(char *)exe.c_str(), (char *)lib.c_str(), (char *)src_t.c_str(), (char *)arch.c_str(), (char *)lvl.c_str(), (char *)out_t.c_str(), nullptr};
    std::vector<std::string> env_store = child_environment(dir);
    std::vector<char *> envp;
    for (std::string &e : env_store) envp.push_back((char *)e.c_str());
    envp.push_back(nullptr);
    pid_t pid = 0;
    if (!wrote || posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv, envp.data()) != 0) {
        remove_job_dir(dir);
        err = "the background compiler could not be started";
        return false;
    }
    BgJob j;
    j.dir = dir;
    j.src = src_t;
    j.out = out_t;
    j.pid = pid;
    j.refs = 1;
    g_bg[key] = j;
    return true;
}
// 1: the code object is in the cache now; 0: not yet (still compiling, or waiting for a slot); -1: failed (err)
int poll_background(rxr_ctx *ctx, int slot, int level, std::string &err) {
    const std::string key = cache_key(ctx->jit_arch, level, ctx->jit_source);
    if (cached_object(key, nullptr)) return 1;  // (this process's, or an earlier process's on disk: no child then)
    std::lock_guard<std::mutex> lk(g_bg_mu);
    auto failed = g_bg_failed.find(key);
    if (failed != g_bg_failed.end()) {
        err = failed->second;
        ctx->jit_wait_key[slot].clear();
        return -1;
    }
    auto it = g_bg.find(key);
    if (it == g_bg.end()) {
        // (if this context was attached, the job has ended between two polls: the cache or the failure table answered above, or it
        // was another context's key -- start over)
        ctx->jit_wait_key[slot].clear();
        if (g_bg.size() >= RXR_JIT_MAX_CHILDREN) return 0;  // wait for a slot
        if (!start_child_locked(key, ctx->jit_source, ctx->jit_arch, level, err)) {
            g_bg_failed[key] = err;
            return -1;
        }
        ctx->jit_wait_key[slot] = key;
        return 0;
    }
    if (ctx->jit_wait_key[slot] != key) {  // somebody else's child compiles this very set: share it
        ctx->jit_wait_key[slot] = key;
        ++it->second.refs;
    }
    int st = 0;
    const pid_t r = waitpid(it->second.pid, &st, WNOHANG);
    if (r == 0) return 0;  // still compiling
    bool ok = r > 0 && WIFEXITED(st) && WEXITSTATUS(st) == 0;
    std::vector<char> obj;
    if (ok) {
        FILE *f = fopen(it->second.out.c_str(), "rb");
        ok = f != nullptr;
        if (f) {
            char buf[65536];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) obj.insert(obj.end(), buf, buf + n);
            fclose(f);
        }
        ok = ok && !obj.empty();
    }
    it->second.pid = 0;
    finish_job_locked(it, false);
    ctx->jit_wait_key[slot].clear();
    if (!ok) {
        err = "the background compiler failed";
        g_bg_failed[key] = err;
        return -1;
    }
    std::lock_guard<std::mutex> lk2(g_cache_mu);
    g_code_objects[key].swap(obj);
    return 1;
}
}  // namespace

// tests (no device needed): the clean-up of stale job directories under `parent`, and where this process keeps its own
extern "C" void rxr_debug_jit_sweep(const char *parent) { sweep_stale_job_dirs(parent); }
// the environment a background compiler would be started with, one "NAME=value" per line; returns the length needed
extern "C" int rxr_debug_jit_child_env(const char *tmpdir, char *out, uint32_t capacity) {
    std::string text;
    for (const std::string &e : child_environment(tmpdir ? tmpdir : "")) text += e + "\n";
    if (out && capacity) snprintf(out, capacity, "%s", text.c_str());
    return (int)text.size() + 1;
}
extern "C" int rxr_debug_jit_job_parent(char *out, uint32_t capacity) {
    std::string err;
    const std::string p = job_parent(err);
    snprintf(out, capacity, "%s", p.empty() ? err.c_str() : p.c_str());
    return p.empty() ? RXR_ERR_INVALID : RXR_OK;
}

// rxr_jitc's half: generated source file -> code object file (no device needed)
extern "C" int rxr_debug_jit_compile_file(const char *src_path, const char *arch, int level, const char *out_path) {
    std::string gen, err;
    {
        FILE *f = fopen(src_path, "rb");
        if (!f) return RXR_ERR_INVALID;
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) gen.append(buf, n);
        fclose(f);
    }
    std::vector<char> obj;
    double seconds = 0.0;
    if (!rxr_jit_compile(gen, arch, level, obj, seconds, err)) {
        fprintf(stderr, "rxr_jitc: %s\n", err.c_str());
        return RXR_ERR_HIP;
    }
    // (written under another name first: the parent must never see half a file)
    const std::string tmp = std::string(out_path) + ".part";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return RXR_ERR_INVALID;
    const bool ok = fwrite(obj.data(), 1, obj.size(), f) == obj.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), out_path) != 0) {
        (void)unlink(tmp.c_str());
        return RXR_ERR_INVALID;
    }
    return RXR_OK;
}

void rxr_jit_drop(rxr_ctx *ctx) {
    ctx->jit_palette_miss = false;
    for (int k = 0; k < 3; ++k) {
        if (ctx->jit_module[k]) (void)hipModuleUnload((hipModule_t)ctx->jit_module[k]);
        ctx->jit_module[k] = ctx->jit_fn[k] = ctx->jit_fn_cut[k] = nullptr;
        ctx->jit_failed[k] = false;
        detach(ctx, k);
    }
    ctx->jit_source.clear();
}

// generated header -> code object of the raster kernel at template level `level` (2 / 7 / 8) for `arch` ("gfx950"), through the
// process-wide cache; needs no device
bool rxr_jit_compile(const std::string &gen, const std::string &arch, int level, std::vector<char> &obj, double &seconds, std::string &err) {
    seconds = 0.0;
    const char *flags_env = getenv("RXR_JIT_FLAGS");
    const std::string key = cache_key(arch, level, gen);
    if (cached_object(key, &obj)) return true;
    const Rtc &R = rtc();
    if (!R.ok) {
        err = R.why;
        return false;
    }
    std::vector<const char *> names(rxr_jit_header_names, rxr_jit_header_names + rxr_jit_n_headers), sources(rxr_jit_header_sources, rxr_jit_header_sources + rxr_jit_n_headers);
    names.push_back("rxr_jit_programs.h");
    sources.push_back(gen.c_str());
    hiprtcProgram prog = nullptr;
    const char *main_src = "#define RXR_JIT 1\n#include \"rxr_kernels.hip\"\n";
    if (R.CreateProgram(&prog, main_src, "rxr_jit_main.hip", (int)names.size(), sources.data(), names.data()) != HIPRTC_SUCCESS) {
        err = "hiprtcCreateProgram failed";
        return false;
    }
    const std::string arch_opt = "--offload-arch=" + arch, level_opt = "-DRXR_JIT_LEVEL=" + std::to_string(level);
    std::vector<std::string> extra;
    if (flags_env) {
        std::string cur;
        for (const char *c = flags_env;; ++c) {
            if (*c == ' ' || *c == 0) {
                if (!cur.empty()) extra.push_back(cur);
                cur.clear();
                if (!*c) break;
            } else cur += *c;
        }
    }
    std::vector<const char *> opts = {arch_opt.c_str(), level_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    for (const std::string &x : extra) opts.push_back(x.c_str());
    const auto t0 = std::chrono::steady_clock::now();
    const hiprtcResult r = R.CompileProgram(prog, (int)opts.size(), opts.data());
    seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        std::string log;
        if (R.GetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n) {
            log.resize(n);
            (void)R.GetProgramLog(prog, &log[0]);
        }
        (void)R.DestroyProgram(&prog);
        err = std::string(R.GetErrorString(r)) + ": " + log.substr(0, 2000);
        return false;
    }
    size_t n = 0;
    if (R.GetCodeSize(prog, &n) != HIPRTC_SUCCESS || n == 0) {
        (void)R.DestroyProgram(&prog);
        err = "empty code object";
        return false;
    }
    obj.resize(n);
    (void)R.GetCode(prog, obj.data());
    (void)R.DestroyProgram(&prog);
    disk_put(key, obj);  // (the background child comes through here: the next process finds what it compiled)
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_code_objects[key] = obj;
    return true;
}

// the set just flattened: generates its programs and keeps the source; the kernels are compiled per template level when the first
// frame that needs one is launched (rxr_jit_launch).  A set the generator does not cover leaves the interpreter in charge.
int rxr_jit_build(rxr_ctx *ctx, const std::vector<uint32_t> &code, const std::vector<DevProgram> &progs) {
    rxr_jit_drop(ctx);
    ctx->jit_info.clear();
    std::string gen, why;
    if (!rxr_jit_generate(code, progs, gen, why)) {
        ctx->jit_info = "not compiled: " + why;
        return RXR_OK;
    }
    hipDeviceProp_t props;
    ctx->jit_arch = "gfx950";
    if (hipGetDeviceProperties(&props, ctx->device) == hipSuccess && props.gcnArchName[0]) {
        ctx->jit_arch = props.gcnArchName;  // "gfx950:sramecc+:xnack-"
        ctx->jit_arch = ctx->jit_arch.substr(0, ctx->jit_arch.find(':'));
    }
    ctx->jit_source.swap(gen);
    char msg[160];
    snprintf(msg, sizeof msg, "generated: %zu program(s), %zu words of jump code", progs.size(), code.size());
    ctx->jit_info = msg;
    return RXR_OK;
}

namespace {
// slot 0 / 1 / 2 = template level 2 / 7 / 8
bool ensure_level(rxr_ctx *ctx, int slot) {
    if (ctx->jit_fn[slot]) return true;
    if (ctx->jit_failed[slot]) return false;
    static const int levels[3] = {2, 7, 8};
    std::vector<char> obj;
    double seconds = 0.0;
    std::string err;
    if (ctx->jit_async) {
        const int st = poll_background(ctx, slot, levels[slot], err);
        if (st < 0) {
            ctx->jit_failed[slot] = true;
            ctx->jit_info = "not compiled: " + err;
            return false;
        }
        if (st == 0) {  // (this frame and the next ones: the interpreter)
            ctx->jit_info = (ctx->jit_wait_key[slot].empty() ? "waiting for a compiler slot: template level " : "compiling in the background: template level ") + std::to_string(levels[slot]);
            return false;
        }
        if (!rxr_jit_compile(ctx->jit_source, ctx->jit_arch, levels[slot], obj, seconds, err)) {  // (answers from the cache)
            ctx->jit_failed[slot] = true;
            ctx->jit_info = "not compiled: " + err;
            return false;
        }
    } else if (!rxr_jit_compile(ctx->jit_source, ctx->jit_arch, levels[slot], obj, seconds, err)) {
        ctx->jit_failed[slot] = true;
        ctx->jit_info = "not compiled: " + err;
        return false;
    }
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipModuleLoadData(&mod, obj.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, "k_raster_jit");
    if (e != hipSuccess) {
        if (mod) (void)hipModuleUnload(mod);
        ctx->jit_failed[slot] = true;
        ctx->jit_info = std::string("not loaded: ") + hipGetErrorString(e);
        return false;
    }
    ctx->jit_module[slot] = mod;
    ctx->jit_fn[slot] = fn;
    hipFunction_t fn_cut = nullptr;
    ctx->jit_fn_cut[slot] = nullptr;
    if (levels[slot] != 2) {  // (template level 2 has none)
        if (hipModuleGetFunction(&fn_cut, mod, "k_raster_jit_cut") == hipSuccess) ctx->jit_fn_cut[slot] = (void *)fn_cut;
        else (void)hipGetLastError();   // (a code object of an older build in the disk cache: the plain kernel serves; the error must not stay behind)
    }
    char msg[200];
    snprintf(msg, sizeof msg, "compiled: template level %d, %zu bytes of code object, %.2f s%s", levels[slot], obj.size(), seconds, seconds == 0.0 ? " (cached)" : "");
    ctx->jit_info = msg;
    return true;
}
}  // namespace

// the raster launch of a frame whose programs are compiled; false: no compiled kernel (the caller launches the interpreter kernels)
bool rxr_jit_launch(rxr_ctx *ctx, const RasterParams *P, hipStream_t s) {
    if (ctx->jit_source.empty() || P->kernel_level < 2u || ctx->jit_palette_miss) return false;
    // RasterParams.kernel_level 4 / 5 (k_raster_vm_sv / _v, template levels 6 / 7): no program of the opaque pass decides
    // visibility (rxr_upload_frame); 2 / 3: one may.  When the frame needs none of the chunk paths either: template level 8.
    const int slot = P->kernel_level >= 4u ? (ctx->frame_needs_chunk_paths ? 1 : 2) : 0;
    if (!ensure_level(ctx, slot)) return false;
    if (P->tiles_x * P->tiles_y == 0) return true;
    RasterParams params = *P;
    size_t size = sizeof(params);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &params, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    // (frames with cut-out or profiled batches: rounds in row mode around them -- binned frames only, the others never reach that code)
    const bool cut = P->split_rounds && ctx->jit_fn_cut[slot] && P->fused_small == 0u && (P->flags & RXR_FLAG_D3_ACTIVE);
    const hipFunction_t kernel = (hipFunction_t)(cut ? ctx->jit_fn_cut[slot] : ctx->jit_fn[slot]);
    hipEvent_t e0, e1;
    if (rxr_launch_times && rxr_launch_pair(&e0, &e1))  // (kernel timing, rxr_launch.h; this form takes the grid in THREADS)
        return hipExtModuleLaunchKernel(kernel, P->tiles_x * RXR_TILE_THREADS, P->tiles_y, 1, RXR_TILE_THREADS, 1, 1, 0, s, nullptr, config,
                                        e0, e1, 0) == hipSuccess;
    return hipModuleLaunchKernel(kernel, P->tiles_x, P->tiles_y, 1, RXR_TILE_THREADS, 1, 1, 0, s, nullptr, config) == hipSuccess;
}
