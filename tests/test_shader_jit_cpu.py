"""The device-free half of the run-time compiler of program sets (rusterix_amd/csrc/rxr_jit.hip): code generation from the jump
code and the hiprtc compilation for gfx950 need no GPU (`rxr_debug_jit_generate`); loading and running the kernels is
tests/test_gpu_shader_jit.py."""
import ctypes as C
import re

import numpy as np

import rusterix_amd
from rusterix_amd import binding as B
from rusterix_amd import scenes
from rusterix_amd.binding import Program
from tests import test_shader_validation as V


def generate(programs, compile_):
    lib = rusterix_amd.load_rxr()
    lib.rxr_debug_jit_generate.argtypes = [C.POINTER(V.RxrShaderSet), C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
    keep = []
    progs = (V.RxrProgram * len(programs))()
    for i, p in enumerate(programs):
        fns = (V.RxrFunction * max(len(p.functions), 1))()
        for k, f in enumerate(p.functions):
            arr = np.asarray(f if len(f) else [0], np.uint32)
            keep.append(arr)
            fns[k] = V.RxrFunction(arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(f))
        keep.append(fns)
        progs[i] = V.RxrProgram(p.globals, p.shade_index, p.shade_locals, fns, len(p.functions))
    s = V.RxrShaderSet(progs, len(programs), None, 0, None, 0, None, None, 0)
    src, msg = C.create_string_buffer(1 << 20), C.create_string_buffer(4096)
    rc = lib.rxr_debug_jit_generate(C.byref(s), compile_, src, len(src), msg, len(msg))
    return rc, src.value.decode(), msg.value.decode()


def test_the_configuration_c5_program_becomes_straight_line_code_and_compiles():
    rc, src, msg = generate([scenes.box_grid_shader()], 1)
    assert rc == 0 and msg.startswith("compiled in"), msg
    body = src[src.index("rxr_jit_prog_0"):]
    # stack slots are variables, the fused "Push c; op" pairs keep their constants bit for bit, the If is two gotos
    assert "v3 s0" in body and "jit_binc<2u>(s0, mk(__uint_as_float(0x40800000u)" in body and len(re.findall(r"goto L\d", body)) == 2
    assert "io.color = s0;" in body and "rxr_jit_shade(const RasterParams &P, uint32_t pi" in src


def test_loops_locals_globals_and_faults_are_generated():
    prog = Program([[("Push", 0.0), ("StoreLocal", 0),
                     ("For", [("Push", 0.0), ("StoreLocal", 1)], [("LoadLocal", 1), ("Push", 3.0), "Lt"], [("LoadLocal", 1), ("Push", 1.0), "Add", ("StoreLocal", 1)],
                      [("LoadLocal", 0), ("Push", 0.1), "Add", ("StoreLocal", 0)]),
                     ("LoadLocal", 0), ("StoreGlobal", 0), ("LoadGlobal", 0), ("Push", 0.0), ("Push", 1.0), "Clamp", "SetColor"]],
                   shade_locals=2, globals=1)
    rc, src, msg = generate([prog, Program([[]])], 0)
    assert rc == 0, msg
    assert "v3 l0" in src and "v3 l1" in src and "v3 g0" in src
    assert re.search(r"if \(\+\+steps > 1048576u\)", src), "the backward jump of a For loop counts steps like the interpreter"
    assert "jit_clamp_ok" in src and f"fault = {8}u" in src        # VMF_CLAMP_BOUNDS
    assert "rxr_jit_prog_1" in src


def test_calls_become_functions_recursion_unrolls_by_call_depth_and_palette_lookups_carry_a_way_back():
    helper = [("LoadLocal", 0), ("Push", 2.0), "Mul", "Return"]
    rc, src, msg = generate([Program([["UV", ("FunctionCall", 1, 1, 1), "SetColor"], helper])], 0)
    assert rc == 0, msg
    assert re.search(r"rxvm::v3 rxr_jit_fn_0_\d+_1_1\(.*rxvm::v3 a0\)", src) and "v3 l0 = a0;" in src and "if (fault) goto Lend;" in src
    fact = [("LoadLocal", 0), ("Push", 1.0), "Le", ("If", [("Push", 1.0), "Return"], None), ("LoadLocal", 0), ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1), "Mul", "Return"]
    # recursion (round 3): one copy of the function per call depth the interpreter's frame stack allows (8), a call in the last copy is
    # the interpreter's VMF_CALL_DEPTH (5), the locals of a chain are checked at run time (`lbase`) like the interpreter's
    rc, src, msg = generate([Program([["UV", ("FunctionCall", 1, 1, 1), "SetColor"], fact])], 0)
    assert rc == 0, msg
    copies = sorted(set(int(m) for m in re.findall(r"rxvm::v3 rxr_jit_fn_0_\d+_1_1_L(\d+)\(", src)))
    assert copies == list(range(1, 9)), copies
    last = src[src.index("_1_1_L8("):]
    last = last[:last.index("\n}\n")]
    assert "fault = 5u" in last and "_L9" not in src
    assert "if (lbase + 2u > 48u)" in src and "const uint32_t lbase" in src and "__forceinline__ rxvm::v3 rxr_jit_fn_0_" in src
    # two recursive call sites per body: 2^8 chains -- the copies become real functions instead of being inlined into each other
    fib = [("LoadLocal", 0), ("Push", 2.0), "Lt", ("If", [("LoadLocal", 0), "Return"], None),
           ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1), ("LoadLocal", 0), ("Push", 2.0), "Sub", ("FunctionCall", 1, 1, 1), "Add", "Return"]
    rc, src, msg = generate([Program([["UV", ("FunctionCall", 1, 1, 1), "SetColor"], fib])], 0)
    assert rc == 0, msg
    assert "__noinline__ rxvm::v3 rxr_jit_fn_0_" in src
    # PaletteIndex (round 3): compiled as the push, with the reference's other case -- a missing or empty slot pushes nothing --
    # raising VMF_JIT_PALETTE_MISS (12), on which rxr_synchronize hands the set back to the interpreter
    rc, src, msg = generate([Program([["UV", "PaletteIndex", "SetColor"]])], 0)
    assert rc == 0, msg
    assert "P.n_palette" in src and "fault = 12u" in src and "P.palette[4u * id" in src


def test_the_background_compiler_is_a_process_that_turns_a_source_file_into_a_code_object(tmp_path):
    """RXR_SHADER_JIT=async hands the generated source to rxr_jitc (a child process: it can be killed, hiprtc in a thread cannot)"""
    import os
    import subprocess

    rc, src, msg = generate([scenes.box_grid_shader()], 0)
    assert rc == 0, msg
    lib = rusterix_amd.lib_paths()["rxr"]
    exe = os.path.join(os.path.dirname(lib), "rxr_jitc")
    assert os.access(exe, os.X_OK), "run __graft_entry__.build()"
    source, out = tmp_path / "set.h", tmp_path / "set.co"
    source.write_text(src)
    pr = subprocess.run([exe, lib, str(source), "gfx950", "8", str(out)], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr
    blob = out.read_bytes()
    assert blob[:4] == b"\x7fELF" and len(blob) > 10000 and not (tmp_path / "set.co.part").exists()
    # a source that does not compile: a non-zero exit status and no output file
    source.write_text(src + "\nthis is not C++\n")
    bad = tmp_path / "bad.co"
    pr = subprocess.run([exe, lib, str(source), "gfx950", "8", str(bad)], capture_output=True, text=True, timeout=300)
    assert pr.returncode != 0 and not bad.exists()


def test_code_objects_outlive_the_process_that_compiled_them(tmp_path):
    """Compiled sets are kept on disk (rxr_jit.hip: RXR_JIT_CACHE_DIR / $XDG_CACHE_HOME / $HOME/.cache, RXR_JIT_CACHE=0 = off): the same set
    in the NEXT process is answered from the file -- the identical code object, in a fraction of the compile time -- and a file that was
    tampered with, a cache directory that others may write to, or a different set are not."""
    import os
    import subprocess
    import time

    rc, src, msg = generate([Program([["Color", ("Push", 0.5), "Mul", "SetColor"]])], 0)
    assert rc == 0, msg
    lib = rusterix_amd.lib_paths()["rxr"]
    exe = os.path.join(os.path.dirname(lib), "rxr_jitc")
    cache = tmp_path / "cache"
    source = tmp_path / "set.h"
    source.write_text(src)
    env = dict(os.environ, RXR_JIT_CACHE="1", RXR_JIT_CACHE_DIR=str(cache))

    def run(out, **more):
        t0 = time.time()
        pr = subprocess.run([exe, lib, str(source), "gfx950", "8", str(out)], capture_output=True, text=True, timeout=300, env=dict(env, **more))
        assert pr.returncode == 0, pr.stderr
        return time.time() - t0, out.read_bytes()

    t_first, first = run(tmp_path / "a.co")
    files = [f for f in os.listdir(cache) if f.endswith(".rxrco")]
    assert len(files) == 1 and (os.stat(cache).st_mode & 0o777) == 0o700 and (os.stat(cache / files[0]).st_mode & 0o777) == 0o600
    assert not [f for f in os.listdir(cache) if ".tmp-" in f]
    t_again, again = run(tmp_path / "b.co")
    assert again == first and t_again < max(1.0, t_first / 3), (t_first, t_again)       # (from the file: no compilation)
    # a different set: its own entry
    source.write_text(src.replace("0.5", "0.25") if "0.5" in src else src + "\n// other\n")
    _, other = run(tmp_path / "c.co")
    assert len([f for f in os.listdir(cache) if f.endswith(".rxrco")]) == 2
    source.write_text(src)
    # a tampered file (one byte of the key material in front of the code object changed): not used, compiled again and replaced
    path = cache / files[0]
    blob = bytearray(path.read_bytes())
    blob[40] ^= 1
    path.write_bytes(bytes(blob))
    t_fixed, fixed = run(tmp_path / "d.co")
    assert fixed == first and t_fixed > t_again * 2, (t_again, t_fixed)
    assert path.read_bytes()[:8] == b"RXRJIT01" and path.read_bytes() != bytes(blob)
    # a directory that group / others may write to is not trusted: nothing is read from it, nothing written to it
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    t_loose, out_loose = run(tmp_path / "e.co", RXR_JIT_CACHE_DIR=str(loose))
    assert out_loose == first and os.listdir(loose) == [] and t_loose > t_again * 2
    # switched off: the directory stays as it is
    before = sorted(os.listdir(cache))
    os.remove(path)
    run(tmp_path / "f.co", RXR_JIT_CACHE="0")
    assert sorted(os.listdir(cache)) == sorted(set(before) - {files[0]})



def test_the_background_compiler_starts_without_a_gpu_and_without_the_parents_tool_libraries(tmp_path, monkeypatch):
    """The compile-only child must be GPU-free by construction (round-3 verdict): under a profiler the parent's environment preloads
    a tool library that initialises the GPU in every process it reaches.  rxr_jitc gets our environment minus LD_PRELOAD / LD_AUDIT /
    HSA_TOOLS_* / ROCP* / ROCPROFILER* / ROCTRACER*, with every device hidden from both runtimes and TMPDIR inside the job's directory --
    and still compiles (hiprtc needs no device: the architecture is an argument)."""
    import os
    import subprocess

    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("LD_AUDIT", "/nonexistent/audit.so")
    monkeypatch.setenv("HSA_TOOLS_LIB", "/opt/rocm/lib/librocprofiler-sdk.so")
    monkeypatch.setenv("HSA_TOOLS_REPORT_LOAD_FAILURE", "1")
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    monkeypatch.setenv("ROCPROF_COUNTERS", "SQ_WAVES")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/x.so")
    monkeypatch.setenv("ROCTRACER_DOMAIN", "hip")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1")
    monkeypatch.setenv("TMPDIR", "/somewhere/else")
    monkeypatch.setenv("RXR_KEEP_ME", "yes")
    lib = rusterix_amd.load_rxr()
    lib.rxr_debug_jit_child_env.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    buf = C.create_string_buffer(1 << 20)
    assert 0 < lib.rxr_debug_jit_child_env(str(tmp_path).encode(), buf, len(buf)) <= len(buf)
    env = dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)
    for name in env:
        assert not name.startswith(("LD_PRELOAD", "LD_AUDIT", "HSA_TOOLS_", "ROCP", "ROCTRACER", "ROCTX")), name
    assert env["HIP_VISIBLE_DEVICES"] == "" and env["ROCR_VISIBLE_DEVICES"] == ""
    assert env["TMPDIR"] == str(tmp_path) and env["RXR_KEEP_ME"] == "yes" and env["PATH"] == os.environ["PATH"]
    # the child really works in that environment
    rc, src, msg = generate([Program([["Color", "SetColor"]])], 0)
    assert rc == 0, msg
    path = rusterix_amd.lib_paths()["rxr"]
    exe = os.path.join(os.path.dirname(path), "rxr_jitc")
    source, out = tmp_path / "set.h", tmp_path / "set.co"
    source.write_text(src)
    pr = subprocess.run([exe, path, str(source), "gfx950", "8", str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert pr.returncode == 0, pr.stderr
    assert out.read_bytes()[:4] == b"\x7fELF"
