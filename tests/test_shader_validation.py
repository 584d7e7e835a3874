"""rxr_check_shaders: the device-free half of rxr_set_shaders -- structural validation, the purity (definite-assignment)
analysis and the refusals documented in include/rxr.h.  Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

import rusterix_amd
from rusterix_amd import binding as B
from rusterix_amd.binding import Program, assemble


class RxrFunction(C.Structure):
    _fields_ = [("words", C.POINTER(C.c_uint32)), ("n_words", C.c_uint32)]


class RxrProgram(C.Structure):
    _fields_ = [("n_globals", C.c_uint32), ("shade_index", C.c_int32), ("shade_locals", C.c_uint32),
                ("functions", C.POINTER(RxrFunction)), ("n_functions", C.c_uint32)]


class RxrShaderSet(C.Structure):
    _fields_ = [("programs", C.POINTER(RxrProgram)), ("n_programs", C.c_uint32), ("patterns", C.c_void_p), ("n_patterns", C.c_uint32),
                ("normal_patterns", C.c_void_p), ("n_normal_patterns", C.c_uint32), ("palette_rgb", C.c_void_p),
                ("palette_present", C.c_void_p), ("n_palette", C.c_uint32)]


def check(*programs):
    lib = rusterix_amd.load_rxr()
    lib.rxr_check_shaders.argtypes = [C.POINTER(RxrShaderSet), C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32]
    keep = []
    progs = (RxrProgram * len(programs))()
    for i, p in enumerate(programs):
        fns = (RxrFunction * max(len(p.functions), 1))()
        for k, f in enumerate(p.functions):
            arr = np.asarray(f if len(f) else [0], np.uint32)
            keep.append(arr)
            fns[k] = RxrFunction(arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(f))
        keep.append(fns)
        progs[i] = RxrProgram(p.globals, p.shade_index, p.shade_locals, fns, len(p.functions))
    s = RxrShaderSet(progs, len(programs), None, 0, None, 0, None, None, 0)
    msg = C.create_string_buffer(512)
    n = C.c_uint32()
    rc = lib.rxr_check_shaders(C.byref(s), C.byref(n), msg, 512)
    return rc, msg.value.decode(), n.value


def P(ops, *functions, **kw):
    return Program([ops] + list(functions), **kw)


def test_accepts_pure_programs_and_reports_the_code_size():
    rc, msg, n = check(P(["UV", ("Push", 4.0), "Mul", "Color", "Add", "SetColor"]))
    assert rc == 0 and msg == ""
    # UV | fused (Push 4; Mul) = 4 words | Color | Add | SetColor | ENDFN + 4 words of padding
    assert n == 1 + 4 + 1 + 1 + 1 + 1 + 4


def test_constant_and_binary_operation_are_fused_only_when_adjacent():
    _, _, fused = check(P([("Push", 1.0), ("Push", 2.0), "Add", "SetColor"]))         # Push(4) + BINC(4) + SetColor + ENDFN + pad
    _, _, plain = check(P([("Push", 1.0), ("Push", 2.0), "Dup", "Clear", "Add", "SetColor"]))  # nothing to fuse
    assert fused == 4 + 4 + 1 + 1 + 4 and plain == 4 + 4 + 1 + 1 + 1 + 1 + 1 + 4


@pytest.mark.parametrize("prog, expect", [
    (P([("Push", 4.0), ("Push", 4.0), "Alloc"]), "Alloc"),
    (P([("For", [], [("Push", 0.0)], [], ["Return"])]), "Return inside For"),
    (P([("LoadGlobal", 0), "SetColor"], globals=1), "read before"),
    (P([("LoadLocal", 0), "SetColor"], shade_locals=1), "read before"),
    (P([("Push", 1.0), ("If", [("Push", 2.0), ("StoreLocal", 0)], None), ("LoadLocal", 0), "SetColor"], shade_locals=1), "read before"),
    (P(["UV", "SetColor", ("Push", 1.0, 2.0, 3.0), "SetUV"]), "reads uv"),
    (P([("FunctionCall", 0, 0, 1), "SetColor"], [("LoadGlobal", 0)], globals=1), "read before"),   # a callee reads an unwritten global
    (P([("GetComponents", list(range(13)))]), "swizzle"),
    (P([], globals=17), "globals"),
    (P([], shade_locals=49), "locals"),
])
def test_refusals(prog, expect):
    rc, msg, _ = check(prog)
    assert rc == B.RXR_ERR_UNSUPPORTED, msg
    assert expect in msg


@pytest.mark.parametrize("prog", [
    # both branches store the local
    P([("Push", 1.0), ("If", [("Push", 2.0), ("StoreLocal", 0)], [("Push", 3.0), ("StoreLocal", 0)]), ("LoadLocal", 0), "SetColor"], shade_locals=1),
    # a For's init and first condition always run
    P([("For", [("Push", 0.0), ("StoreLocal", 0)], [("LoadLocal", 0), ("Push", 3.0), "Lt"], [("LoadLocal", 0), ("Push", 1.0), "Add", ("StoreLocal", 0)], []),
       ("LoadLocal", 0), "SetColor"], shade_locals=1),
    # a global written by shade before the callee that reads it runs
    P(["UV", ("StoreGlobal", 0), ("FunctionCall", 0, 0, 1), "SetColor"], [("LoadGlobal", 0)], globals=1),
    # callees get fresh zeroed locals: reading one is fine
    P([("FunctionCall", 0, 2, 1), "SetColor"], [("LoadLocal", 1)]),
    # a field may be read after the same invocation wrote it
    P([("Push", 0.1, 0.2, 0.3), "SetRoughness", "Roughness", "SetColor"]),
    # a program without a shade function is never run: anything goes
    Program([[("Push", 0.5), "SetEmissive"]], shade_index=None),
    # SetEmissive is a property of the FRAME (which other batches are on screen, rxr_upload_frame), not of the program
    P([("Push", 0.5), "SetEmissive"]),
    P([("Push", 0.5), "SetEmissive", "Emissive", "SetColor"]),
])
def test_accepts(prog):
    rc, msg, _ = check(prog)
    assert rc == 0, msg


def test_a_written_field_taints_reads_in_other_programs_of_the_set():
    writer = P([("Push", 0.1, 0.2, 0.3), "SetBump"])
    reader = P(["Bump", "SetColor"])
    assert check(reader)[0] == 0                       # nobody writes bump: it stays at Execution::new's zero
    rc, msg, _ = check(writer, reader)
    assert rc == B.RXR_ERR_UNSUPPORTED and "bump" in msg


def test_emissive_read_before_written_is_tainted_by_any_writer():
    reader = P(["Emissive", "SetColor"])
    assert check(reader)[0] == 0                       # nobody writes emissive: it stays at Execution::new's zero
    rc, msg, _ = check(P([("Push", 0.5), "SetEmissive"]), reader)
    assert rc == B.RXR_ERR_UNSUPPORTED and "emissive" in msg
    rc, msg, _ = check(P(["Emissive", "SetColor", ("Push", 0.5), "SetEmissive"]))   # its own later write taints the next fragment's read
    assert rc == B.RXR_ERR_UNSUPPORTED and "emissive" in msg


@pytest.mark.parametrize("words", [[9999], [B.NODE_OPCODE["Push"], 1, 2], [B.NODE_OPCODE["If"], 5, 0, 0, 1], [B.NODE_OPCODE["LoadLocal"]]])
def test_malformed_streams_are_invalid(words):
    p = Program([[]])
    p.functions = [words]
    rc, msg, _ = check(p)
    assert rc == B.RXR_ERR_INVALID and msg


def test_shade_index_out_of_range_is_invalid():
    p = Program([[]], shade_index=3)
    assert check(p)[0] == B.RXR_ERR_INVALID


def test_call_of_a_missing_function_is_a_run_time_fault_not_a_refusal():
    # program.user_functions[9] panics only when the call is reached
    assert check(P([("Push", 0.0), ("If", [("FunctionCall", 0, 0, 9)], None)]))[0] == 0
