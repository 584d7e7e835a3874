#!/usr/bin/env python3
"""Second stage of tools/sanitize_cpu.sh: the device-free half of rxr_set_shaders (rxr_check_shaders: structural validation,
definite-assignment analysis, flattening into jump code -- host code of librxr_hip.so, built here with AddressSanitizer and
UBSan on the HOST side only) fed with (a) the random well-formed programs of the GPU test suite and (b) MALFORMED word streams:
every serialised function of (a) with words flipped, truncated, extended or its block lengths corrupted.  The boundary must answer
every one of them with a status code; the sanitizers watch it do so."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rusterix_amd import binding as B  # noqa: E402
from tests import test_gpu_shaders as S  # noqa: E402
from tests import test_shader_validation as V  # noqa: E402

assert "librxr_hip_hostasan.so" in os.environ.get("RXR_DEVICE_SO", ""), "run through tools/sanitize_cpu.sh"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
counts = {}


class Raw:
    """a program whose functions are already word lists (possibly malformed)"""

    def __init__(self, functions, n_globals, shade_index, shade_locals):
        self.functions, self.globals, self.shade_index, self.shade_locals = functions, n_globals, shade_index, shade_locals


generated = 0


def jit_generate(p):
    import ctypes as C

    import rusterix_amd

    lib = rusterix_amd.load_rxr()
    lib.rxr_debug_jit_generate.argtypes = [C.POINTER(V.RxrShaderSet), C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
    keep = []
    fns = (V.RxrFunction * max(len(p.functions), 1))()
    for k, f in enumerate(p.functions):
        arr = np.asarray(f if len(f) else [0], np.uint32)
        keep.append(arr)
        fns[k] = V.RxrFunction(arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(f))
    progs = (V.RxrProgram * 1)(V.RxrProgram(p.globals, p.shade_index, p.shade_locals, fns, len(p.functions)))
    s = V.RxrShaderSet(progs, 1, None, 0, None, 0, None, None, 0)
    src, msg = C.create_string_buffer(1 << 18), C.create_string_buffer(512)
    rc = lib.rxr_debug_jit_generate(C.byref(s), 0, src, len(src), msg, len(msg))
    assert rc in (0, B.RXR_ERR_UNSUPPORTED), (rc, msg.value)
    return 1 if rc == 0 and b"rxr_jit_shade" in src.value else 0


def tally(rc):
    counts[rc] = counts.get(rc, 0) + 1


for seed in range(n):
    rng = np.random.default_rng([0x52585231, 31337, seed])
    prog = S.ProgramGen(rng, n_locals=int(rng.integers(1, 6)), n_functions=int(rng.integers(0, 3))).program()
    rc, msg, words = V.check(prog)
    tally(rc)
    if rc == 0:  # the run-time compiler's code generator (rxr_jit.hip) on the same set, without the compile step
        generated += jit_generate(prog)
    assert "librxr_hip_hostasan.so" in open("/proc/self/maps").read()
    fns = [list(f) for f in prog.functions]
    for _ in range(12):
        mutated = [list(f) for f in fns]
        f = mutated[int(rng.integers(0, len(mutated)))]
        kind = int(rng.integers(0, 6))
        if kind == 0 and f:      # flip a word to anything
            f[int(rng.integers(0, len(f)))] = int(rng.integers(0, 2**32))
        elif kind == 1 and f:    # a small opcode / length where something else was
            f[int(rng.integers(0, len(f)))] = int(rng.integers(0, 130))
        elif kind == 2 and f:    # truncate
            del f[int(rng.integers(0, len(f))):]
        elif kind == 3:          # garbage tail
            f.extend(int(x) for x in rng.integers(0, 2**32, int(rng.integers(1, 9))))
        elif kind == 4 and f:    # a huge block length
            f[int(rng.integers(0, len(f)))] = int(rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x10000, len(f), len(f) + 1]))
        else:                    # header fields out of range
            pass
        hdr = dict(n_globals=prog.globals, shade_index=prog.shade_index, shade_locals=prog.shade_locals)
        if kind == 5:
            which = int(rng.integers(0, 3))
            if which == 0:
                hdr["n_globals"] = int(rng.choice([17, 1000, 0xFFFFFFFF]))
            elif which == 1:
                hdr["shade_index"] = int(rng.choice([-1, len(mutated), 1 << 20]))
            else:
                hdr["shade_locals"] = int(rng.choice([49, 1 << 16, 0xFFFFFFFF]))
        rc, msg, words = V.check(Raw(mutated, hdr["n_globals"], hdr["shade_index"], hdr["shade_locals"]))
        tally(rc)
names = {0: "OK", B.RXR_ERR_INVALID: "INVALID", B.RXR_ERR_UNSUPPORTED: "UNSUPPORTED"}
print(f"shader boundary under sanitizers: clean; run-time compiler generated code for {generated} sets;", ", ".join(f"{names.get(k, k)}: {v}" for k, v in sorted(counts.items(), reverse=True)))
