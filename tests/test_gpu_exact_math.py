"""The raster kernel's short division / sqrt / pow sequences (rusterix_amd/csrc/rxr_exact_math.h) against
hipcc's expansions of the plain operators, bit for bit, on the device.

The sequences are the compiler's own with the instructions removed that are no-ops inside an operand
window, guarded by a wave-uniform window test; this runs both over seeded operand tuples (whole waves
inside the window, at its ends, straddling it, and raw bit patterns with every special value) and
requires zero differing results (NaN == NaN)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

KINDS = ["div2", "div3", "div3_self", "normalize3", "sqrt", "pow", "div1", "static", "normalize3_zeros", "sqrt_sweep", "sat_u32_cvt", "wave_max_dpp", "wave_scan_dpp"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_short_sequences_are_bit_identical(product, seed):
    lib = C.CDLL(__import__("rusterix_amd").lib_paths()["rxr"])
    lib.rxr_selftest_math.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.rxr_selftest_math.restype = C.c_int
    product.lib.rxh_context.restype = C.c_void_p
    ctx = product.lib.rxh_context()
    assert ctx
    out = (C.c_uint64 * len(KINDS))()
    n = 1 << 36  # tuples per kind and seed (about half a second on an MI355X)
    rc = lib.rxr_selftest_math(ctx, n, seed, out)
    assert rc == 0
    assert dict(zip(KINDS, list(out))) == {k: 0 for k in KINDS}
