#!/usr/bin/env python3
"""One-off wide sweep of the seeded scene generators of tests/ (GPU against the oracle), far more seeds than the test suite
runs.  usage: python tools/fuzz_sweep.py [first_seed] [n_seeds] [--device-projection]
--device-projection: the product projects on the device (row N1: Batch3D::clip_and_project and Batch2D::project as kernels); the
oracle always projects on the host"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusterix_amd  # noqa: E402
from rusterix_amd import binding as B  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_fuzz as F  # noqa: E402
from tests import test_gpu_rows as R  # noqa: E402
from tests import test_gpu_shaders as S  # noqa: E402
from tests import test_gpu_shader_jit as J  # noqa: E402

prod, orc = rusterix_amd.load(), load_oracle()
DEVPROJ = "--device-projection" in sys.argv
argv = [a for a in sys.argv if not a.startswith("--")]
first = int(argv[1]) if len(argv) > 1 else 1000
n = int(argv[2]) if len(argv) > 2 else 100
if DEVPROJ:
    import ctypes as C

    prod.lib.rxh_set_device_projection.argtypes = [C.c_int]
    prod.lib.rxh_set_device_projection(1)
bad = []
refused = []


def check(tag, got, ref, tol, max_bad):
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    nb = int((diff > tol).sum())
    if nb > max_bad:
        bad.append((tag, nb, int(diff.max()), np.argwhere(diff > tol)[:2].tolist()))


def compare(tag, build, tol, max_bad):
    """GPU against the oracle; a frame the device refuses (RXR_ERR_UNSUPPORTED, e.g. a fourth nested opacity batch) is counted, not
    a parity failure: the shim would render it on the CPU"""
    try:
        got = scenes.render(build(prod))
    except B.RasterizeError as e:
        if e.code != B.RXR_ERR_UNSUPPORTED:
            raise
        refused.append((tag, str(e)[40:110]))
        return
    check(tag, got, scenes.render(build(orc)), tol, max_bad)


for s in range(first, first + n):
    w, h = 160 + 16 * (s % 3), 100 + 7 * (s % 4)
    compare(("soup", s), lambda api: F.build(api, s, w, h), 1, 3)
    compare(("chunks", s), lambda api: F.build_chunks(api, s, 168, 104), 1, 3)
    compare(("chunks-dense", s), lambda api: F.build_chunks(api, s, 168, 104, dense=40), 1, 3)
    for variant in ("plain", "ties", "cutout", "mixed", "opacity"):
        ww, hh = 203 + 16 * (s % 3), 131 + 9 * (s % 3)
        compare((variant, s), lambda api: R.build(api, s, ww, hh, variant), 0, 0)
    # random Rusteria programs: as a 2D rectangle's shader (exact) and as a lit cube's material (+-1, a handful of pixels)
    try:
        rng = np.random.default_rng([0x52585231, 4242, s])
        prog = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3))).program()
        check(("program-2d", s), scenes.render(S.rect_scene(prod, prog, time=0.5)), scenes.render(S.rect_scene(orc, prog, time=0.5)), 0, 0)
        rng = np.random.default_rng([0x52585231, 777, s])
        prog = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3))).program()
        check(("program-cube", s), scenes.render(S.cube_scene(prod, prog)), scenes.render(S.cube_scene(orc, prog)), 1, 5)
        rp = J.random_recursive_program(s)  # helpers + a self-recursive function with a random body (interpreted, or compiled per call depth)
        check(("program-recursive", s), scenes.render(J.grid_scene(prod, [rp], time=0.5)), scenes.render(J.grid_scene(orc, [rp], time=0.5)), 1, 6)
    except Exception as e:  # a refused program (RXR_ERR_UNSUPPORTED) is not a parity failure
        refused.append((s, str(e)[:80]))
    if s % 10 == 0:
        kw = dict(width=320 + 16 * (s % 5), height=200 + 8 * (s % 7), nx=10 + s % 23, ny=6 + s % 17, stacked=(s % 4) * 300)
        check(("tile-map-2d", s), scenes.render(scenes.tile_map_2d_scene(prod, **kw)), scenes.render(scenes.tile_map_2d_scene(orc, **kw)), 0, 0)
    if (s - first) % 20 == 19:
        print(f"... {s - first + 1} seeds, {len(bad)} failures so far", flush=True)
print("device projection" if DEVPROJ else "host projection", "seeds", first, "..", first + n - 1, "failures:", len(bad), "refused frames / programs:", len(refused), refused[:4])
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
