#!/usr/bin/env python3
"""Debugging aid: one seed of the fuzz sweep's random 2D program (tests/test_gpu_shaders.ProgramGen) interpreted, compiled and on the
oracle.  usage: tools/one_program.py <seed> [--dump]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_shaders as S  # noqa: E402

prod, orc = rusterix_amd.load(), load_oracle()
s = int(sys.argv[1])
rng = np.random.default_rng([0x52585231, 4242, s])
gen = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3)))
prog = gen.program()
if "--dump" in sys.argv:
    for k, f in enumerate(gen.raw):
        print("function", k, f)
ref = scenes.render(S.rect_scene(orc, prog, time=0.5))
for mode in ("0", "1"):
    os.environ["RXR_SHADER_JIT"] = mode
    got = scenes.render(S.rect_scene(prod, prog, time=0.5))
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    print("RXR_SHADER_JIT", mode, "differ", int((d > 0).sum()), "max", int(d.max()), "first", np.argwhere(d > 0)[:2].tolist(),
          "got", got[tuple(np.argwhere(d > 0)[0])].tolist() if d.max() else None, "ref", ref[tuple(np.argwhere(d > 0)[0])].tolist() if d.max() else None)
