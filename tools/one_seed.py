#!/usr/bin/env python3
"""Debugging aid: one seed of tests/test_gpu_rows.build on the product against the oracle.  usage: tools/one_seed.py <device_projection 0|1> <variant> <seed...>"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_rows as R  # noqa: E402

prod, orc = rusterix_amd.load(), load_oracle()
dp, variant, seeds = int(sys.argv[1]), sys.argv[2], [int(x) for x in sys.argv[3:]]
prod.lib.rxh_set_device_projection.argtypes = [C.c_int]
prod.lib.rxh_set_device_projection(dp)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RXR_"))
for s in seeds:
    ww, hh = 203 + 16 * (s % 3), 131 + 9 * (s % 3)
    got = scenes.render(R.build(prod, s, ww, hh, variant))
    ref = scenes.render(R.build(orc, s, ww, hh, variant))
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    if d.max() > 0:
        y, x = np.argwhere(d > 0)[0]
        print(f"[{tag}] seed {s} {variant}: {int((d > 0).sum())} differ, max {int(d.max())}, first at y={y} x={x}: got {got[y, x].tolist()} ref {ref[y, x].tolist()}; "
              f"neighbours got {got[y, max(x - 1, 0)].tolist()} {got[y, min(x + 1, ww - 1)].tolist()} ref {ref[y, max(x - 1, 0)].tolist()} {ref[y, min(x + 1, ww - 1)].tolist()}")
    else:
        print(f"[{tag}] seed {s} {variant}: identical")
