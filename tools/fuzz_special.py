#!/usr/bin/env python3
"""Wide sweep of tests/test_gpu_special_inputs.build_random (random combinations of poisoned numbers), host- and device-projected.
usage: python tools/fuzz_special.py [first_seed] [n_seeds]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_special_inputs as SI  # noqa: E402

prod, orc = rusterix_amd.load(), load_oracle()
prod.lib.rxh_set_device_projection.argtypes = [C.c_int]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = []
for s in range(first, first + n):
    ref = scenes.render(SI.build_random(orc, s))
    for dp in (0, 1):
        prod.lib.rxh_set_device_projection(dp)
        try:
            got = scenes.render(SI.build_random(prod, s))
        except Exception as e:
            bad.append((s, dp, str(e)[:80]))
            continue
        finally:
            prod.lib.rxh_set_device_projection(0)
        d = (got != ref).any(axis=2)
        if d.any():
            y, x = np.argwhere(d)[0]
            bad.append((s, dp, int(d.sum()), (int(y), int(x)), got[y, x].tolist(), ref[y, x].tolist()))
    if (s - first) % 50 == 49:
        print(f"... {s - first + 1} seeds, {len(bad)} failures so far", flush=True)
print("special-value sweep seeds", first, "..", first + n - 1, "failures:", len(bad))
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
