#!/usr/bin/env python3
"""Wide sweep of random Rusteria programs (tests/test_gpu_shaders.ProgramGen, with helper functions) whose constants are special values
-- NaN, +-inf, +-0, denormals, the largest floats -- a good part of the time: 2D rectangle shaders, interpreted, against the oracle,
bit for bit; a program that faults (Clamp with unordered bounds ...) must fault on both sides.
usage: python tools/fuzz_special4.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RXR_SHADER_JIT", "0")
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_shaders as S  # noqa: E402

NAN, INF = float("nan"), float("inf")
POOL = [NAN, INF, -INF, 0.0, -0.0, 1e-40, -1e-40, 3.4e38, -3.4e38, 1e-20, 1.0, -1.0, 0.5, 2.0, 1e30]


class SpecialGen(S.ProgramGen):
    def value(self, depth):
        if self.rng.random() < 0.12:
            return [("Push", *[POOL[int(self.rng.integers(0, len(POOL)))] for _ in range(3)])]
        return super().value(depth)


if __name__ == "__main__":
    prod, orc = rusterix_amd.load(), load_oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad, faults = [], 0
    for s in range(first, first + n):
        rng = np.random.default_rng([0x52585231, 1515, s])
        prog = SpecialGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3))).program()
        res = []
        for api in (prod, orc):
            try:
                res.append(scenes.render(S.rect_scene(api, prog, time=0.5)).copy())
            except Exception as e:
                res.append(str(e)[:70])
        got, ref = res
        if isinstance(got, str) or isinstance(ref, str):
            if isinstance(got, str) and isinstance(ref, str):
                faults += 1
            else:
                bad.append((s, "one side faulted", got if isinstance(got, str) else "device rendered", ref if isinstance(ref, str) else "oracle rendered"))
            continue
        d = (got != ref).any(axis=2)
        if d.any():
            y, x = np.argwhere(d)[0]
            bad.append((s, int(d.sum()), (int(y), int(x)), got[y, x].tolist(), ref[y, x].tolist()))
        if (s - first) % (10 if os.environ.get("RXR_SHADER_JIT") == "1" else 100) == 9 or (s - first) % 100 == 99:
            print(f"... {s - first + 1} seeds, {len(bad)} failures so far, {faults} programs faulted on both sides", flush=True)
    print("program special-value sweep seeds", first, "..", first + n - 1, "failures:", len(bad), "faulted on both sides:", faults)
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
