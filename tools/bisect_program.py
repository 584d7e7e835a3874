#!/usr/bin/env python3
"""Debugging aid: where does a random program (tests/test_gpu_shaders.ProgramGen, fuzz seed) first compute something else on the device
than on the oracle?  `shade` is cut at its statement boundaries (stack depth 0) and every local is shown as the colour.
usage: tools/bisect_program.py <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RXR_SHADER_JIT"] = "0"
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from rusterix_amd.binding import Program  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_shaders as S  # noqa: E402

prod, orc = rusterix_amd.load(), load_oracle()
s = int(sys.argv[1])
rng = np.random.default_rng([0x52585231, 4242, s])
gen = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3)))
prog = gen.program()
shade, funcs = gen.raw[0], gen.raw[1:]
n_locals = prog.shade_locals


def delta(op):
    name = op if isinstance(op, str) else op[0]
    if name in ("Push", "LoadLocal", "LoadGlobal") or name in S.SOURCES:
        return 1
    if name in ("StoreLocal", "StoreGlobal", "Clear", "If") or name.startswith("Set") and name != "SetComponents":
        return -1
    if name in S.UNARY or name in ("GetComponents",):
        return 0
    if name in S.BINARY or name == "SetComponents":
        return -1
    if name in ("Mix", "Smoothstep", "Clamp", "Pack3"):
        return -2
    if name == "FunctionCall":
        return 1 - op[1]
    if name == "For":
        return 0
    raise SystemExit(f"unknown op {op}")


def differ(ops, what):
    p = Program([ops] + funcs, shade_locals=n_locals)
    try:
        got = scenes.render(S.rect_scene(prod, p, time=0.5))
        ref = scenes.render(S.rect_scene(orc, p, time=0.5))
    except Exception as e:
        print(what, "->", str(e)[:100])
        return False
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    if d.max():
        y, x = np.argwhere(d > 0)[0]
        print(what, "DIFFER", int((d > 0).sum()), "pixels, max", int(d.max()), "at", (int(y), int(x)), "got", got[y, x].tolist(), "ref", ref[y, x].tolist())
    return bool(d.max())


depth, cuts = 0, []
for i, op in enumerate(shade):
    depth += delta(op)
    if depth == 0:
        cuts.append(i + 1)
print(len(shade), "ops,", len(cuts), "statement boundaries")
for c in cuts:
    bad = False
    for loc in range(n_locals):
        for scale in (1.0, 0.01):
            if differ(shade[:c] + [("LoadLocal", loc), ("Push", scale), "Mul", "Fract", "SetColor"], f"after op {c} ({str(shade[c - 1])[:50]}): local {loc} x {scale}"):
                bad = True
    if bad:
        print("first statement that differs ends at op", c)
        # the statement = [value ops ...] + a consumer; show the value, then (an If) each branch on its own, cut the same way
        start = max([b for b in cuts if b < c], default=0)
        stmt = shade[start:c]
        last = stmt[-1]
        if not isinstance(last, str) and last[0] == "If":
            for scale in (1.0, 0.01):
                differ(shade[:start] + stmt[:-1] + [("Push", scale), "Mul", "Fract", "SetColor"], f"  the condition x {scale}")
            for name, block in (("then", last[1]), ("else", last[2])):
                if not block:
                    continue
                d2, sub = 0, []
                for i, op in enumerate(block):
                    d2 += delta(op)
                    if d2 == 0:
                        sub.append(i + 1)
                for c2 in sub:
                    hit = False
                    for loc in range(3):
                        for scale in (1.0, 0.01):
                            if differ(shade[:start] + block[:c2] + [("LoadLocal", loc), ("Push", scale), "Mul", "Fract", "SetColor"], f"  {name} branch up to its op {c2} ({str(block[c2 - 1])[:60]}): local {loc} x {scale}"):
                                hit = True
                    if hit:
                        print("  ->", name, "branch: first differing statement:", block[([0] + sub)[sub.index(c2)]:c2])
                        break
        break
