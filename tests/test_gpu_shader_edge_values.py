"""Every exact Rusteria opcode on a grid of SPECIAL operands -- +-0, +-1, +-inf, NaN, denormals, the largest floats -- on the device
(interpreter) against the oracle, bit for bit through encodings that expose what a plain colour hides: the sign of a zero (1 / r > 0),
NaN-ness (r != r), infinity, and the value itself.  The wide fuzz sweep found Max(+0, -0) this way by luck after 9 000 seeds
(tests/test_gpu_shader_jit.py, the signed-zero test); this grid finds that class of difference by construction.

The operands come out of the palette (PaletteIndex pushes the slot's three floats as they are): operand a by the pixel's column,
b by its row, c (ternary opcodes) by their sum."""
import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from rusterix_amd.binding import Program
from tests import test_gpu_shaders as S

pytestmark = pytest.mark.gpu

SPECIALS = [0.0, -0.0, 1.0, -1.0, 0.5, -2.5, float("inf"), float("-inf"), float("nan"), 1e-40, -1e-40, 3.4e38, 1e-20, 7.25, -3.0e38, 2.0]
N = len(SPECIALS)
CELL = 2 * N                      # pixels per cell edge: two pixels per operand (pixel centres stay clear of the index boundaries)
UNARY = S.EXACT_UNARY
BINARY = S.EXACT_BINARY
TERNARY = ["Mix", "Smoothstep"]


def operand(axis, shift=0):
    """palette slot floor(uv[axis] * 4 * N) (+ shift, mod N): the 2D pass hands uv / 4 to programs"""
    ops = ["UV", ("GetComponents", [axis]), ("Push", 4.0 * N), "Mul", "Floor"]
    if shift:
        ops += [("Push", float(shift)), "Add", ("Push", float(N)), "Mod"]
    return ops + ["PaletteIndex"]


def programs_for(op_ops, n_operands):
    """one program per (encoding, component): operands -> op -> encoding -> SetColor"""
    out = []
    load = operand(0)
    if n_operands >= 2:
        load = load + operand(1)
    if n_operands >= 3:
        load = load + ["UV", ("GetComponents", [0]), ("Push", 4.0 * N), "Mul", "Floor", "UV", ("GetComponents", [1]), ("Push", 4.0 * N), "Mul", "Floor", "Add",
                       ("Push", float(N)), "Mod", "PaletteIndex"]
    for comp in range(3):
        get = [("GetComponents", [comp])]
        r = load + op_ops + get + [("StoreLocal", 0)]
        L0 = [("LoadLocal", 0)]
        value = L0 + [("Push", 0.37), "Mul", ("Push", 0.11), "Add", "Fract"]
        sign = [("Push", 1.0)] + L0 + ["Div", ("Push", 0.0), "Gt"]            # 1 / r > 0: the sign, of a zero too
        isnan = L0 + L0 + ["Ne"]
        isinf = L0 + ["Abs", ("Push", 3.0e38), "Gt"]
        small = L0 + ["Abs", ("Push", 1e30), "Mul", ("Push", 0.37), "Mul", "Fract"]   # what survives of tiny results (denormals)
        out.append(Program([r + value + ["SetColor"]], shade_locals=1))
        out.append(Program([r + sign + isnan + isinf + ["Pack3", ("Push", 0.75), "Mul", "SetColor"]], shade_locals=1))
        out.append(Program([r + small + ["SetColor"]], shade_locals=1))
    return out


def grid(api, programs, cols):
    scene = api.Scene.empty()
    rows = (len(programs) + cols - 1) // cols
    for k, prog in enumerate(programs):
        idx = scene.add_program(prog)
        r = api.Batch2D.from_rectangle(float((k % cols) * CELL), float((k // cols) * CELL), float(CELL), float(CELL))
        r.source(B.PixelSource.Pixel((255, 255, 255, 255))).shader(idx)
        scene.add_d2_static(r)
    assets = api.Assets.default()
    # slot i holds (special i, special i + 5, special i + 11): the three components see different operands
    assets.palette([(SPECIALS[i], SPECIALS[(i + 5) % N], SPECIALS[(i + 11) % N]) for i in range(N)])
    w, h = cols * CELL, rows * CELL

    def setup():
        return api.Rasterizer.setup(None, B.Mat4.identity(), B.Mat4.identity())

    return scenes._result(api, scene, assets, setup, w, h, 40, "edge-values")


def run(oracle, product, monkeypatch, names, ops_of, n_operands):
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    programs, owner = [], []
    for name in names:
        ps = programs_for(ops_of(name), n_operands)
        programs += ps
        owner += [name] * len(ps)
    cols = 9
    got = scenes.render(grid(product, programs, cols))
    ref = scenes.render(grid(oracle, programs, cols))
    bad = {}
    d = (got != ref).any(axis=2)
    for y, x in np.argwhere(d):
        k = (y // CELL) * cols + (x // CELL)
        a, b = SPECIALS[(x % CELL) // 2], SPECIALS[(y % CELL) // 2]
        bad.setdefault(owner[k], []).append((k % 9, a, b, got[y, x].tolist(), ref[y, x].tolist()))
    assert not bad, "opcodes that differ from the oracle on special operands (program variant, a, b, device, oracle): " + \
        "; ".join(f"{n}: {v[:3]} (+{max(0, len(v) - 3)} more)" for n, v in bad.items())
    assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 8


def test_unary_opcodes_on_special_operands(oracle, product, monkeypatch):
    run(oracle, product, monkeypatch, UNARY, lambda n: [n], 1)


def test_binary_opcodes_on_special_operands(oracle, product, monkeypatch):
    run(oracle, product, monkeypatch, BINARY, lambda n: [n], 2)


def test_ternary_opcodes_on_special_operands(oracle, product, monkeypatch):
    run(oracle, product, monkeypatch, TERNARY, lambda n: [n], 3)


def test_libm_opcodes_on_special_operands(oracle, product, monkeypatch):
    """the libm-backed opcodes (glibc on the oracle's side, OCML on the device's: not bit-reproducible, judged at one step): what must
    agree EXACTLY is the class of the result -- NaN, infinite, the sign (of a zero too): log(0) = -inf, log(-1) = NaN, pow(0, 0) = 1,
    pow(-1, 0.5) = NaN, atan2(+-0, +-0), tan(inf) = NaN ..."""
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    names = [(n, 1) for n in S.LIBM_UNARY] + [(n, 2) for n in S.LIBM_BINARY]
    programs, owner = [], []
    for name, k in names:
        ps = programs_for([name], k)
        programs += ps
        owner += [(name, v % 3) for v in range(len(ps))]
    cols = 9
    got = scenes.render(grid(product, programs, cols))
    ref = scenes.render(grid(oracle, programs, cols))
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    bad_class, off = {}, {}
    for y, x in np.argwhere(d > 0):
        k = (y // CELL) * cols + (x // CELL)
        name, variant = owner[k]
        a, b = SPECIALS[(x % CELL) // 2], SPECIALS[(y % CELL) // 2]
        if variant == 1:      # sign / NaN / inf flags
            bad_class.setdefault(name, []).append((a, b, got[y, x].tolist(), ref[y, x].tolist()))
        elif d[y, x] > 1:
            off.setdefault(name, []).append((a, b, got[y, x].tolist(), ref[y, x].tolist()))
    assert not bad_class, "libm opcodes whose result CLASS (sign, NaN, inf) differs (a, b, device, oracle): " + "; ".join(f"{n}: {v[:4]}" for n, v in bad_class.items())
    # the value encodings wrap (fract): an ulp next to a wrap point is a full swing; a handful of such operand pairs at most
    assert sum(len(v) for v in off.values()) <= 24, "libm opcodes off by more than one step on special operands: " + "; ".join(f"{n}: {v[:3]} ({len(v)})" for n, v in off.items())


@pytest.mark.parametrize("which", ["unary", "binary"])
def test_compiled_opcodes_on_special_operands(oracle, product, monkeypatch, which):
    """the same grid through the run-time compiled form of the programs (rxr_jit_ops.h: its own implementations of the opcodes): equal
    to the interpreted frame and to the oracle"""
    names, k = (UNARY, 1) if which == "unary" else (BINARY, 2)
    programs = []
    for name in names:
        programs += programs_for([name], k)
    cols = 9
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    interpreted = scenes.render(grid(product, programs, cols)).copy()
    monkeypatch.setenv("RXR_SHADER_JIT", "1")
    compiled = scenes.render(grid(product, programs, cols)).copy()
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    assert np.array_equal(compiled, interpreted), f"compiled and interpreted differ in {(compiled != interpreted).any(axis=2).sum()} pixels; first at {np.argwhere((compiled != interpreted).any(axis=2))[:3].tolist()}"
    assert np.array_equal(compiled, scenes.render(grid(oracle, programs, cols)))
