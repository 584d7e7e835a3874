#!/usr/bin/env python3
"""Wide sweep of 2D scenes with random poisoned numbers: rectangles, triangles and (finite, moderate) segments with NaN / +-inf / +-0 /
denormal / huge entries in vertices and texture coordinates, a Mat3 that is absent, ordinary or poisoned, both sampling modes, tiny
textures, every repeat mode, alpha everywhere, the reference's tile size from 8 to 300 -- host- and device-projected, bit-exact.
usage: python tools/fuzz_special3.py [first_seed] [n_seeds]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rusterix_amd  # noqa: E402
from rusterix_amd import binding as B  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests.test_gpu_fuzz import random_texture  # noqa: E402

NAN, INF = float("nan"), float("inf")
POOL = [NAN, INF, -INF, 0.0, -0.0, 1e-40, 3.0e38, -3.0e38, 1e12, -1e12, 5.0e6, -2.5e6, 0.5, 1.0e30]
LINE_POOL = [0.0, -0.0, 1e-40, 3.0e4, -3.0e4, 0.49, -0.51, 1.0e3]   # (the oracle walks every segment once per tile: moderate lengths)
W, H = 208, 136


def build(api, seed):
    rng = np.random.default_rng([0x52585231, 1414, seed])
    pick = lambda pool: pool[int(rng.integers(0, len(pool)))]   # noqa: E731
    shapes = [(9, 7), (1, 1), (1, 4), (6, 6)]
    assets = api.Assets.default().textures([B.Tile([random_texture(rng, *shapes[int(rng.integers(0, 4))], int(rng.integers(0, 3)))]) for _ in range(3)])
    batches = []
    if rng.random() < 0.7:
        batches.append(api.Batch2D.from_rectangle(0.0, 0.0, float(W), float(H)).source(B.PixelSource.Pixel(tuple(int(c) for c in rng.integers(0, 256, 3)) + (int(rng.integers(100, 256)),))))
    for _ in range(int(rng.integers(2, 9))):
        kind = rng.random()
        if kind < 0.25:
            x, y = float(rng.uniform(-20, W)), float(rng.uniform(-20, H))
            b = api.Batch2D.from_rectangle(x, y, float(rng.uniform(2, 90)), float(rng.uniform(2, 90)))
        elif kind < 0.8:
            n = int(rng.integers(1, 4))
            v = rng.uniform(-20, 230, (3 * n, 2)).astype(np.float32)
            uv = rng.uniform(-0.5, 2.0, (3 * n, 2)).astype(np.float32)
            for _ in range(int(rng.integers(0, 3))):
                (v if rng.random() < 0.6 else uv)[int(rng.integers(0, 3 * n)), int(rng.integers(0, 2))] = pick(POOL)
            b = api.Batch2D.new(v, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), uv)
        else:
            v = rng.uniform(-10, 220, (4, 2)).astype(np.float32)
            if rng.random() < 0.6:
                v[int(rng.integers(0, 4)), int(rng.integers(0, 2))] = pick(LINE_POOL)
            b = api.Batch2D.new(v, np.array([[0, 1, 0], [2, 3, 0]], np.uint32), np.zeros_like(v)).mode([B.MODE_LINES, B.MODE_LINE_STRIP, B.MODE_LINE_LOOP][int(rng.integers(0, 3))])
        if rng.random() < 0.5:
            b.source(B.PixelSource.StaticTileIndex(int(rng.integers(0, 4))))     # (3: a missing tile)
        else:
            b.source(B.PixelSource.Pixel(tuple(int(c) for c in rng.integers(0, 256, 3)) + (int(rng.integers(0, 256)),)))
        b.repeat_mode(int(rng.integers(0, 4)))
        batches.append(b)
    scene = api.Scene.from_static(batches, [])
    m = None
    r = rng.random()
    if r < 0.35:
        m = [[float(rng.uniform(0.5, 1.6)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-20, 20))], [float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.5, 1.6)), float(rng.uniform(-20, 20))], [0.0, 0.0, 1.0]]
        if rng.random() < 0.4:
            m[int(rng.integers(0, 2))][int(rng.integers(0, 3))] = pick([1e-40, 0.0, -0.0, 2.0, -2.0])   # (finite: segments must keep finite end points)
    matrix = B.Mat3.from_rows(m) if m is not None else None
    sample_mode = int(rng.integers(0, 2))
    preserve = bool(rng.random() < 0.3)
    bg = tuple(int(c) for c in rng.integers(0, 256, 4))
    tile = [8, 40, 64, 300][int(rng.integers(0, 4))]

    def setup():
        v_, p_ = api.D3OrbitCamera.new().matrices(float(W), float(H))
        rr = api.Rasterizer.setup(matrix, v_, p_).render_mode(B.RenderMode.render_2d()).sample_mode(sample_mode).background(bg)
        if preserve:
            rr.preserve_transparency(True)
        return rr

    return scenes._result(api, scene, assets, setup, W, H, tile, f"special-2d-{seed}")


if __name__ == "__main__":
    prod, orc = rusterix_amd.load(), load_oracle()
    prod.lib.rxh_set_device_projection.argtypes = [C.c_int]
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad, refused = [], 0
    for s in range(first, first + n):
        ref = None
        for dp in (0, 1):
            prod.lib.rxh_set_device_projection(dp)
            try:
                got = scenes.render(build(prod, s))
            except B.RasterizeError as e:
                if e.code == B.RXR_ERR_UNSUPPORTED and "line end point" in str(e):
                    refused += 1     # (a matrix that sends an end point beyond +-2^30: the oracle's walk would not end either)
                    continue
                bad.append((s, dp, str(e)[:90]))
                continue
            finally:
                prod.lib.rxh_set_device_projection(0)
            if ref is None:
                ref = scenes.render(build(orc, s))
            d = (got != ref).any(axis=2)
            if d.any():
                y, x = np.argwhere(d)[0]
                bad.append((s, dp, int(d.sum()), (int(y), int(x)), got[y, x].tolist(), ref[y, x].tolist()))
        if (s - first) % 100 == 99:
            print(f"... {s - first + 1} seeds, {len(bad)} failures so far", flush=True)
    print("2D special-value sweep seeds", first, "..", first + n - 1, "failures:", len(bad), "refused:", refused)
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
