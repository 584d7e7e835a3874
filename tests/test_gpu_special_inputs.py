"""Special values IN THE SCENE -- NaN, +-inf, +-0, a denormal, the largest float in a vertex position or a texture coordinate of one
triangle after the other -- through the whole path (projection, clipping, set-up, visibility, sampling) against the oracle, bit for
bit: Nearest and Linear sampling, every repeat mode, opaque and cut-out textures, host- and device-projected.  The wide fuzz sweep
found the NaN-coordinate rule of bilinear sampling (tests/test_gpu_rows.py) after ~4 000 seeds; this builds such triangles on purpose."""
import ctypes as C

import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from tests.test_gpu_fuzz import random_texture

pytestmark = pytest.mark.gpu

SPECIALS = [float("nan"), float("inf"), float("-inf"), 0.0, -0.0, 1e-40, 3.0e38, -3.0e38]
W, H = 208, 144


def build(api, sample_mode, repeat_mode, cutout, seed=7, shape=(9, 7), mode="plain"):
    rng = np.random.default_rng([0x52585231, 606, seed])
    textures = [B.Tile([random_texture(rng, shape[0], shape[1], 0)]), B.Tile([random_texture(rng, shape[1], shape[0], 2)])]
    assets = api.Assets.default().textures(textures)
    scene = api.Scene.empty()
    # a backdrop far away, so that a fragment that must NOT be written shows as the backdrop and not as the miss colour
    back = np.array([[-9, -7, -6.0, 1], [9, -7, -6.0, 1], [9, 7, -6.0, 1], [-9, 7, -6.0, 1]], np.float32)
    b = api.Batch3D.new(back, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), np.array([[0, 0], [3, 0], [3, 3], [0, 3]], np.float32)).with_computed_normals().cull_mode(0)
    b.source(B.PixelSource.Pixel((40, 90, 160, 255))).ambient_color((1.0, 1.0, 1.0))
    scene.add_d3_static(b)
    # one batch per (field, special): a triangle of ordinary size with ONE poisoned number
    # mode: "plain" | "opacity" (the triangles go to a chunk's opacity list: d3_rasterize_opacity and the blend) | "program" (every
    # triangle runs a Rusteria program that reads uv and the texel) | "program_alpha" (... and writes `opacity`: the alpha of a
    # fragment then needs the whole front half of the fragment block, DB_FULL_ALPHA)
    chunk = scene.add_chunk() if mode == "opacity" else None
    prog = None
    if mode == "program":
        prog = scene.add_program(B.Program([["Color", "UV", ("Push", 0.5), "Mul", ("Push", 0.5), "Add", "Mul", "SetColor"]]))
    elif mode == "program_alpha":
        prog = scene.add_program(B.Program([["Color", "SetColor", "UV", ("GetComponents", [0]), ("Push", 2.0), "Mul", "Fract", ("Push", 0.5), "Gt", "SetOpacity"]]))
    k = 0
    for field in range(5):                       # x, y, z of vertex 1; u, v of vertex 2
        for s in SPECIALS:
            cx, cy = -2.6 + 0.75 * (k % 8), -1.6 + 0.8 * (k // 8)
            v = np.array([[cx, cy, -1.0 - 0.05 * k, 1], [cx + 0.6, cy + 0.1, -1.2 - 0.05 * k, 1], [cx + 0.2, cy + 0.6, -0.9 - 0.05 * k, 1]], np.float32)
            uv = np.array([[0.1, 0.2], [1.7, 0.3], [0.4, 1.9]], np.float32)
            if field < 3:
                v[1, field] = s
            else:
                uv[2, field - 3] = s
            t = api.Batch3D.new(v, np.array([[0, 1, 2]], np.uint32), uv).with_computed_normals().cull_mode(0)
            t.source(B.PixelSource.StaticTileIndex(1 if cutout else 0)).repeat_mode(repeat_mode).ambient_color((0.9, 0.8, 0.7))
            if prog is not None:
                t.shader(prog)
            if chunk is not None:
                chunk.add_batch3d_opacity(t)
            else:
                scene.add_d3_static(t)
            k += 1
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 3.0)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.1

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).sample_mode(sample_mode).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, assets, setup, W, H, 40, "special-inputs")


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("cutout", [False, True])
@pytest.mark.parametrize("repeat_mode", [B.REPEAT_CLAMP_XY, B.REPEAT_REPEAT_XY, B.REPEAT_REPEAT_X, B.REPEAT_REPEAT_Y])
@pytest.mark.parametrize("sample_mode", [B.SAMPLE_NEAREST, B.SAMPLE_LINEAR])
def test_poisoned_triangles(oracle, product, sample_mode, repeat_mode, cutout, device_projection):
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(build(product, sample_mode, repeat_mode, cutout))
    finally:
        product.lib.rxh_set_device_projection(0)
    ref = scenes.render(build(oracle, sample_mode, repeat_mode, cutout))
    assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 20   # the ordinary triangles are there
    d = (got != ref).any(axis=2)
    assert not d.any(), f"{int(d.sum())} pixels differ; first at {np.argwhere(d)[:3].tolist()}: device {got[tuple(np.argwhere(d)[0])].tolist()} oracle {ref[tuple(np.argwhere(d)[0])].tolist()}"


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (5, 1), (2, 2)])
@pytest.mark.parametrize("sample_mode", [B.SAMPLE_NEAREST, B.SAMPLE_LINEAR])
def test_poisoned_triangles_on_degenerate_textures(oracle, product, sample_mode, shape):
    """the same with textures of one texel, one row, one column (x1 = min(x0 + 1, w - 1) of the bilinear taps, texture.rs:414-460)"""
    for repeat_mode in (B.REPEAT_CLAMP_XY, B.REPEAT_REPEAT_XY):
        for cutout in (False, True):
            got = scenes.render(build(product, sample_mode, repeat_mode, cutout, shape=shape))
            ref = scenes.render(build(oracle, sample_mode, repeat_mode, cutout, shape=shape))
            d = (got != ref).any(axis=2)
            assert not d.any(), f"texture {shape}, repeat {repeat_mode}, cutout {cutout}: {int(d.sum())} pixels differ; first at {np.argwhere(d)[:3].tolist()}"


@pytest.mark.parametrize("mode", ["opacity", "program", "program_alpha"])
@pytest.mark.parametrize("sample_mode", [B.SAMPLE_NEAREST, B.SAMPLE_LINEAR])
def test_poisoned_triangles_in_the_other_passes(oracle, product, sample_mode, mode):
    """the poisoned triangles as opacity batches (blended over the backdrop) and under Rusteria programs (uv and texel read; `opacity`
    written: the visibility pass then runs the front half of the fragment block), cut-out and opaque textures, two repeat modes"""
    for repeat_mode in (B.REPEAT_CLAMP_XY, B.REPEAT_REPEAT_XY):
        for cutout in (False, True):
            got = scenes.render(build(product, sample_mode, repeat_mode, cutout, mode=mode))
            ref = scenes.render(build(oracle, sample_mode, repeat_mode, cutout, mode=mode))
            d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
            assert d.max() == 0, f"{mode}, repeat {repeat_mode}, cutout {cutout}: {int((d > 0).sum())} pixels differ (max {int(d.max())}); first at {np.argwhere(d > 0)[:3].tolist()}: device {got[tuple(np.argwhere(d > 0)[0])].tolist()} oracle {ref[tuple(np.argwhere(d > 0)[0])].tolist()}"


def build_random(api, seed):
    """several poisoned numbers at once, chosen by the seed: any position / w / uv entry of any vertex of ~30 triangles, random sampling
    and repeat modes, opaque / cut-out textures, plain / opacity / program passes, a cull mode per batch"""
    rng = np.random.default_rng([0x52585231, 1212, seed])
    shape = [(9, 7), (1, 1), (3, 1), (8, 8)][int(rng.integers(0, 4))]
    textures = [B.Tile([random_texture(rng, shape[0], shape[1], 0)]), B.Tile([random_texture(rng, shape[1], shape[0], int(rng.integers(1, 3)))])]
    assets = api.Assets.default().textures(textures)
    scene = api.Scene.empty()
    back = np.array([[-9, -7, -6.0, 1], [9, -7, -6.0, 1], [9, 7, -6.0, 1], [-9, 7, -6.0, 1]], np.float32)
    b = api.Batch3D.new(back, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), np.array([[0, 0], [3, 0], [3, 3], [0, 3]], np.float32)).with_computed_normals().cull_mode(0)
    b.source(B.PixelSource.Pixel((40, 90, 160, 255))).ambient_color((1.0, 1.0, 1.0))
    scene.add_d3_static(b)
    chunk = scene.add_chunk() if rng.random() < 0.4 else None
    prog = scene.add_program(B.Program([["Color", "UV", ("Push", 0.5), "Mul", ("Push", 0.5), "Add", "Mul", "SetColor"]])) if rng.random() < 0.3 else None
    pool = SPECIALS + [1e30, -1e30, 0.1, 1.0]
    for k in range(30):
        cx, cy = float(rng.uniform(-2.6, 2.6)), float(rng.uniform(-1.6, 1.6))
        z = float(rng.uniform(-2.5, -0.4))
        v = np.array([[cx, cy, z, 1], [cx + rng.uniform(0.2, 0.9), cy + rng.uniform(-0.2, 0.3), z - rng.uniform(-0.3, 0.3), 1],
                      [cx + rng.uniform(-0.2, 0.4), cy + rng.uniform(0.3, 0.9), z + rng.uniform(-0.3, 0.3), 1]], np.float32)
        uv = rng.uniform(-0.5, 2.0, (3, 2)).astype(np.float32)
        for _ in range(int(rng.integers(0, 4))):
            if rng.random() < 0.6:
                v[int(rng.integers(0, 3)), int(rng.integers(0, 4))] = pool[int(rng.integers(0, len(pool)))]
            else:
                uv[int(rng.integers(0, 3)), int(rng.integers(0, 2))] = pool[int(rng.integers(0, len(pool)))]
        t = api.Batch3D.new(v, np.array([[0, 1, 2]], np.uint32), uv).with_computed_normals().cull_mode(int(rng.integers(0, 3)))
        t.source(B.PixelSource.StaticTileIndex(int(rng.integers(0, 2)))).repeat_mode(int(rng.integers(0, 4))).ambient_color(tuple(float(c) for c in rng.uniform(0.3, 1.0, 3)))
        if prog is not None and rng.random() < 0.5:
            t.shader(prog)
        if chunk is not None and rng.random() < 0.5:
            chunk.add_batch3d_opacity(t)
        else:
            scene.add_d3_static(t)
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", float(rng.uniform(2.0, 4.0)))
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = float(rng.uniform(-0.2, 0.3))
    sample_mode = int(rng.integers(0, 2))

    def setup():
        v_, p_ = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v_, p_).sample_mode(sample_mode).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, assets, setup, W, H, 40, f"special-random{seed}")


@pytest.mark.parametrize("seed", range(40))
def test_random_combinations_of_poisoned_numbers(oracle, product, seed):
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    for dp in (0, 1):
        product.lib.rxh_set_device_projection(dp)
        try:
            got = scenes.render(build_random(product, seed))
        finally:
            product.lib.rxh_set_device_projection(0)
        ref = scenes.render(build_random(oracle, seed))
        d = (got != ref).any(axis=2)
        assert not d.any(), f"seed {seed}, device projection {dp}: {int(d.sum())} pixels differ; first at {np.argwhere(d)[:3].tolist()}: device {got[tuple(np.argwhere(d)[0])].tolist()} oracle {ref[tuple(np.argwhere(d)[0])].tolist()}"


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("tile_size", [8, 40, 64, 300])
@pytest.mark.parametrize("huge", [-1.0e30, 1.0e30, -1.0e12, 1.0e7, -2.0e4])
def test_3d_batches_with_huge_projected_coordinates_follow_the_reference_s_tiles(oracle, product, huge, tile_size, device_projection):
    """the 3D analogue of tests/test_gpu_special_2d.py's huge-coordinate case: a triangle with one vertex far off to the side projects to
    a finite but enormous screen coordinate; its batch's box is Rect {x: min, width: max - min} and the reference skips the batch for
    every tile that the rounded `x + width` does not reach (rasterizer.rs:978-983) -- the device clips to the tiles that pass"""
    def build(api):
        scene = api.Scene.empty()
        back = np.array([[-9, -7, -6.0, 1], [9, -7, -6.0, 1], [9, 7, -6.0, 1], [-9, 7, -6.0, 1]], np.float32)
        b = api.Batch3D.new(back, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), np.zeros((4, 2), np.float32)).with_computed_normals().cull_mode(0)
        b.source(B.PixelSource.Pixel((40, 90, 160, 255))).ambient_color((1.0, 1.0, 1.0))
        scene.add_d3_static(b)
        for k, (axis, vert) in enumerate([(0, 1), (1, 2), (0, 0), (1, 1)]):
            v = np.array([[-0.8 + 0.5 * k, -0.6, -1.0, 1], [0.1 + 0.4 * k, -0.5, -1.3, 1], [-0.3 + 0.5 * k, 0.7, -0.9, 1]], np.float32)
            v[vert, axis] = huge
            t = api.Batch3D.new(v, np.array([[0, 1, 2]], np.uint32), np.array([[0, 0], [1, 0], [0, 1]], np.float32)).with_computed_normals().cull_mode(0)
            t.source(B.PixelSource.Pixel((250 - 50 * k, 60 + 40 * k, 90, 255))).ambient_color((1.0, 1.0, 1.0))
            scene.add_d3_static(t)
        cam = api.D3OrbitCamera.new()
        cam.set_parameter_f32("distance", 3.0)
        cam.azimuth = float(np.float32(np.pi / 2))

        def setup():
            v_, p_ = cam.matrices(float(W), float(H))
            return api.Rasterizer.setup(None, v_, p_).ambient((1.0, 1.0, 1.0, 1.0))

        return scenes._result(api, scene, api.Assets.default(), setup, W, H, tile_size, "huge-3d")

    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(build(product))
    finally:
        product.lib.rxh_set_device_projection(0)
    ref = scenes.render(build(oracle))
    d = (got != ref).any(axis=2)
    assert not d.any(), f"{huge}, tile {tile_size}: {int(d.sum())} pixels differ; first at {np.argwhere(d)[:3].tolist()}: device {got[tuple(np.argwhere(d)[0])].tolist()} oracle {ref[tuple(np.argwhere(d)[0])].tolist()}"
