"""Seeded random scenes: GPU (through the C ABI) against the CPU oracle.

Random triangle soups that cross the near plane and the eye plane (w ~ 0 gives inf / NaN screen
coordinates, which the reference handles only through its comparison semantics), every repeat / cull /
sample mode, textures with random alpha, constant-colour sources, random lights of every type, 2D
overlays.  Colour paths contain log2/exp2/acos, so the bar is +-1 per channel (TOLERANCE) with at most a
handful of pixels where acos decides cone membership."""
import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes

pytestmark = pytest.mark.gpu

TOLERANCE = 1


def random_texture(rng, w, h, alpha_mode):
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    if alpha_mode == 0:
        img[..., 3] = 255
    elif alpha_mode == 1:
        img[..., 3] = np.where(rng.random((h, w)) < 0.4, 255, rng.integers(0, 255, (h, w)))
    else:
        img[..., 3] = np.where(rng.random((h, w)) < 0.5, 255, 0)
    return B.Texture(img.reshape(-1), w, h)


def build(api, seed, width, height):
    rng = np.random.default_rng([0x52585231, seed])
    textures = [B.Tile([random_texture(rng, int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(0, 3)))
                        for _ in range(int(rng.integers(1, 4)))]) for _ in range(4)]
    assets = api.Assets.default().textures(textures)
    scene = api.Scene.empty()
    n_batches = int(rng.integers(1, 5))
    for _ in range(n_batches):
        nt = int(rng.integers(1, 40))
        centre = rng.normal(0.0, 1.2, size=(nt, 1, 3))
        verts = (centre + rng.normal(0.0, 0.8, size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
        if rng.random() < 0.3:   # some vertices exactly on the eye plane / far away
            verts[rng.integers(0, len(verts))] = (0.0, 0.0, 3.0)
            verts[rng.integers(0, len(verts))] *= np.float32(50.0)
        v4 = np.concatenate([verts, np.ones((len(verts), 1), np.float32)], axis=1)
        idx = np.arange(nt * 3, dtype=np.uint32).reshape(nt, 3)
        uv = (rng.random((nt * 3, 2)) * 3.0 - 1.0).astype(np.float32)
        b = api.Batch3D.new(v4, idx, uv).with_computed_normals()
        b.cull_mode(int(rng.integers(0, 3))).repeat_mode(int(rng.integers(0, 4)))
        kind = rng.integers(0, 4)
        if kind == 0:
            b.source(B.PixelSource.Pixel(tuple(int(x) for x in rng.integers(0, 256, 3)) + (255,)))
        elif kind == 1:
            b.source(B.PixelSource.Off)
        else:
            b.source(B.PixelSource.StaticTileIndex(int(rng.integers(0, 4))))
        b.ambient_color(tuple(float(x) for x in rng.random(3) * 0.5))
        if rng.random() < 0.3:
            b.transform(B.Mat4.scaling_3d((float(rng.uniform(0.5, 1.5)), float(rng.uniform(-1.5, 1.5)), 1.0)))
        (scene.add_d3_static if rng.random() < 0.7 else scene.add_d3_dynamic)(b)
    lights = []
    for _ in range(int(rng.integers(0, 6))):
        l = B.Light(int(rng.integers(0, 6))).with_position(tuple(float(x) for x in rng.normal(0, 2, 3)))
        l.with_color(tuple(float(x) for x in rng.random(3))).with_intensity(float(rng.uniform(0.2, 3.0)))
        l.with_start_distance(float(rng.uniform(0.2, 2.0))).with_end_distance(float(rng.uniform(2.0, 8.0)))
        l.with_flicker(float(rng.choice([0.0, 0.0, 0.4])))
        l.direction = tuple(float(x) for x in rng.normal(0, 1, 3))
        l.normal = tuple(float(x) for x in rng.normal(0, 1, 3))
        l.cone_angle = float(rng.uniform(0.2, 1.2))
        l.width, l.height = float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3))
        l.from_linedef = bool(rng.random() < 0.3)
        lights.append(l.compile())
    scene.lights(lights)
    if rng.random() < 0.7:
        r = api.Batch2D.from_rectangle(float(rng.integers(0, 40)), float(rng.integers(0, 30)), float(rng.integers(5, 60)), float(rng.integers(5, 40)))
        r.source(B.PixelSource.StaticTileIndex(int(rng.integers(0, 4)))).repeat_mode(int(rng.integers(0, 4)))
        scene.add_d2_static(r)
    scene.set_animation_frame(int(rng.integers(0, 9)))
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 3.0)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.0
    sample = int(rng.integers(0, 2))
    ambient = rng.random() < 0.6
    sun = rng.random() < 0.3
    amb = tuple(float(x) for x in rng.random(4))
    sun_dir = tuple(float(x) for x in rng.normal(0, 1, 3))

    def setup():
        v, p = cam.matrices(float(width), float(height))
        ra = api.Rasterizer.setup(None, v, p).sample_mode(sample)
        if ambient:
            ra.ambient(amb)
        if sun:
            ra.sun(sun_dir, 0.7)
        return ra

    return scenes._result(api, scene, assets, setup, width, height, 40, f"fuzz{seed}")


@pytest.mark.parametrize("seed", range(24))
def test_random_scene(oracle, product, seed):
    w, h = 160 + 16 * (seed % 3), 100 + 7 * (seed % 4)
    got = scenes.render(build(product, seed, w, h))
    ref = scenes.render(build(oracle, seed, w, h))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    bad = np.argwhere(diff > TOLERANCE)
    assert len(bad) <= 3, f"seed {seed}: {len(bad)} pixels off by more than {TOLERANCE}; first {bad[:3].tolist()} gpu={got[tuple(bad[0])]} oracle={ref[tuple(bad[0])]}"
    assert (diff > 0).mean() < 0.02, f"seed {seed}: {(diff > 0).sum()} pixels differ"


@pytest.mark.parametrize("seed", range(6))
def test_random_scene_device_projection(product, seed):
    import ctypes as C

    lib = product.lib
    lib.rxh_set_device_projection.argtypes = [C.c_int]
    w, h = 176, 107
    lib.rxh_set_device_projection(0)
    want = scenes.render(build(product, 100 + seed, w, h)).copy()
    try:
        lib.rxh_set_device_projection(1)
        got = scenes.render(build(product, 100 + seed, w, h)).copy()
    finally:
        lib.rxh_set_device_projection(0)
    assert np.array_equal(got, want), f"seed {seed}: {(got != want).any(axis=2).sum()} pixels differ between host and device projection"


def build_chunks(api, seed, width, height, dense=1):
    """random chunked scenes: several chunks with opacity-pass batches, profile ids shared between panes and walls, terrain
    textures (some with holes) sampled by world position, per-chunk programs and baked shader textures, a 2D overlay"""
    from tests.test_gpu_shaders import ProgramGen

    rng = np.random.default_rng([0x52585231, 9090, seed])
    setters = ["SetColor", "SetRoughness", "SetMetallic"]   # what every program of the scene may write (and none reads)
    scene = api.Scene.empty()
    textures = [B.Tile([random_texture(rng, int(rng.integers(4, 40)), int(rng.integers(4, 40)), int(rng.integers(0, 3)))]) for _ in range(3)]
    assets = api.Assets.default().textures(textures)
    if rng.random() < 0.5:
        scene.add_program(ProgramGen(rng, 3, 0, setters).program())   # scene.shaders[0]: chunk batches must not pick it up

    def soup(nt, spread=1.0):
        nt = nt * dense                                  # dense > 1: many small triangles -> the binned pipeline and row mode
        centre = rng.normal(0.0, 1.0 * spread, size=(nt, 1, 3))
        verts = (centre + rng.normal(0.0, 0.7 if dense == 1 else 0.15, size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
        v4 = np.concatenate([verts, np.ones((len(verts), 1), np.float32)], axis=1)
        idx = np.arange(nt * 3, dtype=np.uint32).reshape(nt, 3)
        uv = (rng.random((nt * 3, 2)) * 2.0).astype(np.float32)
        return api.Batch3D.new(v4, idx, uv).with_computed_normals().cull_mode(int(rng.integers(0, 3)))

    for c in range(int(rng.integers(1, 4))):
        chunk = scene.add_chunk()
        size = int(rng.integers(2, 9))
        if rng.random() < 0.7:
            tw = int(rng.integers(8, 48))
            chunk.terrain(random_texture(rng, tw, int(rng.integers(8, 48)), int(rng.integers(0, 2))), origin=(int(rng.integers(-3, 3)), int(rng.integers(-3, 3))), size=size)
        else:
            chunk.terrain(None, origin=(0, 0), size=size)
        n_shaders = int(rng.integers(0, 3))
        for _ in range(n_shaders):
            baked = random_texture(rng, 16, 16, int(rng.integers(0, 3))) if rng.random() < 0.4 else None
            chunk.add_shader(ProgramGen(rng, 3, int(rng.integers(0, 2)), setters).program(), baked)
        for _ in range(int(rng.integers(0, 3))):      # opacity pass
            b = soup(int(rng.integers(1, 6))).source(B.PixelSource.StaticTileIndex(int(rng.integers(0, 3))))
            if rng.random() < 0.7:
                b.profile_id(int(rng.integers(0, 3)))
            if n_shaders and rng.random() < 0.4:
                b.shader(int(rng.integers(0, n_shaders)))
            chunk.add_batch3d_opacity(b)
        for _ in range(int(rng.integers(1, 4))):      # opaque
            b = soup(int(rng.integers(2, 12)))
            kind = rng.integers(0, 3)
            b.source(B.PixelSource.Terrain() if kind == 0 else B.PixelSource.StaticTileIndex(int(rng.integers(0, 3))) if kind == 1
                     else B.PixelSource.Pixel(tuple(int(x) for x in rng.integers(0, 256, 3)) + (255,)))
            b.repeat_mode(int(rng.integers(0, 4)))
            if rng.random() < 0.6:
                b.profile_id(int(rng.integers(0, 3)))
            if n_shaders and rng.random() < 0.5:
                b.shader(int(rng.integers(0, n_shaders)))
            chunk.add_batch3d(b)
        if rng.random() < 0.5:
            chunk.terrain_batch3d(soup(int(rng.integers(2, 8)), 1.5).source(B.PixelSource.Terrain()))
    for _ in range(int(rng.integers(0, 2))):
        b = soup(int(rng.integers(2, 10))).source(B.PixelSource.Pixel((60, 200, 90, 255)))
        if rng.random() < 0.5:
            b.profile_id(int(rng.integers(0, 3)))
        scene.add_d3_static(b)
    lights = []
    for _ in range(int(rng.integers(0, 3))):
        l = B.Light(B.LIGHT_POINT).with_position(tuple(float(x) for x in rng.normal(0, 2, 3)))
        l.with_color(tuple(float(x) for x in rng.random(3))).with_intensity(float(rng.uniform(0.5, 2.5)))
        l.with_start_distance(float(rng.uniform(0.2, 2.0))).with_end_distance(float(rng.uniform(3.0, 8.0)))
        lights.append(l.compile())
    scene.lights(lights)
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 4.0)
    cam.azimuth = float(rng.uniform(0.0, 6.28))
    cam.elevation = float(rng.uniform(-0.5, 0.9))
    amb = tuple(float(x) for x in rng.random(4))

    # the editor's brush preview on some seeds (its own random stream: the scenes of the other seeds stay what they were)
    rng_b = np.random.default_rng([0x52585231, 9191, seed])
    brush = None
    if rng_b.random() < 0.35:
        brush = (tuple(float(x) for x in rng_b.normal(0.0, 1.5, 3)), float(rng_b.uniform(0.5, 4.0)), float(rng_b.choice([0.0, 0.3, 1.0, 2.0])))

    def setup():
        v, p = cam.matrices(float(width), float(height))
        r = api.Rasterizer.setup(None, v, p).ambient(amb).sample_mode(int(rng.integers(0, 1)) if False else 0).time(0.25)
        if brush is not None:
            r.brush_preview(*brush)
        return r

    return scenes._result(api, scene, assets, setup, width, height, 40, f"fuzz-chunks{seed}")


@pytest.mark.parametrize("seed", list(range(16)) + [1043])   # 1043: an opaque batch with a profile id before the only opacity batch
def test_random_chunk_scene(oracle, product, seed):
    w, h = 168, 104
    got = scenes.render(build_chunks(product, seed, w, h))
    ref = scenes.render(build_chunks(oracle, seed, w, h))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    bad = np.argwhere(diff > TOLERANCE)
    assert len(bad) <= 3, f"seed {seed}: {len(bad)} pixels off by more than {TOLERANCE}; first {bad[:3].tolist()} gpu={got[tuple(bad[0])]} oracle={ref[tuple(bad[0])]}"


@pytest.mark.parametrize("seed", list(range(8)) + [843, 886])
def test_random_chunk_scene_binned(oracle, product, seed):
    """the same generator with 40 times as many, smaller triangles: more than 128 triangles per frame, so the chunk / program
    kernels (k_raster_chunk, k_raster_vm) go through the binned pipeline, its row-parallel visibility and -- with opacity
    batches or full-alpha candidates in a round -- the walk"""
    w, h = 168, 104
    got = scenes.render(build_chunks(product, 200 + seed, w, h, dense=40))
    ref = scenes.render(build_chunks(oracle, 200 + seed, w, h, dense=40))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    bad = np.argwhere(diff > TOLERANCE)
    assert len(bad) <= 3, f"seed {seed}: {len(bad)} pixels off by more than {TOLERANCE}; first {bad[:3].tolist()} gpu={got[tuple(bad[0])]} oracle={ref[tuple(bad[0])]}"
