"""Counts on either side of the device's internal thresholds (none of which the reference has): the light loop's chunks of 64 lights, the
small-scene path's 128 triangles (RXR_STAGE_TRIS), the 2D pass's 128 primitives before it bins, the per-triangle batch table's 16 384
triangles (RXR_TRI_INFO_MAX), the one-workgroup device projection's 1024 vertices / triangles, a scene without any batch, batches
without triangles -- against the oracle."""
import ctypes as C

import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from tests.test_gpu_rows import small_triangles

pytestmark = pytest.mark.gpu
W, H = 256, 160


@pytest.mark.parametrize("n_lights", [0, 1, 63, 64, 65, 130])
@pytest.mark.parametrize("exact", [False, True])
def test_light_counts_around_the_chunk_of_64(oracle, product, n_lights, exact):
    def build(api):
        cfg = scenes.map_scene(api, width=W, height=H, n_lights=1, logo_size=16)
        rng = np.random.default_rng([0x52585231, 808])
        lights = []
        for _ in range(n_lights):
            p = (float(rng.uniform(1, 14)), float(rng.uniform(0.3, 1.8)), float(rng.uniform(1, 14)))
            lights.append(B.Light(B.LIGHT_POINT).with_position(p).with_color(tuple(float(c) for c in rng.uniform(0.2, 1.0, 3))).with_intensity(float(rng.uniform(0.05, 0.4)))
                          .with_start_distance(0.5).with_end_distance(float(rng.uniform(2.0, 9.0))).compile())
        cfg.scene.lights(lights)
        return cfg

    product.lib.rxh_set_light_math_exact(1 if exact else 0)
    try:
        got = scenes.render(build(product))
    finally:
        product.lib.rxh_set_light_math_exact(0)
    ref = scenes.render(build(oracle))
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    assert d.max() <= 1 and (d > 0).sum() <= 200, f"{n_lights} lights: {int((d > 1).sum())} pixels beyond one step, {int((d > 0).sum())} differ"


def mesh_scene(api, counts, two_d=0):
    rng = np.random.default_rng([0x52585231, 909, sum(counts) + two_d])
    scene = api.Scene.empty()
    for nt in counts:
        if nt == 0:
            v4, idx, uv = np.zeros((3, 4), np.float32) + np.float32([0, 0, 0, 1]), np.zeros((0, 3), np.uint32), np.zeros((3, 2), np.float32)
        else:
            v4, idx, uv = small_triangles(rng, nt, 0.08 if nt > 2000 else 0.25, 1.4)
        b = api.Batch3D.new(v4, idx, uv).with_computed_normals().cull_mode(0)
        b.source(B.PixelSource.Pixel(tuple(int(x) for x in rng.integers(30, 256, 3)) + (255,))).ambient_color((0.9, 0.9, 0.9))
        scene.add_d3_static(b)
    for k in range(two_d):
        x, y = float(rng.uniform(0, W - 20)), float(rng.uniform(0, H - 20))
        r = api.Batch2D.from_rectangle(x, y, float(rng.uniform(4, 40)), float(rng.uniform(4, 40))).source(B.PixelSource.Pixel(tuple(int(c) for c in rng.integers(0, 256, 3)) + (int(rng.integers(60, 256)),)))
        scene.add_d2_static(r)
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 3.0)

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, "thresholds")


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("counts", [[], [0], [0, 0, 5, 0], [127], [128], [129], [64, 64], [64, 65], [1023], [1024], [1025], [600, 424], [600, 425], [16384], [16385], [9000, 7384], [9000, 7385]])
def test_triangle_counts_around_the_internal_thresholds(oracle, product, counts, device_projection):
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(mesh_scene(product, counts))
    finally:
        product.lib.rxh_set_device_projection(0)
    ref = scenes.render(mesh_scene(oracle, counts))
    assert np.array_equal(got, ref), f"triangle counts {counts}: {(got != ref).any(axis=2).sum()} pixels differ"


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("two_d", [1, 63, 64, 65, 511, 512, 513, 600])   # rectangles: two primitives each (128 = 64 rectangles; the sort capacity is 1024)
def test_2d_primitive_counts_around_the_binning_threshold(oracle, product, two_d, device_projection):
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(mesh_scene(product, [40], two_d))
    finally:
        product.lib.rxh_set_device_projection(0)
    ref = scenes.render(mesh_scene(oracle, [40], two_d))
    assert np.array_equal(got, ref), f"{two_d} rectangles: {(got != ref).any(axis=2).sum()} pixels differ"


@pytest.mark.parametrize("n_rect", [511, 512, 513, 700])
def test_more_2d_primitives_in_one_tile_than_the_sort_holds(oracle, product, n_rect):
    """every rectangle covers the whole frame: each tile lists 2 * n_rect primitives -- at 1024 (RXR_SORT2D_MAX) the tile's LDS sort is
    full and the pass walks the frame's primitives in order instead; blended in submission order either way"""
    def build(api):
        rng = np.random.default_rng([0x52585231, 1001, n_rect])
        scene = api.Scene.empty()
        for k in range(n_rect):
            r = api.Batch2D.from_rectangle(float(-k % 3), float(-k % 2), float(W + 3), float(H + 2)).source(B.PixelSource.Pixel(tuple(int(c) for c in rng.integers(0, 256, 3)) + (int(rng.integers(1, 40)),)))
            scene.add_d2_static(r)

        def setup():
            v, p = api.D3OrbitCamera.new().matrices(float(W), float(H))
            return api.Rasterizer.setup(None, v, p).render_mode(B.RenderMode.render_2d()).background((9, 9, 9, 255))

        return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, "2d-stack")

    got, ref = scenes.render(build(product)), scenes.render(build(oracle))
    assert np.array_equal(got, ref), f"{n_rect} stacked rectangles: {(got != ref).any(axis=2).sum()} pixels differ"
