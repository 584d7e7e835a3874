"""The two arithmetic modes of the 3D direct-light loop (rxr_set_light_math, include/rxr.h; shade3d_lights<X, RL> in
rusterix_amd/csrc/rxr_kernels.hip) against the CPU oracle.

Reference path: src/rasterizer.rs:1373-1391 (the light loop), src/map/light.rs:491-552 (CompiledLight::radiance_at /
calculate_point_light), src/rasterizer.rs:1875-1951 (shade_fast_brdf).  BASELINE.json's bar for lit, float-interpolated 3D
fragments is 1 per 8-bit channel:

  * EXACT    the reference's operations correctly rounded: a frame differs from the oracle only where log2f / exp2f / acosf of
             the two math libraries differ (a handful of pixels, by 1);
  * RELAXED  (the library's default) point lights through v_rsq_f32 products: nothing off by more than 1, and -- because an error
             of two ulp reaches a channel only next to a rounding boundary -- only a small fraction of the pixels off at all;
  * frames without a 3D light loop (unlit, 2D) are the SAME frame in both modes.
"""
import ctypes as C

import numpy as np
import pytest

import rusterix_amd
from rusterix_amd import binding as B
from rusterix_amd import scenes
from tests.test_gpu_parity import _light, assert_exact, scene_2d

pytestmark = pytest.mark.gpu

TOLERANCE = 1  # per 8-bit channel (BASELINE.json north_star, lit 3D paths)


@pytest.fixture()
def light_math(product, monkeypatch):
    monkeypatch.delenv("RXR_LIGHT_MATH", raising=False)

    def choose(exact):
        product.lib.rxh_set_light_math_exact(1 if exact else 0)
        assert product.lib.rxh_get_light_math_exact() == (1 if exact else 0)

    yield choose
    product.lib.rxh_set_light_math_exact(0)  # the default


def channel_diff(a, b):
    return np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=2)


def lit_box_grid(api):
    """a binned scene (k_raster_rows / k_raster_rows_rl): 24 x 24 boxes under three point lights, Linear sampling"""
    cfg = scenes.box_grid_scene(api, n=24, width=480, height=270)
    for pos, col in (((1.0, 1.5, 1.0), (1.0, 0.8, 0.6)), ((3.5, 1.0, 2.5), (0.4, 0.9, 1.0)), ((2.4, 0.8, 4.0), (0.9, 0.9, 0.3))):
        cfg.scene.add_dynamic_light(B.Light(B.LIGHT_POINT).with_position(pos).with_color(col).with_intensity(1.5)
                                    .with_start_distance(0.5).with_end_distance(4.0).compile())
    return cfg


def chunk_level_map(api):
    """the map scene with the grid shader as its background: feature level 1 (k_raster_chunk / k_raster_chunk_rl)"""
    cfg = scenes.map_scene(api, width=400, height=240, logo_size=64, n_lights=5)
    cfg.scene.background(api.GridShader())
    return cfg


def mixed_light_types(api):
    """point lights next to spot / area / daylight lights (which keep the exact arithmetic in both modes) and a flickering one"""
    cfg = scenes.map_scene(api, width=320, height=180, logo_size=64, n_lights=2)
    cfg.scene.lights([
        B.Light(B.LIGHT_AMBIENT).with_color((0.1, 0.2, 0.1)).with_intensity(0.5).compile(),
        _light(B.LIGHT_AREA, (3.0, 1.0, 12.0), normal=(0.0, 0.0, -1.0), end=6.0, start=1.0, intensity=1.5, width=2.0, height=1.0),
        _light(B.LIGHT_DAYLIGHT, (7.0, 3.0, 10.0), normal=(0.0, -1.0, 0.0), end=12.0, start=2.0, intensity=0.8),
    ])
    cfg.scene.add_dynamic_light(B.Light(B.LIGHT_POINT).with_position((2.0, 1.0, 6.0)).with_color((0.9, 0.3, 0.2)).with_intensity(1.2)
                                .with_start_distance(1.0).with_end_distance(5.0).with_flicker(0.5).compile())
    # start == end: the smoothstep's denominator is zero (never evaluated: a fragment is either in full range or out of it)
    cfg.scene.add_dynamic_light(B.Light(B.LIGHT_POINT).with_position((11.0, 1.0, 4.0)).with_color((0.2, 0.6, 0.9)).with_intensity(0.8)
                                .with_start_distance(3.0).with_end_distance(3.0).compile())
    return cfg


def far_from_the_origin(api):
    """a lit, textured cube 3000 units from the origin seen from 1.6 units away: an ulp of the world position is 2.4e-4 there, the
    view vector's angle error from a relaxed world position ~1e-4 -- the case the scale-aware flip guard of shade3d_begin is for"""
    c = (3000.0, 1800.0, -2500.0)
    box = (api.Batch3D.from_box(c[0] - 0.5, c[1] - 0.5, c[2] - 0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
           .source(B.PixelSource.StaticTileIndex(0)))
    scene = api.Scene.from_static([], [box]).background(api.VGrayGradientShader())
    scene.lights([B.Light(B.LIGHT_POINT).with_position((c[0] + 1.2, c[1] + 1.5, c[2] + 0.8)).with_color((1.0, 0.9, 0.7)).with_intensity(2.0)
                  .with_start_distance(0.5).with_end_distance(6.0).compile(),
                  B.Light(B.LIGHT_POINT).with_position((c[0] - 1.4, c[1] + 0.2, c[2] + 1.1)).with_color((0.5, 0.7, 1.0)).with_intensity(1.5)
                  .with_start_distance(0.5).with_end_distance(5.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.logo_texture(1, 64))])
    cam = api.D3OrbitCamera.new()
    cam.center = c
    cam.set_parameter_f32("distance", 1.6)

    def setup():
        v, p = cam.matrices(400.0, 300.0)
        return api.Rasterizer.setup(None, v, p).ambient((0.3, 0.3, 0.3, 1.0))

    return scenes._result(api, scene, assets, setup, 400, 300, 40, "far cube")


SCENES = {
    "map, 1 light": lambda api: scenes.map_scene(api, width=640, height=360, logo_size=128, n_lights=1),
    "map, 16 lights": lambda api: scenes.map_scene(api, width=640, height=360, logo_size=128, n_lights=16),
    "map, 4 lights, Linear": lambda api: scenes.map_scene(api, width=400, height=240, logo_size=128, n_lights=4, sample_mode=B.SAMPLE_LINEAR),
    "teapot with a point light": lambda api: scenes.teapot_scene(api, width=480, height=270, with_light=True, logo_size=64),
    "lit box grid (binned, row mode)": lit_box_grid,
    "map behind the grid background (feature level 1)": chunk_level_map,
    "mixed light types": mixed_light_types,
    "lit cube far from the origin": far_from_the_origin,
}


@pytest.mark.parametrize("name", list(SCENES))
def test_both_modes_against_the_oracle(oracle, product, light_math, name):
    build = SCENES[name]
    ref = scenes.render(build(oracle)).copy()
    light_math(True)
    exact = scenes.render(build(product)).copy()
    light_math(False)
    relaxed = scenes.render(build(product)).copy()
    n = ref.shape[0] * ref.shape[1]
    lit_pixels = int((ref[..., :3].max(axis=2) > 0).sum())
    assert lit_pixels > n // 10, f"{name}: the scene shows nothing"
    d_exact, d_relaxed, d_modes = channel_diff(exact, ref), channel_diff(relaxed, ref), channel_diff(relaxed, exact)
    assert int(d_exact.max()) <= TOLERANCE and int(d_relaxed.max()) <= TOLERANCE and int(d_modes.max()) <= TOLERANCE, \
        f"{name}: max channel difference exact {int(d_exact.max())}, relaxed {int(d_relaxed.max())}, between the modes {int(d_modes.max())}"
    # exact: only libm differences (pow32_fast's log2f / exp2f); relaxed: a rounding boundary has to be within ~1e-6 of the value
    assert int((d_exact > 0).sum()) <= 2 + n // 50_000, f"{name}: exact mode, {int((d_exact > 0).sum())} of {n} pixels differ from the oracle"
    assert int((d_relaxed > 0).sum()) <= 4 + n // 2_000, f"{name}: relaxed mode, {int((d_relaxed > 0).sum())} of {n} pixels differ from the oracle"
    # alpha, coverage and depth are untouched by the mode
    assert np.array_equal(relaxed[..., 3], exact[..., 3])


def test_frames_without_a_3d_light_loop_do_not_depend_on_the_mode(oracle, product, light_math):
    builders = {
        "unlit textured cube": lambda api: scenes.cube_scene(api, width=320, height=240, tile_size=40, textured=True, distance=3.0, logo_size=64, rect_size=40.0),
        "teapot, ambient only": lambda api: scenes.teapot_scene(api, width=320, height=180, logo_size=64, rect_size=40.0),
        "2D with 2D lights": lambda api: scene_2d(api, lights=True),
    }
    for what, build in builders.items():
        ref = scenes.render(build(oracle)).copy()
        light_math(True)
        exact = scenes.render(build(product)).copy()
        light_math(False)
        relaxed = scenes.render(build(product)).copy()
        assert_exact(relaxed, exact, f"{what}: relaxed vs exact")
        assert_exact(exact, ref, f"{what}: vs the oracle")


@pytest.mark.parametrize("name", ["map, 16 lights", "lit box grid (binned, row mode)", "map behind the grid background (feature level 1)"])
def test_the_exact_normal_sequences_inside_the_relaxed_kernels(oracle, product, light_math, monkeypatch, name):
    """the relaxed kernels decide the normal's flip toward the camera from relaxed values only where |n.v| >= the guard (1e-4); a
    guard of 0.99 (RXR_RL_FLIP_GUARD, a test knob) sends practically every wave down the fallback -- the exact normalisations
    followed by the fused light term -- which must stay inside the same bars"""
    build = SCENES[name]
    ref = scenes.render(build(oracle)).copy()
    light_math(False)
    relaxed = scenes.render(build(product)).copy()
    monkeypatch.setenv("RXR_RL_FLIP_GUARD", "0.99")
    guarded = scenes.render(build(product)).copy()
    n = ref.shape[0] * ref.shape[1]
    d_ref, d_modes = channel_diff(guarded, ref), channel_diff(guarded, relaxed)
    assert int(d_ref.max()) <= TOLERANCE and int(d_modes.max()) <= TOLERANCE
    assert int((d_ref > 0).sum()) <= 4 + n // 2_000, f"{name}: {int((d_ref > 0).sum())} of {n} pixels differ from the oracle"
    assert np.array_equal(guarded[..., 3], relaxed[..., 3])


# ---- scenes built to sit on the relaxed arithmetic's weak spots (round-2 verdict, item 4) -----------------------------------------
def _room(api, lights, width=480, height=270, ambient=(0.05, 0.05, 0.05, 1.0), cam_pos=(6.0, 1.0, 4.5), cam_dir=(0.0349, 0.0, 0.9994), offset=(0.0, 0.0, 0.0)):
    """a floor and a facing wall (both large, flat, textured) under `lights`; `offset` moves room, lights and camera together"""
    o = np.array(offset, np.float32)
    floor_v = np.array([(0, 0, 0, 1), (15, 0, 0, 1), (15, 0, 15, 1), (0, 0, 15, 1)], np.float32)
    floor_v[:, :3] += o
    floor = (api.Batch3D.new(floor_v, np.array([(0, 1, 2), (0, 2, 3)], np.uint32), np.array([(0, 0), (15, 0), (15, 15), (0, 15)], np.float32))
             .source(B.PixelSource.StaticTileIndex(scenes.MAP_TILES["brickfloor"])).repeat_mode(B.REPEAT_REPEAT_XY).with_computed_normals())
    wv, wi, wuv = scenes._quad_wall(15, 12, 0, 12, 4.0)
    wv[:, :3] += o
    wall = api.Batch3D.new(wv, wi, wuv).source(B.PixelSource.StaticTileIndex(scenes.MAP_TILES["brickwall"])).repeat_mode(B.REPEAT_REPEAT_XY).with_computed_normals()
    scene = api.Scene.from_static([], [floor, wall]).background(api.VGrayGradientShader())
    moved = []
    for kw in lights:
        kw = dict(kw)
        pos = tuple(float(x) for x in (np.array(kw.pop("pos"), np.float32) + o))
        l = (B.Light(B.LIGHT_POINT).with_position(pos).with_color(kw.pop("color", (1.0, 0.9, 0.8))).with_intensity(kw.pop("intensity", 1.5))
             .with_start_distance(kw.pop("start", 1.0)).with_end_distance(kw.pop("end", 6.0)))
        assert not kw
        moved.append(l.compile())
    scene.lights(moved)
    assets = scenes.map_assets(api, 64)
    cam = api.D3FirstPCamera.new()
    pos = np.array(cam_pos, np.float32) + o
    cam.position = tuple(pos)
    cam.center = tuple(pos + np.array(cam_dir, np.float32))

    def setup():
        v, p = cam.matrices(float(width), float(height))
        return api.Rasterizer.setup(None, v, p).ambient(ambient)

    return scenes._result(api, scene, assets, setup, width, height, 40, "adversarial room")


def grazing_incidence(api):
    """lights a hair above, exactly in and a hair below the floor's plane and the wall's plane: n.l runs through 0 (the Lambert
    cut-off, the clamp of the fused term) across the frame, and n.h with it"""
    return _room(api, [dict(pos=(6.0, 0.004, 8.0), end=9.0, intensity=3.0), dict(pos=(3.0, 0.0, 9.5), end=9.0, intensity=3.0),
                       dict(pos=(9.0, -0.003, 7.0), end=9.0, intensity=3.0),
                       dict(pos=(5.0, 1.0, 11.998), end=8.0, intensity=2.0, color=(0.6, 0.8, 1.0)), dict(pos=(8.0, 1.5, 12.0), end=8.0, intensity=2.0),
                       dict(pos=(7.0, 2.0, 12.004), end=8.0, intensity=2.0)])


def rounded_cube_terminator(api):
    """a cube with averaged vertex normals (they interpolate like a sphere's): every face has a terminator line n.l = 0 running through
    it, three close lights"""
    box = (api.Batch3D.from_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals().source(B.PixelSource.StaticTileIndex(0)))
    scene = api.Scene.from_static([], [box]).background(api.VGrayGradientShader())
    scene.lights([B.Light(B.LIGHT_POINT).with_position(p).with_color(c).with_intensity(2.5).with_start_distance(0.2).with_end_distance(4.0).compile()
                  for p, c in (((0.9, 0.1, 0.2), (1.0, 0.8, 0.6)), ((-0.2, 0.95, 0.7), (0.5, 0.9, 1.0)), ((0.1, -0.3, 1.1), (0.9, 0.9, 0.4)))])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.logo_texture(1, 64))])
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 2.2)

    def setup():
        v, p = cam.matrices(400.0, 300.0)
        return api.Rasterizer.setup(None, v, p).ambient((0.02, 0.02, 0.02, 1.0))

    return scenes._result(api, scene, assets, setup, 400, 300, 40, "rounded cube")


def smoothstep_tails(api):
    """range spheres that cut through the flat floor and wall -- the rim of every lit disk, where the smoothstep leaves 0, is in the
    frame -- with falloff bands from generous to a thousandth of a unit (start - end = -1e-3: the reciprocal amplifies the distance's
    rounding a thousandfold)"""
    return _room(api, [dict(pos=(4.0, 1.0, 9.0), start=1.0, end=3.0), dict(pos=(8.0, 1.2, 8.0), start=2.9, end=3.0, color=(0.5, 1.0, 0.6)),
                       dict(pos=(6.0, 0.8, 10.5), start=2.499, end=2.5, color=(0.6, 0.7, 1.0)), dict(pos=(10.5, 1.0, 10.0), start=0.0, end=2.75),
                       dict(pos=(2.0, 2.0, 11.0), start=3.2, end=3.2001, color=(1.0, 0.5, 0.5))], ambient=(0.1, 0.1, 0.1, 1.0))


def sixteen_overlapping_lights(api, intensity):
    """sixteen point lights that ALL reach every fragment: sixteen relaxed terms accumulate in one pixel (dim ones: the sum stays in the
    middle of the byte range, where a step is smallest relative to the value; bright ones: channels run into the clamp)"""
    rng = np.random.default_rng(77)
    return _room(api, [dict(pos=(float(3 + 9 * rng.random()), float(0.3 + 2.5 * rng.random()), float(5 + 6.5 * rng.random())), start=1.0, end=40.0,
                            intensity=intensity, color=tuple(float(c) for c in 0.4 + 0.6 * rng.random(3))) for _ in range(16)], ambient=(0.0, 0.0, 0.0, 1.0))


def far_camera_near_lights(api):
    """the whole room 10^4 units from the origin (an ulp of a coordinate is 1e-3 there: a thousandth of the light distances) with lights a
    unit or two from the surfaces"""
    return _room(api, [dict(pos=(6.0, 1.0, 9.0), start=0.5, end=5.0, intensity=2.0), dict(pos=(8.5, 0.5, 10.5), start=0.5, end=4.0, color=(0.5, 0.8, 1.0)),
                       dict(pos=(4.0, 1.8, 11.0), start=0.5, end=4.0, color=(1.0, 0.6, 0.5))], offset=(10000.0, -6000.0, 8000.0))


ADVERSARIAL = {
    "grazing incidence on flat surfaces": grazing_incidence,
    "terminator lines on a rounded cube": rounded_cube_terminator,
    "smoothstep tails cutting through flat surfaces": smoothstep_tails,
    "16 dim lights over every fragment": lambda api: sixteen_overlapping_lights(api, 0.05),
    "16 bright lights over every fragment": lambda api: sixteen_overlapping_lights(api, 4.0),
    "room 10^4 units from the origin": far_camera_near_lights,
}


@pytest.mark.parametrize("name", list(ADVERSARIAL))
def test_adversarial_scenes_stay_within_one_step_in_both_modes(oracle, product, light_math, name):
    """Scenes placed on the rounding boundaries of the relaxed light loop: n.l and n.h crossing 0, the smoothstep leaving 0 (down to a
    falloff band a thousandth of a unit wide), sixteen terms accumulated per pixel, coordinates whose ulp is a thousandth of the light
    distance.  The tolerance of BASELINE.json for lit 3D fragments -- one step per channel against the CPU oracle -- must hold in BOTH
    modes; if a case ever exceeds it, relaxed has to stop being the library's default."""
    build = ADVERSARIAL[name]
    ref = scenes.render(build(oracle)).copy()
    light_math(True)
    exact = scenes.render(build(product)).copy()
    light_math(False)
    relaxed = scenes.render(build(product)).copy()
    n = ref.shape[0] * ref.shape[1]
    assert int((ref[..., :3].max(axis=2) > 0).sum()) > n // 10, f"{name}: the scene shows nothing"
    assert len(np.unique(ref[..., :3])) > 32, f"{name}: the scene has no gradients to sit on"
    d_exact, d_relaxed = channel_diff(exact, ref), channel_diff(relaxed, ref)
    print(f"{name}: pixels off by 1 -- exact {int((d_exact > 0).sum())}, relaxed {int((d_relaxed > 0).sum())} of {n}")
    assert int(d_exact.max()) <= TOLERANCE, f"{name}: exact mode is off by {int(d_exact.max())}"
    assert int(d_relaxed.max()) <= TOLERANCE, f"{name}: relaxed mode is off by {int(d_relaxed.max())}"
    assert np.array_equal(relaxed[..., 3], exact[..., 3]) and np.array_equal(exact[..., 3], ref[..., 3])
    # even here the relaxed frame is the reference's frame but for a scattering of last-step roundings
    assert int((d_relaxed > 0).sum()) <= 4 + n // 2_000, f"{name}: relaxed mode, {int((d_relaxed > 0).sum())} of {n} pixels differ from the oracle"


def test_lights_with_non_finite_parameters_take_the_exact_loop(oracle, product, light_math):
    """intensity = inf on a light whose range ends before every fragment: the reference skips the light (distance >= end_distance,
    light.rs:539) and so does the exact loop; the fused relaxed term would compute inf * 0.  rxr_upload_frame sends frames with a
    non-finite light parameter through the exact kernels in either mode (round-2 advisor finding)"""
    def build(api):
        cfg = scenes.map_scene(api, width=320, height=180, logo_size=64, n_lights=3)
        cfg.scene.add_dynamic_light(B.Light(B.LIGHT_POINT).with_position((7.0, 30.0, 8.0)).with_color((1.0, 1.0, 1.0)).with_intensity(float("inf"))
                                    .with_start_distance(1.0).with_end_distance(2.0).compile())
        return cfg

    ref = scenes.render(build(oracle)).copy()
    frames = []
    for exact in (True, False):
        light_math(exact)
        frames.append(scenes.render(build(product)).copy())
        assert int(channel_diff(frames[-1], ref).max()) <= TOLERANCE, f"exact={exact}: a light out of range with infinite intensity changed the frame"
    assert_exact(frames[1], frames[0], "a frame with a non-finite light parameter is the exact frame in both modes")


def test_the_environment_overrides_the_context_mode(product, light_math, monkeypatch):
    build = SCENES["map, 16 lights"]
    light_math(True)
    exact = scenes.render(build(product)).copy()
    light_math(False)
    relaxed = scenes.render(build(product)).copy()
    monkeypatch.setenv("RXR_LIGHT_MATH", "exact")   # context: relaxed
    assert_exact(scenes.render(build(product)).copy(), exact, "RXR_LIGHT_MATH=exact over a relaxed context")
    light_math(True)
    monkeypatch.setenv("RXR_LIGHT_MATH", "relaxed")  # context: exact
    assert_exact(scenes.render(build(product)).copy(), relaxed, "RXR_LIGHT_MATH=relaxed over an exact context")


def test_the_bench_frame_in_both_modes(oracle, product, light_math):
    """BASELINE.json configs[3] at full size: the relaxed kernel really is another computation (some pixels differ from the exact
    frame), and stays inside the tolerance with a few dozen pixels off"""
    kw = dict(width=3840, height=2160, n_lights=16)
    ref = scenes.render(scenes.map_scene(oracle, **kw)).copy()
    light_math(True)
    exact = scenes.render(scenes.map_scene(product, **kw)).copy()
    light_math(False)
    relaxed = scenes.render(scenes.map_scene(product, **kw)).copy()
    d_exact, d_relaxed, d_modes = channel_diff(exact, ref), channel_diff(relaxed, ref), channel_diff(relaxed, exact)
    assert int(d_exact.max()) <= TOLERANCE and int((d_exact > 0).sum()) <= 16
    assert int(d_relaxed.max()) <= TOLERANCE and int((d_relaxed > 0).sum()) <= 256
    assert 0 < int((d_modes > 0).sum()) <= 256 and int(d_modes.max()) <= TOLERANCE


def test_an_unknown_mode_is_refused(product):
    rxr = rusterix_amd.rxr_abi()
    ctx = product.lib.rxh_context()
    assert ctx, product.lib.rxh_last_error()
    assert rxr.rxr_set_light_math(ctx, 7) == B.RXR_ERR_INVALID
    assert b"rxr_set_light_math" in rxr.rxr_last_error(ctx)
    assert rxr.rxr_set_light_math(None, 0) == B.RXR_ERR_INVALID
    assert rxr.rxr_set_light_math(ctx, 1) == 0  # RXR_LIGHT_MATH_RELAXED, the default


def test_members_of_a_multi_device_context_use_the_mode(product, light_math):
    """three logical members on GPU 0: the frame is the single-context frame of the same mode, stripe for stripe"""
    build = SCENES["map, 16 lights"]
    frames = {}
    for exact in (True, False):
        light_math(exact)
        product.lib.rxh_set_device(0)
        frames[exact] = scenes.render(build(product)).copy()
    try:
        ids = (C.c_int * 3)(0, 0, 0)
        for exact in (True, False):
            product.lib.rxh_set_devices(ids, 3)
            light_math(exact)
            assert_exact(scenes.render(build(product)).copy(), frames[exact], f"3 members, exact={exact}")
    finally:
        product.lib.rxh_set_device(0)
