"""Chunk paths of the raster loops on the device against the oracle: terrain batches and the
terrain texture sampled by world position (src/chunk.rs:133-151, src/rasterizer.rs:343-356, :515-525, :1189-1191), per-chunk
shader programs (chunk.shaders) and baked shader textures (chunk.shader_textures, :1226-1267)."""
import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from rusterix_amd.binding import Program

pytestmark = pytest.mark.gpu

TOLERANCE = 1
W, H = 208, 144


def terrain_texture(tag, w, h, holes=False):
    rng = np.random.default_rng([0x52585231, tag])
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    img[..., 3] = 255
    if holes:
        yy, xx = np.mgrid[0:h, 0:w]
        img[((xx // 3 + yy // 3) % 3) == 0, 3] = 90
    return B.Texture(img.reshape(-1), w, h)


def floor_quad(api, x0, z0, x1, z1, y=0.0):
    v = np.array([[x0, y, z0, 1], [x1, y, z0, 1], [x1, y, z1, 1], [x0, y, z1, 1]], np.float32)
    i = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    return api.Batch3D.new(v, i, uv).with_computed_normals().cull_mode(B.CULL_OFF)


def terrain_scene(api, holes=False, with_texture=True, lights=True, origin=(0, 0), size=8, tex_w=64, brush=None, under=True, opacity_terrain=False):
    scene = api.Scene.empty()
    if under:
        scene.add_d3_static(floor_quad(api, -2.0, -2.0, 10.0, 10.0, y=-0.5).source(B.PixelSource.Pixel((40, 60, 200, 255))))
    chunk = scene.add_chunk()
    chunk.terrain(terrain_texture(11, tex_w, tex_w, holes) if with_texture else None, origin=origin, size=size)
    chunk.terrain_batch3d(floor_quad(api, 0.0, 0.0, 8.0, 8.0).source(B.PixelSource.Terrain()))
    chunk.add_batch3d(api.Batch3D.from_box(3.0, 0.0, 3.0, 1.0, 1.0, 1.0).with_computed_normals().source(B.PixelSource.Pixel((200, 180, 40, 255))))
    if opacity_terrain:   # a terrain-textured pane in the opacity pass (rasterizer.rs:1596-1625)
        chunk.add_batch3d_opacity(floor_quad(api, 1.0, 5.0, 7.0, 7.5, y=0.6).source(B.PixelSource.Terrain()))
    if lights:
        scene.lights([B.Light(B.LIGHT_POINT).with_position((4.0, 2.0, 4.0)).with_color((1.0, 0.9, 0.8)).with_intensity(2.0)
                      .with_start_distance(1.0).with_end_distance(9.0).compile()])
    assets = api.Assets.default()
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 9.0)
    cam.center = (4.0, 0.0, 4.0)
    cam.azimuth = 1.1
    cam.elevation = 0.9

    def setup():
        v, p = cam.matrices(float(W), float(H))
        r = api.Rasterizer.setup(None, v, p).ambient((0.6, 0.6, 0.6, 1.0))
        if brush is not None:
            r.brush_preview(*brush)
        return r

    return scenes._result(api, scene, assets, setup, W, H, 40, "chunk-terrain")


def compare(oracle, product, build, tol=0):
    got = scenes.render(build(product))
    ref = scenes.render(build(oracle))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    assert int(diff.max()) <= tol, f"{(diff > tol).sum()} pixels differ by more than {tol} (max {diff.max()}); first at {np.argwhere(diff > tol)[:3].tolist()}"
    return got


def test_terrain_batch_samples_the_terrain_texture_by_world_position(oracle, product):
    got = compare(oracle, product, lambda api: terrain_scene(api), tol=TOLERANCE)
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 500, "the terrain texture should be visible"


def test_terrain_unlit_is_bit_exact(oracle, product):
    compare(oracle, product, lambda api: terrain_scene(api, lights=False))


@pytest.mark.parametrize("origin, size, tex_w", [((2, -3), 4, 64), ((0, 0), 16, 40), ((-8, -8), 3, 100)])
def test_terrain_origin_size_and_odd_texture_sizes(oracle, product, origin, size, tex_w):
    compare(oracle, product, lambda api: terrain_scene(api, lights=False, origin=origin, size=size, tex_w=tex_w))


def test_terrain_texels_with_alpha_are_cut_out(oracle, product):
    """texels whose alpha is not 255 are not written (src/rasterizer.rs:1408): the blue floor underneath shows through"""
    got = compare(oracle, product, lambda api: terrain_scene(api, holes=True, lights=False))
    blue = (got[..., 2] > got[..., 0] + 60) & (got[..., 2] > got[..., 1] + 60)
    assert 0.05 < blue.mean() < 0.6


def test_terrain_source_without_a_terrain_texture_draws_nothing(oracle, product):
    got = compare(oracle, product, lambda api: terrain_scene(api, with_texture=False, lights=False))
    assert ((got[..., 2] > got[..., 0] + 60) & (got[..., 2] > got[..., 1] + 60)).mean() > 0.3   # only the floor underneath and the box


def test_terrain_batch2d(oracle, product):
    def build(api):
        scene = api.Scene.empty()
        chunk = scene.add_chunk()
        chunk.terrain(terrain_texture(12, 48, 48), origin=(1, 2), size=6)
        chunk.add_batch2d(api.Batch2D.from_rectangle(10.0, 10.0, 60.0, 40.0).source(B.PixelSource.Pixel((255, 0, 0, 255))))
        chunk.terrain_batch2d(api.Batch2D.from_rectangle(0.0, 0.0, float(W), float(H)).source(B.PixelSource.Terrain()))
        scene.add_d2_static(api.Batch2D.from_rectangle(100.0, 60.0, 50.0, 50.0).source(B.PixelSource.Pixel((0, 255, 0, 128))))
        m2d = B.Mat3.from_rows([[24.0, 0.0, 30.0], [0.0, 24.0, 20.0], [0.0, 0.0, 1.0]])   # 24 pixels per world unit

        def setup():
            return api.Rasterizer.setup(m2d, B.Mat4.identity(), B.Mat4.identity())

        return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, "chunk-terrain-2d")

    got = compare(oracle, product, build)
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 300


def chunk_shader_scene(api, baked, opacity_list=False, two_chunks=False):
    scene = api.Scene.empty()
    scene.add_program(Program([[("Push", 0.0, 0.0, 1.0), "SetColor"]]))   # scene.shaders[0]: must NOT be what chunk batches use
    tint = Program([["Color", ("Push", 1.0, 0.3, 0.3), "Mul", "UV", ("Push", 2.0), "Mul", "Add", "SetColor"]])
    green = Program([["Color", ("Push", 0.2, 1.0, 0.2), "Mul", "SetColor", "UV", ("GetComponents", [0]), ("Push", 8.0), "Mul", "Fract", "SetRoughness"]])
    chunk = scene.add_chunk()
    s0 = chunk.add_shader(tint, terrain_texture(21, 16, 16, holes=False) if baked else None)
    box = api.Batch3D.from_box(-1.2, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
    box.source(B.PixelSource.StaticTileIndex(0)).repeat_mode(B.REPEAT_REPEAT_XY).shader(s0)
    (chunk.add_batch3d_opacity if opacity_list else chunk.add_batch3d)(box)
    rect = api.Batch2D.from_rectangle(4.0, 4.0, 40.0, 30.0).source(B.PixelSource.StaticTileIndex(0)).shader(s0)
    chunk.add_batch2d(rect)
    target = scene.add_chunk() if two_chunks else chunk
    s1 = target.add_shader(green, None)
    box2 = api.Batch3D.from_box(0.2, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
    box2.source(B.PixelSource.StaticTileIndex(0)).repeat_mode(B.REPEAT_REPEAT_XY).shader(s1)
    target.add_batch3d(box2)
    scene.add_d3_static(floor_quad(api, -3.0, -3.0, 3.0, 3.0, y=-0.6).source(B.PixelSource.Pixel((90, 90, 90, 255))))
    scene.lights([B.Light(B.LIGHT_POINT).with_position((0.5, 1.5, 2.0)).with_color((1.0, 1.0, 0.9)).with_intensity(2.0)
                  .with_start_distance(1.0).with_end_distance(8.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.noise_texture(8, 32, 32))])
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 3.2)
    cam.azimuth = 1.2
    cam.elevation = 0.5

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((0.4, 0.4, 0.4, 1.0))

    return scenes._result(api, scene, assets, setup, W, H, 40, "chunk-shaders")


@pytest.mark.parametrize("two_chunks", [False, True])
def test_chunk_programs(oracle, product, two_chunks):
    """chunk batches index chunk.shaders, not scene.shaders (src/rasterizer.rs:1285-1288)"""
    got = compare(oracle, product, lambda api: chunk_shader_scene(api, baked=False, two_chunks=two_chunks), tol=TOLERANCE)
    assert ((got[..., 2] > 200) & (got[..., 0] < 30) & (got[..., 1] < 30)).mean() < 0.01, "scene.shaders[0] (pure blue) must not run"


def test_baked_shader_texture_replaces_the_texel_in_the_opaque_pass_only(oracle, product):
    a = compare(oracle, product, lambda api: chunk_shader_scene(api, baked=True), tol=TOLERANCE)
    b = compare(oracle, product, lambda api: chunk_shader_scene(api, baked=False), tol=TOLERANCE)
    assert (a != b).any(axis=2).mean() > 0.02, "the baked texture should change the first box"
    # the 2D rectangle runs the program either way (src/rasterizer.rs:760-797 has no shader_textures)
    assert np.array_equal(a[6:30, 6:40], b[6:30, 6:40])


def test_baked_shader_texture_with_alpha(oracle, product):
    # a baked texture whose texels are not all opaque: its alpha decides whether the fragment is written (:1262, :1408)
    def build_baked_holes(api):
        scene = api.Scene.empty()
        chunk = scene.add_chunk()
        s0 = chunk.add_shader(Program([["Color", "SetColor"]]), terrain_texture(31, 16, 16, holes=True))
        box = api.Batch3D.from_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
        box.source(B.PixelSource.Pixel((255, 255, 255, 255))).repeat_mode(B.REPEAT_REPEAT_XY).shader(s0)
        chunk.add_batch3d(box)
        scene.add_d3_static(api.Batch3D.from_box(-1.5, -1.5, -2.2, 3.0, 3.0, 0.2).with_computed_normals().source(B.PixelSource.Pixel((20, 200, 40, 255))))
        cam = api.D3OrbitCamera.new()
        cam.set_parameter_f32("distance", 2.6)
        cam.azimuth = 1.3
        cam.elevation = 0.3

        def setup():
            v, p = cam.matrices(float(W), float(H))
            return api.Rasterizer.setup(None, v, p).ambient((0.8, 0.8, 0.8, 1.0))

        return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, "baked-holes")

    got = compare(oracle, product, build_baked_holes)
    green = (got[..., 1] > got[..., 0] + 60) & (got[..., 1] > got[..., 2] + 60)
    centre = green[H // 2 - 15:H // 2 + 15, W // 2 - 15:W // 2 + 15]
    assert 0.1 < centre.mean() < 0.9


def test_chunk_program_in_the_opacity_pass(oracle, product):
    compare(oracle, product, lambda api: chunk_shader_scene(api, baked=True, opacity_list=True), tol=TOLERANCE)


# ---- surface_id across chunks (src/rasterizer.rs:314-357, :1044-1048, :1682) ---------------------------------------------
def two_window_scene(api, swap_chunk_order):
    """Two chunks, each a wall with a 'window': an opacity-pass pane with profile id N in front of an opaque wall with the
    same profile id (so the wall is skipped where the pane was drawn and the green backdrop shows through).  Seen in a
    row, the panes overlap on screen.  A chunk's walls are tested against the opacity winner of the chunks processed SO
    FAR, so which wall shows a hole depends on the chunk order -- the final winner alone gets it wrong."""
    scene = api.Scene.empty()

    def wall(z, profile, colour):
        pane = api.Batch3D.from_box(-0.6, -0.6, z, 1.2, 1.2, 0.02).with_computed_normals().source(B.PixelSource.Pixel((90, 120, 250, 120)))
        pane.profile_id(profile)
        w = api.Batch3D.from_box(-1.5, -1.0, z - 0.3, 3.0, 2.0, 0.05).with_computed_normals().source(B.PixelSource.Pixel(colour + (255,)))
        w.profile_id(profile)
        return pane, w

    near = wall(1.0, 7, (200, 60, 60))
    far = wall(-1.0, 9, (60, 60, 200))
    for pane, w in ([far, near] if swap_chunk_order else [near, far]):
        chunk = scene.add_chunk()
        chunk.add_batch3d_opacity(pane)
        chunk.add_batch3d(w)
    scene.add_d3_static(api.Batch3D.from_box(-3.0, -3.0, -3.0, 6.0, 6.0, 0.1).with_computed_normals().source(B.PixelSource.Pixel((30, 220, 60, 255))))
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 4.5)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.0

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, "two-windows")


@pytest.mark.parametrize("swap", [False, True])
def test_surface_id_follows_the_chunk_order(oracle, product, swap):
    got = compare(oracle, product, lambda api: two_window_scene(api, swap))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) >= 4


def test_surface_id_order_matters_in_this_scene(oracle):
    a = scenes.render(two_window_scene(oracle, False))
    b = scenes.render(two_window_scene(oracle, True))
    assert (a != b).any(axis=2).mean() > 0.01, "the scene is meant to expose the order dependence"


def nested_windows_scene(api, k):
    """k windows seen in a row, each in its own chunk, submitted FAR -> NEAR: every pane is nearer than all panes submitted before
    it, so every one of them is a prefix minimum of its pixels (the "staircase" of rxr_kernels.hip front_insert has k steps).  Each
    chunk's wall carries its pane's profile id and is tested against the opacity winner of the chunks processed so far
    (rasterizer.rs:314-357, :1044-1048)."""
    scene = api.Scene.empty()
    for i in range(k):
        z = -1.5 + 3.0 * i / max(1, k - 1)  # far ... near (the camera looks down -z from z = +4.5)
        pane = api.Batch3D.from_box(-0.6, -0.6, z, 1.2, 1.2, 0.02).with_computed_normals().source(B.PixelSource.Pixel((90, 120, 250, 120))).profile_id(10 + i)
        wall = api.Batch3D.from_box(-1.5, -1.0, z - 0.2, 3.0, 2.0, 0.05).with_computed_normals().source(B.PixelSource.Pixel((40 + 50 * i, 200 - 40 * i, 60, 255)))
        wall.profile_id(10 + i)
        chunk = scene.add_chunk()
        chunk.add_batch3d_opacity(pane)
        chunk.add_batch3d(wall)
    scene.add_d3_static(api.Batch3D.from_box(-3.0, -3.0, -3.0, 6.0, 6.0, 0.1).with_computed_normals().source(B.PixelSource.Pixel((30, 220, 60, 255))))
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 4.5)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.0

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, f"{k}-nested-windows")


def test_three_nested_opacity_batches_are_exact(oracle, product):
    compare(oracle, product, lambda api: nested_windows_scene(api, 3))


def test_four_nested_opacity_batches_are_refused_loudly(oracle, product):
    """the device keeps three prefix minima per pixel, one per opacity group (here every pane is a group of its own: each chunk's
    profiled wall separates it from the next pane); a fourth used to be dropped silently.  Now the
    frame fails with RXR_ERR_UNSUPPORTED, so that a caller can take the CPU path for it."""
    scenes.render(nested_windows_scene(oracle, 4))  # the reference algorithm has no such limit
    with pytest.raises(B.RasterizeError) as e:
        scenes.render(nested_windows_scene(product, 4))
    assert e.value.code == B.RXR_ERR_UNSUPPORTED and "opacity batches nest" in str(e.value)
    # the context is usable afterwards
    compare(oracle, product, lambda api: nested_windows_scene(api, 2))


def panes_of_one_chunk_scene(api, k, second_chunk):
    """k panes of ONE chunk submitted far -> near (k prefix minima per pixel, all from one opacity GROUP: no profiled opaque batch
    lies between them in submission order), then that chunk's walls, one per pane and with the pane's profile id: what the walls
    see is the nearest pane only.  `second_chunk` adds a chunk with its own pane + wall behind them (a second group)."""
    scene = api.Scene.empty()
    chunk = scene.add_chunk()
    for i in range(k):
        z = -1.5 + 3.0 * i / max(1, k - 1)
        off = 0.12 * (i % 3) - 0.12  # the panes overlap only partly, so that some pixels see fewer of them
        chunk.add_batch3d_opacity(api.Batch3D.from_box(-0.6 + off, -0.6 - off, z, 1.2, 1.2, 0.02).with_computed_normals()
                                  .source(B.PixelSource.Pixel((90, 120 + 10 * i, 250 - 20 * i, 120))).profile_id(10 + i))
    for i in range(k):
        z = -1.5 + 3.0 * i / max(1, k - 1)
        chunk.add_batch3d(api.Batch3D.from_box(-1.5, -1.0, z - 0.2, 3.0, 2.0, 0.05).with_computed_normals()
                          .source(B.PixelSource.Pixel((40 + 30 * i, 200 - 25 * i, 60, 255))).profile_id(10 + i))
    if second_chunk:
        c2 = scene.add_chunk()
        c2.add_batch3d_opacity(api.Batch3D.from_box(-0.3, -0.3, 1.9, 0.9, 0.9, 0.02).with_computed_normals().source(B.PixelSource.Pixel((250, 90, 90, 100))).profile_id(99))
        c2.add_batch3d(api.Batch3D.from_box(-1.0, -0.8, 1.7, 2.0, 1.6, 0.05).with_computed_normals().source(B.PixelSource.Pixel((200, 200, 40, 255))).profile_id(99))
    scene.add_d3_static(api.Batch3D.from_box(-3.0, -3.0, -3.0, 6.0, 6.0, 0.1).with_computed_normals().source(B.PixelSource.Pixel((30, 220, 60, 255))))
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 4.5)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.0

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, f"{k}-panes-one-chunk")


@pytest.mark.parametrize("k,second_chunk", [(4, False), (6, False), (6, True), (9, True)])
def test_many_nested_panes_of_one_group_are_exact(oracle, product, k, second_chunk):
    """the staircase keeps one entry per opacity GROUP (run of opacity batches without a profiled opaque batch between them): any
    number of nested panes of one chunk is exact; before, the fourth was refused"""
    got = compare(oracle, product, lambda api: panes_of_one_chunk_scene(api, k, second_chunk))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) >= 4


def wall_before_the_pane_scene(api, pane_first):
    """ONE opacity pane (profile 7), in the second chunk, in front of a wall with the same profile id that lives in the FIRST
    chunk: the wall is drawn before any opacity batch has run, so it must not be skipped behind the pane (surface_id is still
    None, rasterizer.rs:314-357, :1044-1048).  With the chunks the other way round it is."""
    scene = api.Scene.empty()
    pane = api.Batch3D.from_box(-0.6, -0.6, 1.0, 1.2, 1.2, 0.02).with_computed_normals().source(B.PixelSource.Pixel((90, 120, 250, 120))).profile_id(7)
    wall = api.Batch3D.from_box(-1.5, -1.0, 0.0, 3.0, 2.0, 0.05).with_computed_normals().source(B.PixelSource.Pixel((200, 60, 60, 255))).profile_id(7)
    order = [("pane", pane), ("wall", wall)] if pane_first else [("wall", wall), ("pane", pane)]
    for kind, batch in order:
        chunk = scene.add_chunk()
        (chunk.add_batch3d_opacity if kind == "pane" else chunk.add_batch3d)(batch)
    scene.add_d3_static(api.Batch3D.from_box(-3.0, -3.0, -3.0, 6.0, 6.0, 0.1).with_computed_normals().source(B.PixelSource.Pixel((30, 220, 60, 255))))
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 4.5)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.0

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, api.Assets.default(), setup, W, H, 40, "wall-before-pane")


@pytest.mark.parametrize("pane_first", [False, True])
def test_opaque_batch_submitted_before_the_only_opacity_batch(oracle, product, pane_first):
    """found by tools/fuzz_sweep.py (seed 1043): with opacity batches in a single chunk the frame used to run the kernel that
    tests every opaque candidate against the FINAL opacity winner"""
    got = compare(oracle, product, lambda api: wall_before_the_pane_scene(api, pane_first))
    centre = got[H // 2, W // 2]
    if pane_first:
        assert centre[1] > centre[0], "behind the pane the wall is skipped: the green backdrop shows through"
    else:
        assert centre[0] > centre[1], "the wall was drawn before the pane existed: it stays"


@pytest.mark.parametrize("brush", [((4.0, 0.0, 4.0), 2.5, 0.5), ((1.0, 0.0, 7.0), 6.0, 0.0), ((9.5, 0.0, -1.0), 3.0, 5.0), ((4.0, 0.3, 4.0), 0.0, 1.0)])
def test_brush_preview_on_terrain_texels_and_on_missed_pixels(oracle, product, brush):
    """Rasterizer.brush_preview (rasterizer.rs:13-17): a white disc blended into the terrain texels of chunk batches (:1192-1213,
    opacity pass :1601-1622) and over the pixels no 3D fragment reached (:435-458: screen_ray through the pixel corner, plane
    y = 0).  Without the blue floor under the terrain the frame has missed pixels around it."""
    got = compare(oracle, product, lambda api: terrain_scene(api, lights=False, brush=brush, under=False, opacity_terrain=True))
    plain = scenes.render(terrain_scene(product, lights=False, under=False, opacity_terrain=True))
    if brush[1] > 0.0:
        assert (got != plain).any(), "the brush left no trace"


def test_brush_preview_with_lights_and_holes(oracle, product):
    compare(oracle, product, lambda api: terrain_scene(api, holes=True, brush=((3.0, 0.0, 5.0), 3.0, 0.4)), tol=TOLERANCE)


NAN_, INF_ = float("nan"), float("inf")


@pytest.mark.parametrize("brush", [((NAN_, 0.0, 4.0), 2.5, 0.5), ((4.0, INF_, 4.0), 2.5, 0.5), ((4.0, 0.0, 4.0), NAN_, 0.5), ((4.0, 0.0, 4.0), INF_, 0.5),
                                   ((4.0, 0.0, 4.0), -1.0, 0.5), ((4.0, 0.0, 4.0), 2.5, NAN_), ((4.0, 0.0, 4.0), 2.5, INF_), ((4.0, 0.0, 4.0), 2.5, -3.0),
                                   ((4.0, 0.0, 4.0), 1e-40, 1e-40), ((3.0e38, 0.0, -3.0e38), 3.0e38, 0.001)])
def test_brush_preview_with_poisoned_parameters(oracle, product, brush):
    """the brush disc with NaN / inf / negative / denormal position, radius and falloff (rasterizer.rs:1193-1212, :435-458): on terrain
    texels and on missed pixels, bit for bit"""
    compare(oracle, product, lambda api: terrain_scene(api, lights=False, brush=brush, under=False, opacity_terrain=True))


def binned_grid_with_panes_scene(api, profiled, second_chunk):
    """A BINNED frame (a 20 x 20 box lattice: 4800 triangles, row mode) with an opacity pass: translucent panes in a chunk's opacity list
    in front of the lattice.  `profiled`: every third lattice batch carries the first pane's profile id (its fragments are skipped where
    that pane is the opacity layer, rasterizer.rs:1044-1048) -- rounds of candidates with such a batch walk, the others take row mode;
    `second_chunk`: a second chunk with a pane of its own (the exact prefix order: feature level 1)."""
    cfg = scenes.box_grid_scene(api, n=20, width=640, height=360, profile_every=3 if profiled else 0, profile_id=10)
    scene = cfg.scene
    chunk = scene.add_chunk()
    chunk.add_batch3d_opacity(api.Batch3D.from_box(1.2, 0.2, 1.0, 1.6, 1.2, 0.02).with_computed_normals()
                              .source(B.PixelSource.Pixel((90, 160, 250, 120))).profile_id(10))
    chunk.add_batch3d_opacity(api.Batch3D.from_box(2.0, 0.0, 2.4, 1.0, 1.5, 0.02).with_computed_normals()
                              .source(B.PixelSource.Pixel((250, 120, 60, 90))).profile_id(11))
    if second_chunk:
        c2 = scene.add_chunk()
        c2.add_batch3d_opacity(api.Batch3D.from_box(0.4, 0.1, 3.0, 1.4, 1.0, 0.02).with_computed_normals().source(B.PixelSource.Pixel((60, 250, 90, 140))).profile_id(10))
        c2.add_batch3d(api.Batch3D.from_box(0.2, 0.0, 3.4, 2.0, 0.8, 0.05).with_computed_normals().source(B.PixelSource.Pixel((200, 200, 40, 255))).profile_id(10))
    return cfg


@pytest.mark.parametrize("profiled,second_chunk", [(False, False), (True, False), (True, True)])
def test_opacity_pass_over_a_binned_frame_in_row_mode(oracle, product, profiled, second_chunk):
    """since the end of round 4 a frame with an opacity pass keeps row mode for its rounds of candidates without a profile id (before, one
    translucent pane made a binned frame's raster kernel 3.6 times slower) and the opacity pass stages candidates of the opacity lists
    only: the frame must equal the oracle's byte for byte -- the panes blended over the lattice, the profiled batches cut out behind
    the pane with their id"""
    got = compare(oracle, product, lambda api: binned_grid_with_panes_scene(api, profiled, second_chunk))
    plain = scenes.render(scenes.box_grid_scene(product, n=20, width=640, height=360))
    assert (got != plain).any(axis=2).mean() > 0.02, "the panes change nothing: the test tests nothing"
