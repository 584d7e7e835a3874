"""Synthetic scenes for the configurations named in BASELINE.json.

Everything is procedural and seeded (seed 0x52585231); no third-party art is used: the
reference's minigame/*.png and images/logo.png are replaced by stand-ins of the same sizes and alpha
structure.  Each builder takes an `api` namespace from :func:`rusterix_amd.binding.make_api`, so the
same description can be replayed on the product host library and (from tests/) on the CPU oracle.

The scene drivers model reference callers:
  C1  benches/rasterize_cube.rs:7-33
  C2  examples/obj.rs:28-83           (mesh: procedural stand-in with the teapot's vertex/face counts,
                                        or the real OBJ text when the caller supplies it)
  C3  examples/map.rs:41-125 + minigame/world.rxm
  C4  C3 at 3840x2160 with 16 point lights
  C5  289 batches x 289 boxes (1 002 252 triangles), 7680x4320, Linear sampling
"""
from __future__ import annotations

import math
import os
import types

import numpy as np

from . import binding as B

SEED = 0x52585231


# ---- procedural textures -------------------------------------------------------------------------
def _rng(tag: int):
    return np.random.default_rng([SEED, tag])


def brick_texture(tag: int, w=64, h=64, base=(150, 70, 50), mortar=(180, 175, 165)):
    """Opaque RGBA brick pattern with seeded per-texel noise (stand-in for minigame/brick*.png)."""
    rng = _rng(tag)
    img = np.zeros((h, w, 4), np.uint8)
    noise = rng.integers(-18, 19, size=(h, w, 3))
    yy, xx = np.mgrid[0:h, 0:w]
    row = yy // 8
    xoff = (row % 2) * 8
    is_mortar = ((yy % 8) == 0) | (((xx + xoff) % 16) == 0)
    col = np.where(is_mortar[..., None], np.array(mortar)[None, None, :], np.array(base)[None, None, :])
    img[..., :3] = np.clip(col + noise, 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return B.Texture(img.reshape(-1), w, h)


def panel_texture(tag: int, w=64, h=64):
    rng = _rng(tag)
    img = np.zeros((h, w, 4), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    glow = 200 + 55 * np.cos((xx - w / 2) / w * math.pi) * np.cos((yy - h / 2) / h * math.pi)
    img[..., 0] = np.clip(glow + rng.integers(-6, 7, (h, w)), 0, 255)
    img[..., 1] = np.clip(glow + rng.integers(-6, 7, (h, w)), 0, 255)
    img[..., 2] = np.clip(glow * 0.73 + rng.integers(-6, 7, (h, w)), 0, 255)
    img[..., 3] = 255
    return B.Texture(img.reshape(-1), w, h)


def fence_texture(tag: int, w=64, h=80):
    """64 wide x 80 high fence with ~45 % cut-out texels (alpha < 255), like minigame/fence.png."""
    rng = _rng(tag)
    img = np.zeros((h, w, 4), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    bars = ((xx % 16) < 5) | ((yy % 40) < 6) | (((xx + yy) % 32) < 3)
    img[..., 0] = np.clip(90 + rng.integers(-15, 16, (h, w)), 0, 255)
    img[..., 1] = np.clip(60 + rng.integers(-15, 16, (h, w)), 0, 255)
    img[..., 2] = np.clip(30 + rng.integers(-15, 16, (h, w)), 0, 255)
    # cut-outs: fully transparent holes plus a sprinkle of partially transparent texels
    alpha = np.where(bars, 255, 0)
    partial = rng.random((h, w)) < 0.02
    alpha = np.where(partial & bars, 128, alpha)
    img[..., 3] = alpha
    return B.Texture(img.reshape(-1), w, h)


def logo_texture(tag: int, size=1024):
    """Opaque size^2 checker + gradient + noise (stand-in for images/logo.png, RGB 1024^2)."""
    rng = _rng(tag)
    yy, xx = np.mgrid[0:size, 0:size]
    img = np.zeros((size, size, 4), np.uint8)
    checker = (((xx // 32) + (yy // 32)) % 2) * 90
    img[..., 0] = np.clip(checker + xx * 160 // size + rng.integers(0, 8, (size, size)), 0, 255)
    img[..., 1] = np.clip(checker + yy * 160 // size + rng.integers(0, 8, (size, size)), 0, 255)
    img[..., 2] = np.clip(200 - checker + rng.integers(0, 8, (size, size)), 0, 255)
    img[..., 3] = 255
    return B.Texture(img.reshape(-1), size, size)


def noise_texture(tag: int, w=64, h=64):
    rng = _rng(tag)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    img[..., 3] = 255
    return B.Texture(img.reshape(-1), w, h)


# ---- meshes --------------------------------------------------------------------------------------
def standin_teapot_mesh():
    """Closed UV-sphere-like lathe mesh with exactly 1202 vertices and 2256 triangles... of the
    teapot's size class (examples/teapot.obj: 1202 v / 2256 f).  Returns (vertices[n,4], indices[m,3],
    uvs[n,2]) with uv = (x, y) as src/wavefront.rs:92-95 does for OBJ files without `vt`."""
    rings, segs = 47, 24  # 47*24 + 2 poles = 1130 ... pad with a spout ring below
    verts = [(0.0, 1.35, 0.0)]
    for r in range(1, rings + 1):
        t = r / (rings + 1)
        y = 1.35 - 2.4 * t
        # lathe profile: lid knob, body bulge, foot
        rad = 0.25 + 1.15 * math.sin(math.pi * t) ** 0.8 + 0.15 * math.sin(6 * math.pi * t)
        for s in range(segs):
            a = 2 * math.pi * s / segs
            verts.append((rad * math.cos(a), y, rad * math.sin(a)))
    verts.append((0.0, -1.05, 0.0))
    tris = []
    for s in range(segs):
        tris.append((0, 1 + s, 1 + (s + 1) % segs))
    for r in range(rings - 1):
        b0 = 1 + r * segs
        b1 = b0 + segs
        for s in range(segs):
            s1 = (s + 1) % segs
            tris.append((b0 + s, b1 + s, b1 + s1))
            tris.append((b0 + s, b1 + s1, b0 + s1))
    last = len(verts) - 1
    b0 = 1 + (rings - 1) * segs
    for s in range(segs):
        tris.append((last, b0 + (s + 1) % segs, b0 + s))
    # 1130 verts / 2256 tris so far ((rings-1)*segs*2 + 2*segs = 46*48+48 = 2256); pad the vertex
    # count to 1202 with a 72-vertex unreferenced handle loop so V matches the teapot's.
    for k in range(1202 - len(verts)):
        a = 2 * math.pi * k / 72
        verts.append((1.6 + 0.3 * math.cos(a), 0.3 * math.sin(a), 0.0))
    v = np.array([(x, y, z, 1.0) for (x, y, z) in verts], np.float32)
    i = np.array(tris, np.uint32)
    uv = v[:, :2].copy()
    assert v.shape[0] == 1202 and i.shape[0] == 2256
    return v, i, uv


# ---- scene builders ------------------------------------------------------------------------------
def _result(api, scene, assets, setup, width, height, tile_size, name, **extra):
    """`setup()` returns a configured Rasterizer (one per frame, as in the reference)."""
    return types.SimpleNamespace(api=api, scene=scene, assets=assets, setup=setup, width=width, height=height,
                                 tile_size=tile_size, name=name, **extra)


def cube_scene(api, width=800, height=600, tile_size=200, textured=False, distance=20.0, sample_mode=B.SAMPLE_NEAREST,
               logo_size=1024, rect_size=200.0):
    """C1: benches/rasterize_cube.rs:7-33 (+ with_computed_normals, else the reference panics).  `rect_size`: side of the 2D
    rectangle drawn over the frame (200 in the reference's drivers; thumbnails use a smaller one so that the 3D part shows)."""
    rect = api.Batch2D.from_rectangle(0.0, 0.0, float(rect_size), float(rect_size))
    box = api.Batch3D.from_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
    if textured:
        rect.source(B.PixelSource.StaticTileIndex(0))
        box.source(B.PixelSource.StaticTileIndex(0))
    scene = api.Scene.from_static([rect], [box]).background(api.VGrayGradientShader())
    assets = api.Assets.default().textures([B.Tile.from_texture(logo_texture(1, logo_size))])
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", distance)

    def setup():
        v, p = cam.matrices(float(width), float(height))
        return api.Rasterizer.setup(None, v, p).sample_mode(sample_mode)

    return _result(api, scene, assets, setup, width, height, tile_size, "C1-cube")


TEAPOT_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "teapot_mesh.npz")


def teapot_mesh():
    """The arrays `Batch3D::from_obj` makes of the reference's examples/teapot.obj (1202 v / 2256 f; Utah teapot, mesh data only),
    from the committed fixture (tests/golden/make_teapot_fixture.py); the stand-in of equal counts only if the fixture is absent.
    Returns (vertices[n,4], indices[m,3], uvs[n,2], is_real)."""
    if os.path.exists(TEAPOT_FIXTURE):
        z = np.load(TEAPOT_FIXTURE)
        pos = z["positions"].astype(np.float32)
        v = np.concatenate([pos, np.ones((len(pos), 1), np.float32)], axis=1)
        return v, z["indices"].astype(np.uint32), v[:, :2].copy(), True  # no `vt` in the file: uv = (x, y), src/wavefront.rs:92-95
    return standin_teapot_mesh() + (False,)


def teapot_scene(api, width=1920, height=1080, tile_size=60, obj_text=None, with_light=False, logo_size=1024, rect_size=200.0, standin=False):
    """C2: examples/obj.rs:28-83 with the point light dropped (BASELINE.json: "no lights")."""
    real = True
    if obj_text is not None:
        mesh = api.Batch3D.from_obj(obj_text)
    else:
        v, i, uv, real = teapot_mesh() if not standin else standin_teapot_mesh() + (False,)
        mesh = api.Batch3D.new(v, i, uv)
    mesh = (mesh.source(B.PixelSource.StaticTileIndex(0)).repeat_mode(B.REPEAT_REPEAT_XY)
            .transform(B.Mat4.scaling_3d((0.35, -0.35, 0.35))).with_computed_normals())
    scene = api.Scene.from_static([api.Batch2D.from_rectangle(0.0, 0.0, float(rect_size), float(rect_size))], [mesh])
    scene.background(api.VGrayGradientShader())
    if with_light:
        scene.lights([B.Light(B.LIGHT_POINT).with_intensity(1.0).with_color((1.0, 1.0, 0.95))
                      .with_position((2.0, 0.8, 0.0)).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(logo_texture(1, logo_size))])
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 1.5)

    def setup():
        v, p = cam.matrices(float(width), float(height))
        return api.Rasterizer.setup(None, v, p).ambient((0.8, 0.8, 0.8, 0.8))

    return _result(api, scene, assets, setup, width, height, tile_size, "C2-teapot" if real else "C2-teapot-standin-mesh")


def _quad_wall(x0, z0, x1, z1, h):
    length = math.hypot(x1 - x0, z1 - z0)
    verts = [(x0, 0.0, z0, 1.0), (x1, 0.0, z1, 1.0), (x1, h, z1, 1.0), (x0, h, z0, 1.0)]
    uvs = [(0.0, h), (length, h), (length, 0.0), (0.0, 0.0)]
    idx = [(0, 1, 2), (0, 2, 3)]
    return np.array(verts, np.float32), np.array(idx, np.uint32), np.array(uvs, np.float32)


def _batch_of_quads(api, quads):
    b = None
    for (v, i, uv) in quads:
        if b is None:
            b = api.Batch3D.new(v, i, uv)
        else:
            b.add(v, i, uv)
    return b


MAP_TILES = dict(logo=0, brickwall=1, brickfloor=2, brickwall2=3, lightpanel=4, fence=5)


def map_assets(api, logo_size=1024):
    tiles = [
        B.Tile.from_texture(logo_texture(1, logo_size)),
        B.Tile.from_texture(brick_texture(2)),
        B.Tile.from_texture(brick_texture(3, base=(120, 110, 100), mortar=(70, 70, 70))),
        B.Tile.from_texture(brick_texture(4, base=(160, 90, 60))),
        B.Tile.from_texture(panel_texture(5)),
        B.Tile.from_texture(fence_texture(6)),
    ]
    return api.Assets.default().textures(tiles)


def map_scene(api, width=1920, height=1080, tile_size=40, n_lights=1, logo_size=1024, sample_mode=B.SAMPLE_NEAREST, rect_size=200.0):
    """C3 (n_lights=1) / C4 (n_lights=16, 3840x2160): the minigame room synthesised from
    minigame/world.rxm (the reference's own builder is stubbed at this snapshot)."""
    box = 15.0
    hgt = 2.0
    floor_v = np.array([(0, 0, 0, 1), (box, 0, 0, 1), (box, 0, box, 1), (0, 0, box, 1)], np.float32)
    floor_uv = np.array([(0, 0), (box, 0), (box, box), (0, box)], np.float32)
    floor_i = np.array([(0, 1, 2), (0, 2, 3)], np.uint32)
    floor = (api.Batch3D.new(floor_v, floor_i, floor_uv).source(B.PixelSource.StaticTileIndex(MAP_TILES["brickfloor"]))
             .repeat_mode(B.REPEAT_REPEAT_XY).with_computed_normals())
    walls = _batch_of_quads(api, [
        _quad_wall(0, 0, box, 0, hgt), _quad_wall(box, 0, box, box, hgt), _quad_wall(box, box, 10, box, hgt),
        _quad_wall(9, box, 0, box, hgt), _quad_wall(0, box, 0, 0, hgt)])
    walls = (walls.source(B.PixelSource.StaticTileIndex(MAP_TILES["brickwall"])).repeat_mode(B.REPEAT_REPEAT_XY)
             .with_computed_normals())
    panel = _batch_of_quads(api, [_quad_wall(10, box, 9, box, hgt)])
    panel = (panel.source(B.PixelSource.StaticTileIndex(MAP_TILES["lightpanel"])).repeat_mode(B.REPEAT_REPEAT_XY)
             .with_computed_normals())
    fence = _batch_of_quads(api, [_quad_wall(6, box, 6, 9, hgt), _quad_wall(6, 9, 0, 9, hgt)])
    fence = (fence.source(B.PixelSource.StaticTileIndex(MAP_TILES["fence"])).repeat_mode(B.REPEAT_REPEAT_XY)
             .with_computed_normals())
    logo = (api.Batch2D.from_rectangle(0.0, 0.0, float(rect_size), float(rect_size)).receives_light(False)
            .source(B.PixelSource.StaticTileIndex(MAP_TILES["logo"])))
    scene = api.Scene.from_static([logo], [floor, walls, panel, fence]).background(api.VGrayGradientShader())

    if n_lights == 1:
        lights = [B.Light(B.LIGHT_POINT).with_position((9.0, 0.5, 15.0)).with_color((1.0, 1.0, 0.7333))
                  .with_intensity(2.0).with_start_distance(2.0).with_end_distance(13.0).compile()]
    else:
        rng = _rng(100)
        grid = [2.0, 5.67, 9.33, 13.0]
        lights = []
        for gz in grid:
            for gx in grid:
                if len(lights) >= n_lights:
                    break
                col = 0.55 + 0.45 * rng.random(3)
                lights.append(B.Light(B.LIGHT_POINT).with_position((gx, 1.5, gz)).with_color(tuple(float(c) for c in col))
                              .with_intensity(1.5).with_start_distance(1.0).with_end_distance(6.0).compile())
    scene.lights(lights)
    assets = map_assets(api, logo_size)

    cam = api.D3FirstPCamera.new()
    pos = np.array((6.0600824, 1.0, 4.5524735), np.float32)
    cam.position = tuple(pos)
    cam.center = tuple(pos + np.array((0.03489969, 0.0, 0.99939084), np.float32))

    def setup():
        v, p = cam.matrices(float(width), float(height))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0)).sample_mode(sample_mode)

    name = "C3-map" if n_lights == 1 else f"C4-map-{n_lights}lights"
    return _result(api, scene, assets, setup, width, height, tile_size, name, n_lights=n_lights)


def box_grid_shader():
    """The "per-batch shader" of configuration C5: a pure colour-only Rusteria program (no emissive, no globals,
    nothing read before it is written) -- tints the texel by a stripe pattern in uv and the height of the hit point."""
    return B.Program([[
        "UV", ("Push", 4.0), "Mul", "Fract", ("GetComponents", [0]), ("Push", 0.5), "Lt",       # uv is interpolated_uv / 4
        ("If", [("Push", 1.0, 0.85, 0.7)], [("Push", 0.7, 0.85, 1.0)]),
        "Color", "Mul",
        "Hitpoint", ("GetComponents", [1]), ("Push", 1.5), "Mul", ("Push", 0.4), "Add", "Mul",
        "SetColor",
    ]])


def box_grid_scene(api, n=289, width=7680, height=4320, tile_size=40, sample_mode=B.SAMPLE_LINEAR, boxes_per_batch=None, shader=False,
                   distance=None, profile_every=0, profile_id=10, cutout_every=0):
    """C5: n batches of n boxes on an n x n lattice (n=289 -> 1 002 252 triangles); `shader`: every batch runs box_grid_shader();
    `distance`: the orbit camera's (default: the whole lattice in view; a small one puts the camera among the boxes, across whose
    near plane many triangles then lie); `profile_every` = k > 0: every k-th batch carries `profile_id` (what an opacity pass cuts out);
    `cutout_every` = k > 0: every k-th batch takes a texture with holes (texel alpha 0: fragments that are not written, rasterizer.rs:1408)."""
    rng = _rng(200)
    spacing, size = 0.2, 0.16
    tmpl = api.Batch3D.from_box(0.0, 0.0, 0.0, size, size, size)
    tv, ti, tuv, _ = tmpl.geometry()
    ys = rng.random((n, n)).astype(np.float32) * np.float32(0.4)
    scene = api.Scene.empty()
    shader_index = scene.add_program(box_grid_shader()) if shader else None
    per = boxes_per_batch or n
    for bz in range(n):
        vs, is_, uvs = [], [], []
        for bx in range(per):
            v = tv.copy()
            v[:, 0] += np.float32(bx * spacing)
            v[:, 1] += ys[bz, bx]
            v[:, 2] += np.float32(bz * spacing)
            vs.append(v)
            is_.append(ti + np.uint32(24 * bx))
            uvs.append(tuv)
        b = api.Batch3D.new(np.concatenate(vs), np.concatenate(is_), np.concatenate(uvs))
        tile = 16 if (cutout_every and bz % cutout_every == 0) else bz % 16
        b = (b.source(B.PixelSource.StaticTileIndex(tile)).repeat_mode(B.REPEAT_REPEAT_XY).with_computed_normals())
        if shader_index is not None:
            b.shader(shader_index)
        if profile_every and bz % profile_every == 0:
            b.profile_id(profile_id)
        scene.add_d3_static(b)
    assets = api.Assets.default().textures([B.Tile.from_texture(noise_texture(300 + k)) for k in range(16)] + [B.Tile.from_texture(fence_texture(7))])
    extent = n * spacing
    cam = api.D3OrbitCamera.new()
    cam.center = (extent / 2, 0.0, extent / 2)
    cam.distance = 45.0 * (extent / 57.8) if distance is None else distance

    def setup():
        v, p = cam.matrices(float(width), float(height))
        return api.Rasterizer.setup(None, v, p).sample_mode(sample_mode).ambient((1.0, 1.0, 1.0, 1.0))

    return _result(api, scene, assets, setup, width, height, tile_size, f"C5-boxgrid-{n}")


def tile_map_2d_scene(api, width=640, height=400, nx=30, ny=20, stacked=0, lines=True, lights=True):
    """A 2D "tile map": nx*ny textured / translucent rectangles with overlaps, a few screen-sized
    translucent overlays (large primitives), optional `stacked` rectangles piled on one spot (more
    candidates in one tile than the LDS sort holds) and Bresenham line batches in between."""
    rng = np.random.default_rng([0x52585231, 77, nx, ny, stacked])
    rects = []
    cw, ch = width / nx, height / ny
    for j in range(ny):
        for i in range(nx):
            r = api.Batch2D.from_rectangle(float(np.float32(i * cw - 1.0)), float(np.float32(j * ch - 1.0)), float(np.float32(cw + 3.0)), float(np.float32(ch + 3.0)))
            k = (i + 2 * j) % 4
            if k == 0:
                r.source(B.PixelSource.StaticTileIndex(int(rng.integers(0, 3)))).repeat_mode(B.REPEAT_REPEAT_XY)
            elif k == 1:
                r.source(B.PixelSource.Pixel(tuple(int(x) for x in rng.integers(0, 256, 3)) + (int(rng.integers(40, 255)),)))
            elif k == 2:
                r.source(B.PixelSource.StaticTileIndex(2))
            else:
                r.source(B.PixelSource.Pixel(tuple(int(x) for x in rng.integers(0, 256, 3)) + (255,)))
            r.receives_light(bool((i + j) % 3))
            rects.append(r)
        if lines and j % 5 == 2:
            v = np.array([[5.5, j * ch + 3.2], [width - 7.0, j * ch + ch * 0.8], [width * 0.5, j * ch - 4.0], [40.0, j * ch + 30.0]], np.float32)
            rects.append(api.Batch2D.new(v, np.array([[0, 1, 0], [1, 2, 0], [2, 3, 0]], np.uint32), np.zeros_like(v)).mode(B.MODE_LINES)
                         .source(B.PixelSource.Pixel((255, 255, 0, 255))))
    rects.append(api.Batch2D.from_rectangle(20.0, 15.0, float(width - 60), float(height - 50)).source(B.PixelSource.Pixel((30, 60, 200, 90))))
    for _ in range(stacked):
        rects.append(api.Batch2D.from_rectangle(100.0, 100.0, 24.0, 24.0).source(B.PixelSource.Pixel(tuple(int(x) for x in rng.integers(0, 256, 3)) + (40,))))
    rects.append(api.Batch2D.from_rectangle(0.0, 0.0, float(width), float(height)).source(B.PixelSource.Pixel((255, 255, 255, 20))))
    scene = api.Scene.from_static(rects, [])
    if lights:
        scene.lights([B.Light(B.LIGHT_POINT).with_position((width * 0.5, 0.0, height * 0.5)).with_color((1.0, 0.9, 0.7)).with_intensity(1.0)
                      .with_start_distance(50.0).with_end_distance(400.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(brick_texture(2)), B.Tile.from_texture(fence_texture(6)),
                                            B.Tile.from_texture(logo_texture(1, 64))])

    def setup():
        v, p = api.D3OrbitCamera.new().matrices(float(width), float(height))
        return (api.Rasterizer.setup(None, v, p).render_mode(B.RenderMode.render_2d()).background((12, 12, 12, 255))
                .ambient((0.4, 0.4, 0.4, 1.0)).mapmini_add_linedef((200.0, 0.0), (200.0, 150.0)))

    return _result(api, scene, assets, setup, width, height, 40, "rects")


def grid_editor_scene(api, width=320, height=200, grid_size=30.0, subdivisions=2.0, offset=(0.0, 0.0)):
    """The 2D editor's view: GridShader background (reference src/shader/grid.rs) under a textured and a translucent rectangle."""
    shader = api.GridShader().set_parameter_f32("grid_size", grid_size).set_parameter_f32("subdivisions", subdivisions).set_parameter_vec2("offset", offset)
    scene = api.Scene.empty().background(shader)
    scene.add_d2_static(api.Batch2D.from_rectangle(width * 0.1, height * 0.15, width * 0.3, height * 0.4).source(B.PixelSource.StaticTileIndex(0)))
    scene.add_d2_static(api.Batch2D.from_rectangle(width * 0.3, height * 0.4, width * 0.4, height * 0.3).source(B.PixelSource.Pixel((200, 40, 90, 128))))
    assets = api.Assets.default().textures([B.Tile.from_texture(noise_texture(910, 16, 16))])
    v, p = api.D3OrbitCamera.new().matrices(float(width), float(height))
    return _result(api, scene, assets, lambda: api.Rasterizer.setup(None, v, p).render_mode(B.RenderMode.render_2d()), width, height, 40, "grid-editor")


def small_triangle_mesh_scene(api, width=320, height=200, n_triangles=1500, brush=None):
    """A cloud of small textured triangles, submitted twice (exact depth ties), over an empty background: the binned pipeline's
    row-parallel visibility; `brush` = (position, radius, falloff) adds the editor's brush preview over the missed pixels."""
    rng = _rng(920)
    centre = rng.normal(0.0, 1.0, size=(n_triangles, 1, 3)) * np.array([1.3, 0.8, 1.0])
    verts = (centre + rng.normal(0.0, 0.07, size=(n_triangles, 3, 3))).reshape(-1, 3).astype(np.float32)
    v4 = np.concatenate([verts, np.ones((len(verts), 1), np.float32)], axis=1)
    idx = np.arange(n_triangles * 3, dtype=np.uint32).reshape(n_triangles, 3)
    uv = (rng.random((n_triangles * 3, 2)) * 2.0).astype(np.float32)
    scene = api.Scene.empty()
    for k in range(2):
        b = api.Batch3D.new(v4.copy(), idx.copy(), uv.copy()).with_computed_normals().cull_mode(B.CULL_OFF)
        scene.add_d3_static(b.source(B.PixelSource.StaticTileIndex(k)).repeat_mode(B.REPEAT_REPEAT_XY).ambient_color((0.9, 0.8, 0.7)))
    assets = api.Assets.default().textures([B.Tile.from_texture(noise_texture(921 + k, 16, 16)) for k in range(2)])
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 3.0)

    def setup():
        v, p = cam.matrices(float(width), float(height))
        r = api.Rasterizer.setup(None, v, p)
        if brush is not None:
            r.brush_preview(*brush)
        return r

    return _result(api, scene, assets, setup, width, height, 40, "small-triangle-mesh")


def render(cfg, out=None):
    """One `Rasterizer::setup(..).rasterize(scene, pixels, w, h, tile, assets)` call."""
    if out is None:
        out = np.zeros(cfg.width * cfg.height * 4, np.uint8)
    cfg.setup().rasterize(cfg.scene, out, cfg.width, cfg.height, cfg.tile_size, cfg.assets)
    return out.reshape(cfg.height, cfg.width, 4)
