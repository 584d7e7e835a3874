"""Batch3D::clip_and_project + Edges::new + bounding box on the device.

The device-projected path must be indistinguishable from the host-projected one: the projected
arrays read back from the GPU equal the C++ host mirror's (which the CPU tests pin bit-for-bit to the
oracle), and the rendered frames are byte-identical -- including the lit ones, because both paths feed
the same raster kernels."""
import ctypes as C

import numpy as np
import pytest

import rusterix_amd
from rusterix_amd import binding as B
from rusterix_amd import scenes

pytestmark = pytest.mark.gpu


class RxrEdges(C.Structure):
    _fields_ = [("a", C.c_float * 3), ("b", C.c_float * 3), ("c", C.c_float * 3), ("visible", C.c_uint32)]


@pytest.fixture()
def devproj(product):
    lib = product.lib
    lib.rxh_set_device_projection.argtypes = [C.c_int]
    lib.rxh_context.restype = C.c_void_p
    rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
    rxr.rxr_read_projected_mesh.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(RxrEdges), C.POINTER(C.c_float),
                                            C.c_uint32, C.c_uint32]

    def read_mesh(index, cap_v, cap_t):
        counts = (C.c_uint32 * 2)()
        pv = np.zeros((cap_v, 4), np.float32)
        uv = np.zeros((cap_v, 2), np.float32)
        nr = np.zeros((cap_v, 3), np.float32)
        idx = np.zeros((cap_t, 3), np.uint32)
        ed = (RxrEdges * cap_t)()
        bb = np.zeros(5, np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        rc = rxr.rxr_read_projected_mesh(lib.rxh_context(), index, counts, fp(pv), fp(uv), fp(nr), idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         ed, fp(bb), cap_v, cap_t)
        assert rc == 0
        nv, nt = counts[0], counts[1]
        edges = np.array([[*e.a, *e.b, *e.c, float(e.visible)] for e in ed[:nt]], np.float32).reshape(nt, 10)
        return dict(projected_vertices=pv[:nv], clipped_uvs=uv[:nv], clipped_normals=nr[:nv], clipped_indices=idx[:nt], edges=edges,
                    bounding_box=bb)

    class Ctl:
        def on(self):
            lib.rxh_set_device_projection(1)

        def off(self):
            lib.rxh_set_device_projection(0)

        read = staticmethod(read_mesh)

    ctl = Ctl()
    yield ctl
    ctl.off()


SCENES = [
    ("cube", scenes.cube_scene, dict(width=320, height=200, distance=3.0, textured=True, logo_size=64)),
    ("cube_near_clip", scenes.cube_scene, dict(width=333, height=211, distance=0.7, textured=True, logo_size=64)),
    ("cube_inside", scenes.cube_scene, dict(width=256, height=160, distance=0.3, textured=True, logo_size=64)),
    ("teapot_lit", scenes.teapot_scene, dict(width=480, height=270, logo_size=64, with_light=True)),
    ("map16", scenes.map_scene, dict(width=640, height=360, logo_size=64, n_lights=16)),
    ("box_grid", scenes.box_grid_scene, dict(n=24, width=512, height=288)),
]
# (appended, the slices below keep their meaning) the multi-launch pre-pass WITH clipped triangles: frames in which no triangle crosses
# the near plane leave k_proj_scan / k_clip_emit early, this one must not
SCENES_CLIPPED_LARGE = [("box_grid_inside", scenes.box_grid_scene, dict(n=24, width=512, height=288, distance=0.6))]


def _panes(api, **kw):
    """six nested panes of one chunk + a second chunk (tests/test_gpu_chunks.py): chunk lists, opacity groups and surface_id with
    device-projected meshes"""
    from tests.test_gpu_chunks import panes_of_one_chunk_scene

    return panes_of_one_chunk_scene(api, **kw)


SCENES.append(("nested_panes", _panes, dict(k=6, second_chunk=True)))


@pytest.mark.parametrize("name,builder,kw", SCENES + SCENES_CLIPPED_LARGE, ids=[s[0] for s in SCENES + SCENES_CLIPPED_LARGE])
def test_frames_identical_to_host_projection(product, devproj, name, builder, kw):
    devproj.off()
    want = scenes.render(builder(product, **kw)).copy()
    devproj.on()
    cfg = builder(product, **kw)
    got = scenes.render(cfg).copy()
    assert np.array_equal(got, want), f"{name}: {(got != want).any(axis=2).sum()} pixels differ"
    again = scenes.render(cfg)          # second frame of the same scene: meshes stay registered, only matrices travel
    assert np.array_equal(again, want)


_STATIC_ONLY = SCENES[:4] + SCENES[5:6] + SCENES_CLIPPED_LARGE   # (scenes of static batches only)


@pytest.mark.parametrize("name,builder,kw", _STATIC_ONLY, ids=[s[0] for s in _STATIC_ONLY])
def test_projected_arrays_match_host_mirror(product, devproj, name, builder, kw):
    # host mirror projection (bit-identical to the oracle: tests/test_host_and_abi.py)
    devproj.off()
    ref_cfg = builder(product, **kw)
    ref_cfg.setup().project(ref_cfg.scene, ref_cfg.width, ref_cfg.height)
    devproj.on()
    cfg = builder(product, **kw)
    scenes.render(cfg)
    i = 0
    while True:
        try:
            ref = ref_cfg.scene.projected_batch3d(B.LIST_STATIC, i)
        except IndexError:
            break
        nv, nt = ref["projected_vertices"].shape[0], ref["clipped_indices"].shape[0]
        if ref["bounding_box"][0] == 0.0:      # frustum-rejected batch: everything cleared on both sides
            got = devproj.read(i, 8, 8)
            assert got["bounding_box"][0] == 0.0 and got["projected_vertices"].shape[0] == 0
        else:
            got = devproj.read(i, nv + 16, nt + 16)
            for key in ("projected_vertices", "clipped_uvs", "clipped_normals", "clipped_indices", "edges", "bounding_box"):
                assert got[key].shape == ref[key].shape, (name, i, key, got[key].shape, ref[key].shape)
                assert got[key].tobytes() == ref[key].tobytes(), (name, i, key)
        i += 1
    assert i > 0


@pytest.mark.parametrize("cull", [B.CULL_OFF, B.CULL_FRONT, B.CULL_BACK])
def test_cull_modes_and_moving_transform(product, devproj, cull):
    def build(api, angle):
        cam = api.D3OrbitCamera.new()
        cam.set_parameter_f32("distance", 1.1)
        v, p = cam.matrices(256.0, 192.0)
        c, s = np.float32(np.cos(angle)), np.float32(np.sin(angle))
        rot = B.Mat4.from_rows([[c, 0, s, 0.1], [0, 1, 0, -0.05], [-s, 0, c, 0.2], [0, 0, 0, 1]])
        box = (api.Batch3D.from_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(cull).source(B.PixelSource.StaticTileIndex(0))
               .transform(rot).with_computed_normals())
        far = api.Batch3D.from_box(40.0, 40.0, 40.0, 1.0, 1.0, 1.0).source(B.PixelSource.Pixel((9, 9, 9, 255))).with_computed_normals()  # frustum-rejected
        scene = api.Scene.from_static([], [box, far])
        assets = api.Assets.default().textures([B.Tile.from_texture(scenes.brick_texture(2))])
        return scenes._result(api, scene, assets, lambda: api.Rasterizer.setup(None, v, p).ambient((0.9, 0.9, 0.9, 1.0)), 256, 192, 40, "cull")

    for angle in (0.0, 0.4, 1.3):
        devproj.off()
        want = scenes.render(build(product, angle)).copy()
        devproj.on()
        got = scenes.render(build(product, angle)).copy()
        assert np.array_equal(got, want), (cull, angle)
        assert (got[..., :3].max(axis=2) > 0).any()


def test_device_projection_vs_oracle(oracle, product, devproj):
    devproj.on()
    kw = dict(width=400, height=240, logo_size=64, distance=0.7, textured=True)
    got = scenes.render(scenes.cube_scene(product, **kw))
    ref = scenes.render(scenes.cube_scene(oracle, **kw))
    assert np.array_equal(got, ref)


def test_missing_normals_is_an_error(product, devproj):
    devproj.on()
    scene = product.Scene.from_static([], [product.Batch3D.from_box(-0.5, -0.5, -0.5, 1, 1, 1)])
    v, p = product.D3OrbitCamera.new().matrices(64.0, 64.0)
    out = np.zeros(64 * 64 * 4, np.uint8)
    with pytest.raises(B.RasterizeError) as e:
        product.Rasterizer.setup(None, v, p).rasterize(scene, out, 64, 64, 16, product.Assets.default())
    assert e.value.code == B.RXR_ERR_INVALID


# ---- the 2D half of row N1: Batch2D::project (src/batch/batch2d.rs:373-425) and the 2D primitive records on the device ----------------
def _2d_cases():
    from tests.test_gpu_parity import lines_scene, scene_2d

    def with_matrix(api):
        cfg = scene_2d(api)
        m = B.Mat3.from_rows([[1.5, 0.0, 12.0], [0.0, 1.5, -8.0], [0.0, 0.0, 1.0]])

        def setup():
            v, p = api.D3OrbitCamera.new().matrices(float(cfg.width), float(cfg.height))
            return api.Rasterizer.setup(m, v, p).render_mode(B.RenderMode.render_2d()).ambient((1.0, 1.0, 1.0, 1.0))

        cfg.setup = setup
        return cfg

    def rotated(api):
        cfg = lines_scene(api)
        m = B.Mat3.from_rows([[0.8, -0.6, 40.0], [0.6, 0.8, -20.0], [0.0, 0.0, 1.0]])  # a rotation: both columns of the product matter

        def setup():
            v, p = api.D3OrbitCamera.new().matrices(float(cfg.width), float(cfg.height))
            return api.Rasterizer.setup(m, v, p).render_mode(B.RenderMode.render_2d()).background((5, 5, 5, 255))

        cfg.setup = setup
        return cfg

    return {
        "rectangles": lambda api: scene_2d(api),
        "rectangles, lights + linedef + occluder": lambda api: scene_2d(api, ambient=(0.5, 0.5, 0.5, 1.0), lights=True, linedef=True),
        "rectangles, ragged frame": lambda api: scene_2d(api, width=203, height=77),
        "rectangles under a Mat3": with_matrix,
        "lines, strips and loops": lambda api: lines_scene(api),
        "lines under a rotation": rotated,
        "tile map (binned 2D pass)": lambda api: scenes.tile_map_2d_scene(api, width=640, height=400, nx=30, ny=20),
        "tile map, stacked": lambda api: scenes.tile_map_2d_scene(api, width=400, height=250, nx=20, ny=12, stacked=3),
        "map scene: a 2D logo over the 3D frame": lambda api: scenes.map_scene(api, width=400, height=250, logo_size=64, n_lights=2),
    }


@pytest.mark.parametrize("name", list(_2d_cases()))
def test_2d_batches_projected_on_the_device(oracle, product, devproj, name):
    """the registered object-space 2D batches + the frame's Mat3 give the frame the host-projected batches give: bit for bit (2D is
    integer / exact), against the oracle and against the host-projected GPU frame; a second frame reuses the registration"""
    build = _2d_cases()[name]
    ref = scenes.render(build(oracle)).copy()
    devproj.off()
    host = scenes.render(build(product)).copy()
    devproj.on()
    cfg = build(product)
    dev = scenes.render(cfg).copy()
    dev2 = scenes.render(cfg).copy()
    devproj.off()
    lit3d = "map scene" in name   # (its 3D part is lit: +-1 against the oracle, identical between the two GPU paths)
    assert np.array_equal(dev, host), f"{name}: device- and host-projected frames differ in {(dev != host).any(axis=2).sum()} pixels"
    assert np.array_equal(dev2, dev)
    if lit3d:
        assert int(np.abs(dev.astype(np.int16) - ref.astype(np.int16)).max()) <= 1
    else:
        assert np.array_equal(dev, ref), f"{name}: {(dev != ref).any(axis=2).sum()} pixels differ from the oracle"
    assert (dev[..., :3].max(axis=2) > 0).mean() > 0.02


def test_2d_line_end_points_out_of_range_are_refused_like_on_the_host(product, devproj):
    """a visible segment whose projected end point lies beyond +-2^30 (the Bresenham walk would not end in a frame's time): the host
    builder refuses the frame at upload, the device sets a status word and rxr_synchronize answers RXR_ERR_UNSUPPORTED"""
    def build(api):
        v = np.array([[10.0, 10.0], [3.0e9, 40.0]], np.float32)
        lines = api.Batch2D.new(v, np.array([[0, 1, 0]], np.uint32), np.zeros_like(v)).mode(B.MODE_LINES).source(B.PixelSource.Pixel((255, 0, 0, 255)))
        scene = api.Scene.from_static([lines], [])

        def setup():
            v_, p_ = api.D3OrbitCamera.new().matrices(160.0, 96.0)
            return api.Rasterizer.setup(None, v_, p_).render_mode(B.RenderMode.render_2d())

        return scenes._result(api, scene, api.Assets.default(), setup, 160, 96, 40, "far line")

    for on in (False, True):
        (devproj.on if on else devproj.off)()
        with pytest.raises(B.RasterizeError) as e:
            scenes.render(build(product))
        assert e.value.code == B.RXR_ERR_UNSUPPORTED and "2^30" in str(e.value), str(e.value)
    devproj.off()
    # and the context renders the next frame
    from tests.test_gpu_parity import scene_2d

    assert scenes.render(scene_2d(product)).any()


def test_projected_arrays_of_poisoned_triangles_match_the_host_mirror(product, devproj):
    """tests/test_gpu_special_inputs.py's scene -- one NaN / +-inf / +-0 / denormal / huge number per triangle in a position or a texture
    coordinate -- through clip_and_project on the device: vertices, uvs, normals, indices, the Edges records and the bounding box equal
    the host mirror's (NaN == NaN: the two sides' NaNs may differ in sign and payload)"""
    from tests import test_gpu_special_inputs as SI

    devproj.off()
    ref_cfg = SI.build(product, B.SAMPLE_LINEAR, B.REPEAT_REPEAT_XY, False)
    ref_cfg.setup().project(ref_cfg.scene, ref_cfg.width, ref_cfg.height)
    devproj.on()
    cfg = SI.build(product, B.SAMPLE_LINEAR, B.REPEAT_REPEAT_XY, False)
    scenes.render(cfg)
    i, compared = 0, 0
    while True:
        try:
            ref = ref_cfg.scene.projected_batch3d(B.LIST_STATIC, i)
        except IndexError:
            break
        nv, nt = ref["projected_vertices"].shape[0], ref["clipped_indices"].shape[0]
        got = devproj.read(i, nv + 16, nt + 16)
        if got["projected_vertices"].shape[0] == 0 or ref["bounding_box"][0] == 0.0 and nv == 0:
            assert got["projected_vertices"].shape[0] == nv or ref["bounding_box"][0] == 0.0, (i, got["projected_vertices"].shape, nv)
        else:
            for key in ("projected_vertices", "clipped_uvs", "clipped_normals", "clipped_indices", "edges", "bounding_box"):
                assert got[key].shape == ref[key].shape, (i, key, got[key].shape, ref[key].shape)
                same = np.array_equal(got[key], ref[key], equal_nan=True) if got[key].dtype.kind == "f" else np.array_equal(got[key], ref[key])
                assert same, (i, key, got[key].tolist()[:4], ref[key].tolist()[:4])
            compared += 1
        i += 1
    assert i == 41 and compared >= 30, (i, compared)


def _sparse_mesh_scene(api, kind):
    """sparse frames of device-projected meshes: `grid` a distant lattice (a band in the middle of the frame, row mode), `grid_and_rect` the
    same with a host-projected 2D rectangle low in the frame (the table's host part), `nothing` every mesh off the frame"""
    from tests.test_gpu_parity import _sparse_scene

    cfg = _sparse_scene(api, "nothing" if kind == "nothing" else "grid")
    if kind == "grid_and_rect":
        rect = api.Batch2D.from_rectangle(500.0, 400.0, 90.0, 60.0).source(B.PixelSource.Pixel((40, 200, 90, 255)))
        cfg.scene.add_d2_static(rect)
    return cfg


@pytest.mark.parametrize("kind", ["grid", "grid_and_rect", "nothing"])
def test_sparse_frames_take_their_row_spans_from_the_device_boxes(oracle, product, devproj, kind, monkeypatch):
    """Device-projected meshes have their boxes on the device only: k_spans_from_meshes completes the row-span table there (the same box
    arithmetic as rxr_upload_frame's for host-projected batches), the tiles outside it are filled, not rastered.  The frame equals the
    oracle's, the host-projected one, the one with the spans switched off, and the one assembled from row bands."""
    import ctypes as C

    monkeypatch.setenv("RXR_CONTENT_MIN_TILES", "0")   # (the table only pays on large frames: these are small)
    devproj.off()
    want = scenes.render(_sparse_mesh_scene(product, kind)).copy()
    ref = scenes.render(_sparse_mesh_scene(oracle, kind))
    assert np.array_equal(want, ref), kind
    devproj.on()
    cfg = _sparse_mesh_scene(product, kind)
    got = scenes.render(cfg).copy()
    assert np.array_equal(got, ref), f"{kind}: {(got != ref).any(axis=2).sum()} pixels differ"
    lib = product.lib
    rxr = C.CDLL(__import__("rusterix_amd").lib_paths()["rxr"])   # (a handle of its own: the argtypes set below stay here)
    lib.rxh_context.restype = C.c_void_p
    info = (C.c_uint32 * 4)()
    assert rxr.rxr_debug_content(C.c_void_p(lib.rxh_context()), info) == 0
    assert info[0] == 0 and info[3] == 2, list(info)     # (no rows known to the host; the table is the device's)
    if kind != "nothing":
        hit_cols = np.nonzero((got[..., :3].max(axis=2) > 0).any(axis=0))[0]
        assert hit_cols.min() > 32 or hit_cols.max() < cfg.width - 32, "the scene leaves no tile columns empty: it tests nothing"
    again = scenes.render(cfg).copy()                     # (the second frame re-uploads the host part and completes it again)
    assert np.array_equal(again, ref)
    monkeypatch.setenv("RXR_ROW_SPANS", "0")
    off = scenes.render(cfg).copy()
    assert rxr.rxr_debug_content(C.c_void_p(lib.rxh_context()), info) == 0 and info[3] == 0
    monkeypatch.delenv("RXR_ROW_SPANS")
    assert np.array_equal(off, ref)
    # row bands of one resident frame (every band launch completes the table again: idempotent), then the pipelined download
    lib.rxh_rasterizer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rxr.rxr_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    rxr.rxr_download_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
    r = cfg.setup()
    assert lib.rxh_rasterizer_upload(r._h, cfg.scene._h, cfg.width, cfg.height, cfg.tile_size, cfg.assets._h) == 0
    ctx = C.c_void_p(lib.rxh_context())
    bands = np.zeros_like(got)
    H = cfg.height
    for a, b in [(0, 37), (37, H // 2 + 5), (H // 2 + 5, H)]:
        assert rxr.rxr_render_rows(ctx, a, b) == 0
        assert rxr.rxr_download_rows(ctx, bands.ctypes.data_as(C.POINTER(C.c_uint8)), a, b) == 0   # (full-frame layout: row r at r * width * 4)
    assert np.array_equal(bands, ref), kind
    rxr.rxr_render_download.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
    piped = np.full_like(got, 7)
    assert rxr.rxr_render_download(ctx, piped.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert np.array_equal(piped, ref), kind


def test_pipelined_download_of_a_sparse_device_projected_frame(product, devproj, monkeypatch):
    """rxr_render_download on a frame of 4 Mpixel and more whose meshes are projected on the device: the bands are rastered behind ONE
    pre-pass, k_spans_from_meshes hands the completed row-span table back to the host (page-locked memory + an event), the rows that
    nothing reaches are written by the host instead of crossing PCIe.  Equal to render_rows + download_rows, to the host-projected
    frame, and to itself on a second call."""
    import ctypes as C

    monkeypatch.setenv("RXR_CONTENT_MIN_TILES", "0")
    kw = dict(n=40, width=2304, height=1832, distance=22.0)   # 19 200 triangles in the middle of the frame: binned, row mode
    devproj.off()
    want = scenes.render(scenes.box_grid_scene(product, **kw)).copy()
    hit_rows = np.nonzero((want[..., :3].max(axis=2) > 0).any(axis=1))[0]
    assert len(hit_rows) and hit_rows.min() > 200 and hit_rows.max() < 1832 - 200, "the scene leaves no rows empty: it tests nothing"
    devproj.on()
    cfg = scenes.box_grid_scene(product, **kw)
    piped = np.full_like(want, 7)
    piped[...] = scenes.render(cfg)                 # Rasterizer::rasterize -> rxr_render_download
    assert np.array_equal(piped, want), f"{(piped != want).any(axis=2).sum()} pixels differ"
    lib = product.lib
    rxr = C.CDLL(__import__("rusterix_amd").lib_paths()["rxr"])   # (a handle of its own: the argtypes set below stay here)
    lib.rxh_context.restype = C.c_void_p
    info = (C.c_uint32 * 4)()
    assert rxr.rxr_debug_content(C.c_void_p(lib.rxh_context()), info) == 0 and info[3] == 2, list(info)
    lib.rxh_rasterizer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rxr.rxr_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    rxr.rxr_download_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
    rxr.rxr_render_download.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
    r = cfg.setup()
    assert lib.rxh_rasterizer_upload(r._h, cfg.scene._h, cfg.width, cfg.height, cfg.tile_size, cfg.assets._h) == 0
    ctx = C.c_void_p(lib.rxh_context())
    single = np.zeros_like(want)
    assert rxr.rxr_render_rows(ctx, 0, cfg.height) == 0
    assert rxr.rxr_download_rows(ctx, single.ctypes.data_as(C.POINTER(C.c_uint8)), 0, cfg.height) == 0
    assert np.array_equal(single, want)
    again = np.full_like(want, 9)                   # (every byte of the caller's buffer is written: by the link or by the host)
    assert rxr.rxr_render_download(ctx, again.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert np.array_equal(again, want)
    sent = (C.c_uint64 * 2)()
    assert rxr.rxr_debug_download_bytes(ctx, sent) == 0
    assert sent[0] + sent[1] == want.nbytes and 0 < sent[0] < 0.5 * want.nbytes, list(sent)
    # ... also when the columns outside the spans travel with the rows (RXR_NO_COLUMN_TRIM), and into an unaligned buffer (the host's
    # fills then write bytes)
    monkeypatch.setenv("RXR_NO_COLUMN_TRIM", "1")
    whole_rows = np.full_like(want, 5)
    assert rxr.rxr_render_download(ctx, whole_rows.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert np.array_equal(whole_rows, want)
    monkeypatch.delenv("RXR_NO_COLUMN_TRIM")
    raw = np.full(want.size + 8, 3, np.uint8)
    odd = raw[1:1 + want.size]
    assert rxr.rxr_render_download(ctx, odd.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert np.array_equal(odd.reshape(want.shape), want) and raw[0] == 3 and (raw[1 + want.size:] == 3).all()


def _column_scene(api, width=2304, height=1832):
    """a column of 240 small boxes (y = -7 .. 3) from below the bottom edge of the frame to above its top edge: content in EVERY tile row, in a fifth
    of the tile columns -- no rows to clamp, only row ends"""
    tmpl = api.Batch3D.from_box(0.0, 0.0, 0.0, 0.1, 0.1, 0.1)
    tv, ti, tuv, _ = tmpl.geometry()
    rng = np.random.default_rng(5)
    scene = api.Scene.empty()
    for b in range(4):
        vs, is_, uvs = [], [], []
        for k in range(60):
            v = tv.copy()
            v[:, 0] += np.float32(rng.uniform(-0.25, 0.25))
            v[:, 1] += np.float32(-7.0 + (10.0 / 240.0) * (b * 60 + k))   # (the orbit camera looks down: the column has to start far below)
            v[:, 2] += np.float32(rng.uniform(-0.25, 0.25))
            vs.append(v)
            is_.append(ti + np.uint32(24 * k))
            uvs.append(tuv)
        batch = api.Batch3D.new(np.concatenate(vs), np.concatenate(is_), np.concatenate(uvs))
        scene.add_d3_static(batch.source(B.PixelSource.StaticTileIndex(b % 2)).repeat_mode(B.REPEAT_REPEAT_XY).with_computed_normals())
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.noise_texture(300 + k)) for k in range(2)])
    cam = api.D3OrbitCamera.new()
    cam.center = (0.0, 0.0, 0.0)
    cam.distance = 3.0

    def setup():
        v, p = cam.matrices(float(width), float(height))
        return api.Rasterizer.setup(None, v, p).ambient((1.0, 1.0, 1.0, 1.0))

    return scenes._result(api, scene, assets, setup, width, height, 40, "column")


@pytest.mark.parametrize("device_projected", [False, True])
def test_pipelined_download_trims_row_ends_of_a_frame_without_empty_rows(oracle, product, devproj, device_projected, monkeypatch):
    """content in every tile row but in few tile columns: nothing to clamp in rows, yet only the strips inside the row spans cross PCIe and
    the host writes the row ends -- equal to the oracle's frame and to the download of whole rows, in both projection modes"""
    import ctypes as C

    monkeypatch.setenv("RXR_CONTENT_MIN_TILES", "0")
    ref = scenes.render(_column_scene(oracle))
    hit = got_cols = (ref[..., :3].max(axis=2) > 0)
    rows_hit = hit.any(axis=1)
    assert rows_hit[:16].any() and rows_hit[-16:].any(), "the column does not reach the first and the last tile row: the test tests nothing"
    assert got_cols.any(axis=0).mean() < 0.5
    (devproj.on if device_projected else devproj.off)()
    cfg = _column_scene(product)
    piped = np.full_like(ref, 7)
    piped[...] = scenes.render(cfg)
    assert np.array_equal(piped, ref), f"{(piped != ref).any(axis=2).sum()} pixels differ"
    rxr = C.CDLL(__import__("rusterix_amd").lib_paths()["rxr"])
    product.lib.rxh_context.restype = C.c_void_p
    info = (C.c_uint32 * 4)()
    assert rxr.rxr_debug_content(C.c_void_p(product.lib.rxh_context()), info) == 0
    assert info[3] == (2 if device_projected else 1), list(info)
    sent = (C.c_uint64 * 2)()
    assert rxr.rxr_debug_download_bytes(C.c_void_p(product.lib.rxh_context()), sent) == 0
    assert sent[0] + sent[1] == ref.nbytes and sent[0] < 0.6 * ref.nbytes, list(sent)   # (the strips: a part of every row)
    monkeypatch.setenv("RXR_NO_COLUMN_TRIM", "1")
    whole = np.full_like(ref, 9)
    whole[...] = scenes.render(cfg)
    assert np.array_equal(whole, ref)
    assert rxr.rxr_debug_download_bytes(C.c_void_p(product.lib.rxh_context()), sent) == 0
    assert sent[0] == ref.nbytes and sent[1] == 0, list(sent)                          # (no empty rows: everything travels)
