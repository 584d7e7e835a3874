"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/rxr.h
declares, fails loudly without a GPU (no CPU fallback), and the C++ host mirror's projection
(Scene::project -> clip_and_project / project / Edges) is bit-identical to the oracle's restatement.
No compute call is made on a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rusterix_amd
from rusterix_amd import binding as B
from rusterix_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rxr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:int|void|const char \*|void \*)\s*\*?\s*(rxr_[a-z_0-9]+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ["rxr_create", "rxr_destroy", "rxr_last_error", "rxr_device_count", "rxr_set_textures", "rxr_upload_frame",
                 "rxr_render_rows", "rxr_render_rows_to", "rxr_render_stripes_to", "rxr_download_rows", "rxr_rasterize", "rxr_render_download",
                 "rxr_synchronize", "rxr_get_stats", "rxr_device_framebuffer", "rxr_profile_begin", "rxr_profile_read", "rxr_profile_stride",
                 "rxr_set_meshes", "rxr_read_projected_mesh", "rxr_set_shaders", "rxr_selftest_math"]:
        assert must in names, must


def test_library_exports_every_declared_symbol():
    lib = rusterix_amd.load_rxr()
    for name in declared_functions():
        assert hasattr(lib, name), f"librxr_hip.so does not export {name}"


def test_host_library_exports_builder_api():
    api = rusterix_amd.load()
    for sym in ["rxh_rasterizer_rasterize", "rxh_rasterizer_upload", "rxh_context", "rxh_set_device", "rxh_last_error", "rxh_scene_project",
                "rxh_scene_add_program", "rxh_assets_set_patterns", "rxh_assets_set_palette", "rxh_chunk_set_terrain",
                "rxh_chunk_set_terrain_batch2d", "rxh_chunk_add_shader_texture", "rxh_set_device_projection", "rxh_rasterizer_brush_preview",
                "rxh_scene_set_background_grid", "rxh_set_light_math_exact", "rxh_get_light_math_exact"]:
        assert hasattr(api.lib, sym), sym


def test_struct_sizes_match_the_header():
    """rxr_light is passed by pointer from Python: its ctypes mirror must match the C layout."""
    assert C.sizeof(B.RxrLight) == 88


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    lib = rusterix_amd.load_rxr()
    lib.rxr_device_count.restype = C.c_int
    assert lib.rxr_device_count() == 0
    ctx = C.c_void_p()
    lib.rxr_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    rc = lib.rxr_create(C.byref(ctx), 0)
    assert rc == B.RXR_ERR_NO_DEVICE and not ctx.value
    lib.rxr_last_error.restype = C.c_char_p
    lib.rxr_last_error.argtypes = [C.c_void_p]
    assert b"no HIP device" in lib.rxr_last_error(None)
    # the reference-shaped call fails with the same status instead of producing pixels some other way
    prod = rusterix_amd.load()
    cfg = scenes.cube_scene(prod, width=32, height=32, logo_size=16)
    out = np.full(32 * 32 * 4, 7, np.uint8)
    with pytest.raises(B.RasterizeError) as e:
        cfg.setup().rasterize(cfg.scene, out, 32, 32, 16, cfg.assets)
    assert e.value.code == B.RXR_ERR_NO_DEVICE
    assert (out == 7).all(), "no pixel may be written without the HIP path"


def test_argument_validation_without_a_context():
    lib = rusterix_amd.load_rxr()
    for fn in ("rxr_upload_frame", "rxr_render_rows", "rxr_synchronize"):
        getattr(lib, fn).restype = C.c_int
    assert lib.rxr_upload_frame(None, None) == B.RXR_ERR_INVALID
    assert lib.rxr_render_rows(None, 0, 0) == B.RXR_ERR_INVALID
    assert lib.rxr_synchronize(None) == B.RXR_ERR_INVALID
    lib.rxr_destroy(None)  # must be a no-op


# ---- host mirror vs oracle: projection outputs must be bit-identical ----------------------------------
def projected(api, cfg):
    cfg.setup().project(cfg.scene, cfg.width, cfg.height)
    out = []
    for i in range(64):
        try:
            out.append(cfg.scene.projected_batch3d(B.LIST_STATIC, i))
        except IndexError:
            break
    return out


@pytest.mark.parametrize("builder,kw", [
    (scenes.cube_scene, dict(width=800, height=600, distance=20.0)),
    (scenes.cube_scene, dict(width=320, height=200, distance=0.7)),          # near-plane clipping
    (scenes.teapot_scene, dict(width=480, height=270, logo_size=16)),
    (scenes.map_scene, dict(width=640, height=360, logo_size=16, n_lights=2)),
    (scenes.box_grid_scene, dict(n=6, width=256, height=144)),
    # 48 batches of 48 boxes: above RXR_PARALLEL_MIN_WEIGHT, so Scene::project hands the batches to the host worker pool
    # (rusterix_amd/csrc/rxr_parallel.h; the reference's par_iter_mut, src/scene.rs:162-211) -- same bits, batch for batch
    (scenes.box_grid_scene, dict(n=48, width=640, height=360)),
])
def test_projection_matches_oracle_bit_for_bit(oracle, builder, kw):
    prod = rusterix_amd.load()
    a = projected(prod, builder(prod, **kw))
    b = projected(oracle, builder(oracle, **kw))
    assert len(a) == len(b) and len(a) > 0
    for x, y in zip(a, b):
        for key in ("projected_vertices", "clipped_uvs", "clipped_normals", "clipped_indices", "edges", "bounding_box"):
            assert x[key].shape == y[key].shape, key
            assert x[key].tobytes() == y[key].tobytes(), key
        assert x["has_normals"] == y["has_normals"]


@pytest.mark.parametrize("cull", [B.CULL_OFF, B.CULL_FRONT, B.CULL_BACK])
def test_cull_modes_match_oracle(oracle, cull):
    prod = rusterix_amd.load()

    def build(api):
        cam = api.D3OrbitCamera.new()
        cam.set_parameter_f32("distance", 0.8)
        v, p = cam.matrices(200.0, 150.0)
        box = api.Batch3D.from_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(cull).with_computed_normals()
        scene = api.Scene.from_static([], [box])
        api.Rasterizer.setup(None, v, p).project(scene, 200, 150)
        return scene.projected_batch3d(B.LIST_STATIC, 0)

    x, y = build(prod), build(oracle)
    for key in ("projected_vertices", "clipped_indices", "edges", "bounding_box"):
        assert x[key].tobytes() == y[key].tobytes(), key
    if cull != B.CULL_OFF:
        assert (x["edges"][:, 9] == 0).any()


def test_obj_parser_matches_oracle(oracle):
    prod = rusterix_amd.load()
    text = "# c\no t\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\nvn 0 0 1\nf 1//1 2//1 3//1\nf 2/9/1 4/9/1 3/9/1\n"
    a = prod.Batch3D.from_obj(text).geometry()
    b = oracle.Batch3D.from_obj(text).geometry()
    assert a[0].shape == (4, 4) and a[1].tolist() == [[0, 1, 2], [1, 3, 2]]
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    assert np.array_equal(a[2], a[0][:, :2])  # no vt: uv = (x, y), wavefront.rs:92-95


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/teapot.obj"), reason="reference checkout not present")
def test_real_teapot_counts():
    """examples/teapot.obj: 1202 v, 2256 f -- only where the reference is mounted."""
    prod = rusterix_amd.load()
    b = prod.Batch3D.from_obj(open("/root/reference/examples/teapot.obj").read())
    assert b.counts() == (1202, 2256)


def test_teapot_fixture_is_the_mesh_the_scenes_use():
    """tests/golden/teapot_mesh.npz (made by tests/golden/make_teapot_fixture.py from the reference's examples/teapot.obj): C2 runs
    on the real geometry wherever this repository goes, also where /root/reference does not exist (the GPU box)"""
    v, idx, uv, real = scenes.teapot_mesh()
    assert real and v.shape == (1202, 4) and idx.shape == (2256, 3) and (v[:, 3] == 1.0).all()
    assert np.array_equal(uv, v[:, :2])  # no `vt` in the file: wavefront.rs:92-95
    assert idx.max() == 1201 and len(np.unique(idx)) > 1100
    assert scenes.teapot_scene(rusterix_amd.load(), width=64, height=36, logo_size=16).name == "C2-teapot"


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/teapot.obj"), reason="reference checkout not present")
def test_teapot_fixture_equals_the_reference_file(oracle):
    """both OBJ readers (product host mirror and oracle) turn the reference's file into exactly the fixture's arrays"""
    text = open("/root/reference/examples/teapot.obj").read()
    v, idx, uv, _ = scenes.teapot_mesh()
    for api in (rusterix_amd.load(), oracle):
        g = api.Batch3D.from_obj(text).geometry()
        assert g[0].tobytes() == v.tobytes() and g[1].astype(np.uint32).tobytes() == idx.tobytes() and g[2].tobytes() == uv.tobytes()


def test_tile_span_without_the_searches_equals_the_searched_one():
    """rxr_device.h: rxr_ref_tile_span_quick (k_spans_from_meshes: the row spans of device-projected frames) guesses the two bounds of the
    reference's per-tile batch box test (rasterizer.rs:978-983) and walks to where the predicate flips; rxr_ref_tile_span searches for them.
    Same interval for every box: random ones, boxes on tile and pixel boundaries, off-screen, huge, infinite and NaN ones (host code only)."""
    lib = rusterix_amd.load_rxr()
    lib.rxr_debug_tile_spans.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_void_p]
    lib.rxr_debug_tile_spans.restype = None
    rng = np.random.default_rng(77)
    f32 = np.float32
    for size, ts in [(7680, 40), (4320, 40), (1920, 60), (1080, 64), (333, 40), (211, 7), (16, 100), (1, 1), (32768, 1), (640, 0)]:
        edges = np.arange(0, size + 2 * max(ts, 1), max(ts, 1), dtype=np.float64)
        near = (edges[:, None] + np.array([-1.0, -0.5, -2.0 ** -10, 0.0, 2.0 ** -10, 0.5, 1.0])[None, :]).ravel()
        special = np.array([0.0, -0.0, -1.0, size, size - 0.5, size + 0.5, 2097151.75, 2097152.0, -2097152.0, 4.0e6, -4.0e6, 3.0e38, -3.0e38, np.inf, -np.inf, np.nan])
        pool = np.concatenate([near, special, rng.uniform(-2.0 * size, 3.0 * size, 4000)]).astype(f32)
        lo = rng.choice(pool, 60000).astype(f32)
        hi = rng.choice(pool, 60000).astype(f32)
        with np.errstate(invalid="ignore", over="ignore"):
            extent = (hi - lo).astype(f32)     # (negative extents included: an empty or inverted box)
        for pad in (0.0, 0.5):
            out = [np.zeros((len(lo), 2), np.uint32) for _ in range(2)]
            for quick in (0, 1):
                lib.rxr_debug_tile_spans(lo.ctypes.data, extent.ctypes.data, len(lo), size, ts, f32(pad), quick, out[quick].ctypes.data)
            bad = np.nonzero((out[0] != out[1]).any(axis=1))[0]
            assert len(bad) == 0, (size, ts, pad, lo[bad[:3]], extent[bad[:3]], out[0][bad[:3]], out[1][bad[:3]])
            assert (out[0][:, 0] < out[0][:, 1]).sum() > 1000 or ts == 0 or size <= 16, "hardly any box meets the frame: the test tests nothing"
