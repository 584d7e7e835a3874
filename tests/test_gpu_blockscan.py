"""Binned scenes with a spatially coherent triangle order (meshes): the bin lists built by k_blockscan (rusterix_amd/csrc/rxr_kernels.hip, rxr_device.h
RXR_BLOCKSCAN_*) instead of the general count / scan / fill pipeline.  The lists only have to name the same candidates per bin
(the visibility pass is an arg-min over them: reference src/rasterizer.rs:1020-1060), so every frame must equal the general
pipeline's frame and the oracle's byte for byte; a bin or a block of bins with more candidates than the kernel keeps must send
the frame through the general pipeline, not drop candidates."""
import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from tests.test_gpu_parity import assert_exact

pytestmark = pytest.mark.gpu


def builders():
    return {
        "teapot 960x540": lambda api: scenes.teapot_scene(api, width=960, height=540, logo_size=64, rect_size=40.0),
        "teapot 333x211 (ragged tiles)": lambda api: scenes.teapot_scene(api, width=333, height=211, logo_size=64, rect_size=20.0),
        "box grid 16 x 16 (3072 triangles)": lambda api: scenes.box_grid_scene(api, n=16, width=640, height=360),
        "box grid 18 x 18 (3888 triangles), Nearest": lambda api: scenes.box_grid_scene(api, n=18, width=480, height=270, sample_mode=B.SAMPLE_NEAREST),
    }


@pytest.mark.parametrize("name", list(builders()))
def test_blockscan_frames_equal_the_general_pipeline_and_the_oracle(oracle, product, monkeypatch, name):
    build = builders()[name]
    ref = scenes.render(build(oracle)).copy()
    monkeypatch.setenv("RXR_BLOCKSCAN", "0")
    general = scenes.render(build(product)).copy()
    monkeypatch.delenv("RXR_BLOCKSCAN")
    got = scenes.render(build(product)).copy()
    assert_exact(got, general, f"{name}: k_blockscan vs the general pipeline")
    assert_exact(got, ref, f"{name}: vs the oracle")
    assert (got[..., :3].max(axis=2) > 0).mean() > 0.02


@pytest.mark.parametrize("cap", [1, 3, 16])
def test_a_full_bin_sends_the_frame_through_the_general_pipeline(oracle, product, monkeypatch, cap):
    """RXR_BLOCKSCAN_CAP slots per bin: the teapot has bins with far more candidates, the overflow word is raised and
    rxr_synchronize renders the frame again with count / scan / fill -- the caller sees the complete frame"""
    build = builders()["teapot 960x540"]
    ref = scenes.render(build(oracle)).copy()
    monkeypatch.setenv("RXR_BLOCKSCAN_CAP", str(cap))
    got = scenes.render(build(product)).copy()
    assert_exact(got, ref, f"teapot with {cap} slots per bin")
    # a second frame through the same context: a frame of the same shape (triangles, bins) as one that overflowed goes straight
    # to the general pipeline (a caller that uploads every frame must not pay the failed attempt every time)
    got2 = scenes.render(build(product)).copy()
    assert_exact(got2, ref, f"teapot with {cap} slots per bin, second frame")


def test_stripes_and_bands_of_a_mid_sized_scene(product):
    """launch-local tile rows (bands, interleaved stripes): bin_range is the general pipeline's own function"""
    import ctypes as C

    import rusterix_amd

    cfg = scenes.teapot_scene(product, width=640, height=250, logo_size=64, rect_size=20.0)   # 250 rows: ragged last stripe
    full = scenes.render(cfg).copy()
    rxr = rusterix_amd.rxr_abi()
    host = product.lib
    r = cfg.setup()
    assert host.rxh_rasterizer_upload(r._h, cfg.scene._h, cfg.width, cfg.height, cfg.tile_size, cfg.assets._h) == 0
    ctx = host.rxh_context()
    out = np.zeros((cfg.height, cfg.width, 4), np.uint8)
    for row0, row1 in ((0, 64), (64, 176), (176, 250)):
        assert rxr.rxr_render_rows(ctx, row0, row1) == 0
        assert rxr.rxr_download_rows(ctx, out.ctypes.data_as(C.c_void_p), row0, row1) == 0
    assert_exact(out, full, "three bands")


def test_empty_groups_are_listed_nowhere(oracle, product):
    """regression: the mark of a group without a bin range once passed the overlap test of the block at the origin; with device
    projection the unused slots behind a mesh keep the boxes of the PREVIOUS frame, and those triangles came back in the top left
    corner.  A 6000-triangle mesh first, then the device-projected teapot."""
    scenes.render(scenes.small_triangle_mesh_scene(product, width=640, height=400, n_triangles=6000))
    build = builders()["teapot 960x540"]
    ref = scenes.render(build(oracle)).copy()
    product.lib.rxh_set_device_projection(1)
    try:
        got = scenes.render(build(product)).copy()
    finally:
        product.lib.rxh_set_device_projection(0)
    assert_exact(got, ref, "device-projected teapot after a larger mesh")


def test_device_projected_mid_sized_scene(oracle, product):
    """N1 + k_blockscan: the capacity-based triangle pools leave unused slots behind every mesh; k_setup3d gives them empty boxes
    in this mode (the general pipeline skips whole workgroups of them instead)"""
    build = builders()["teapot 960x540"]
    ref = scenes.render(build(oracle)).copy()
    product.lib.rxh_set_device_projection(1)
    try:
        got = scenes.render(build(product)).copy()
        got2 = scenes.render(build(product)).copy()   # (meshes resident: matrices only)
    finally:
        product.lib.rxh_set_device_projection(0)
    assert_exact(got, ref, "device-projected teapot")
    assert_exact(got2, ref, "device-projected teapot, second frame")


def test_large_coherent_scene_takes_the_scatter_form(oracle, product, monkeypatch):
    """more than 256 groups of 64 triangles: every group appends itself to the blocks of bins its range meets (k_setup3d), groups all
    over the screen go to the wide list; 64 x 64 boxes = 49 152 triangles on a small frame"""
    def build(api):
        return scenes.box_grid_scene(api, n=64, width=1600, height=900)

    ref = scenes.render(build(oracle)).copy()
    got = scenes.render(build(product)).copy()
    assert_exact(got, ref, "box grid 64 x 64")
    monkeypatch.setenv("RXR_BLOCKSCAN", "0")
    assert_exact(scenes.render(build(product)).copy(), got, "box grid 64 x 64: general pipeline vs k_blockscan")


# ---- the 2D bins (k_blockscan2d): lists in submission order, no per-tile sort ---------------------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(width=333, height=211, nx=17, ny=11), dict(width=1920, height=1080, nx=60, ny=34),
                                dict(lights=False, lines=False), dict(stacked=200, nx=12, ny=8)])
def test_2d_tile_maps_through_the_block_scan(oracle, product, monkeypatch, kw):
    """ordered 2D blending (reference src/rasterizer.rs:876-895): the lists k_blockscan2d leaves are already in submission
    order; the frame must equal the general pipeline's (count / scan / fill + a bitonic sort per tile) and the oracle's"""
    ref = scenes.render(scenes.tile_map_2d_scene(oracle, **kw)).copy()
    got = scenes.render(scenes.tile_map_2d_scene(product, **kw)).copy()
    monkeypatch.setenv("RXR_BLOCKSCAN2D", "0")
    general = scenes.render(scenes.tile_map_2d_scene(product, **kw)).copy()
    assert_exact(got, general, f"2D tile map {kw}: k_blockscan2d vs the general pipeline")
    assert_exact(got, ref, f"2D tile map {kw}: vs the oracle")


def test_a_pile_of_2d_primitives_falls_back(oracle, product):
    """600 rectangles stacked on one spot: more than a bin's 256 slots -> the 2D overflow word -> the general pipeline (whose tiles
    then walk every primitive in order, as before); the second frame of the same shape goes there directly"""
    kw = dict(stacked=600, nx=12, ny=8)
    ref = scenes.render(scenes.tile_map_2d_scene(oracle, **kw)).copy()
    assert_exact(scenes.render(scenes.tile_map_2d_scene(product, **kw)).copy(), ref, "stacked rectangles")
    assert_exact(scenes.render(scenes.tile_map_2d_scene(product, **kw)).copy(), ref, "stacked rectangles, second frame")


def test_scatter_form_in_interleaved_stripes(product):
    """the scatter form of the 3D block scan with launch-local tile rows: three logical members on GPU 0 render interleaved 16-row
    stripes of a 49 152-triangle grid (rxr_render_stripes_to inside rxr_rasterize); the frame equals the single-context frame"""
    import ctypes as C

    def build():
        return scenes.box_grid_scene(product, n=64, width=1280, height=720)

    product.lib.rxh_set_device(0)
    ref = scenes.render(build()).copy()
    try:
        ids = (C.c_int * 3)(0, 0, 0)
        product.lib.rxh_set_devices(ids, 3)
        got = scenes.render(build()).copy()
        got2 = scenes.render(build()).copy()
    finally:
        product.lib.rxh_set_device(0)
    assert_exact(got, ref, "3 members, scatter form")
    assert_exact(got2, ref, "3 members, scatter form, second frame")
