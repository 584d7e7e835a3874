"""2D batches with one poisoned number each -- NaN, +-inf, +-0, a denormal, a large coordinate in a vertex position or a texture
coordinate of a blended triangle, in the end point of a segment, in an entry of the Mat3 -- through Batch2D::project and the ordered 2D
pass against the oracle, bit for bit (the 2D path is exact): both sampling modes, host- and device-projected."""
import ctypes as C

import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from tests.test_gpu_fuzz import random_texture

pytestmark = pytest.mark.gpu
NAN, INF = float("nan"), float("inf")
SPECIALS = [NAN, INF, -INF, 0.0, -0.0, 1e-40, 1.0e6, -1.0e6]     # (end points beyond +-2^30 are refused: tests/test_gpu_device_projection.py)
W, H = 224, 144


LINE_SPECIALS = [0.0, -0.0, 1e-40, 1.0e6, -1.0e6, 0.49, -0.51, 123456.7]   # finite: NaN and +-inf end points are refused (below)


def build(api, sample_mode, matrix=None, lines=LINE_SPECIALS):
    rng = np.random.default_rng([0x52585231, 707])
    assets = api.Assets.default().textures([B.Tile([random_texture(rng, 9, 7, 1)]), B.Tile([random_texture(rng, 6, 6, 0)])])
    batches = [api.Batch2D.from_rectangle(0.0, 0.0, float(W), float(H)).source(B.PixelSource.Pixel((30, 60, 90, 255)))]
    k = 0
    for field in range(4):                      # x, y of vertex 1; u, v of vertex 2
        for s in SPECIALS:
            cx, cy = 8.0 + 26.0 * (k % 8), 6.0 + 30.0 * (k // 8)
            v = np.array([[cx, cy], [cx + 22.0, cy + 3.0], [cx + 6.0, cy + 24.0]], np.float32)
            uv = np.array([[0.1, 0.2], [1.4, 0.3], [0.4, 1.7]], np.float32)
            if field < 2:
                v[1, field] = s
            else:
                uv[2, field - 2] = s
            t = api.Batch2D.new(v, np.array([[0, 1, 2]], np.uint32), uv).source(B.PixelSource.StaticTileIndex(k % 2)).repeat_mode(k % 4)
            batches.append(t)
            k += 1
    # segments with a poisoned end point
    for j, s in enumerate(lines):
        v = np.array([[10.0 + 20.0 * j, 125.0], [s, 140.0], [30.0 + 20.0 * j, s]], np.float32)
        ln = api.Batch2D.new(v, np.array([[0, 1, 0], [0, 2, 0]], np.uint32), np.zeros_like(v)).mode(B.MODE_LINES).source(B.PixelSource.Pixel((255, 220, 10, 255)))
        batches.append(ln)
    scene = api.Scene.from_static(batches, [])

    def setup():
        v_, p_ = api.D3OrbitCamera.new().matrices(float(W), float(H))
        return api.Rasterizer.setup(matrix, v_, p_).render_mode(B.RenderMode.render_2d()).sample_mode(sample_mode).background((5, 5, 5, 255))

    return scenes._result(api, scene, assets, setup, W, H, 40, "special-2d")


MATRICES = [None, B.Mat3.from_rows([[1.1, 0.2, 3.0], [-0.1, 0.9, 2.0], [0.0, 0.0, 1.0]]), B.Mat3.from_rows([[1.0, 0.0, NAN], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]),
            B.Mat3.from_rows([[INF, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]), B.Mat3.from_rows([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 1.0]])]


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("matrix", range(len(MATRICES)))
@pytest.mark.parametrize("sample_mode", [B.SAMPLE_NEAREST, B.SAMPLE_LINEAR])
def test_poisoned_2d_batches(oracle, product, sample_mode, matrix, device_projection):
    lines = LINE_SPECIALS if matrix in (0, 1, 4) else []      # (a NaN / inf matrix entry makes every end point NaN: refused, below)
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(build(product, sample_mode, MATRICES[matrix], lines))
    finally:
        product.lib.rxh_set_device_projection(0)
    ref = scenes.render(build(oracle, sample_mode, MATRICES[matrix], lines))
    d = (got != ref).any(axis=2)
    assert not d.any(), f"{int(d.sum())} pixels differ; first at {np.argwhere(d)[:3].tolist()}: device {got[tuple(np.argwhere(d)[0])].tolist()} oracle {ref[tuple(np.argwhere(d)[0])].tolist()}"
    if matrix in (0, 1):
        assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 6


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("bad", [NAN, INF, 2.0e9])
def test_segments_the_reference_cannot_draw_the_same_way_twice_are_refused(product, bad, device_projection):
    """an end point beyond +-2^30 (or +-inf: `as isize` saturates) makes the Bresenham walk endless in practice; a NaN end point is
    `0` as isize, a point outside the batch's bounding box (min / max drop NaN), and the reference skips the batch for every tile its
    box does not meet -- which pixels it draws depends on ITS tile size.  Both are refused (RXR_ERR_UNSUPPORTED: the caller's CPU path
    draws them), host- and device-projected; the context renders the next frame as usual."""
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        with pytest.raises(B.RasterizeError) as e:
            scenes.render(build(product, B.SAMPLE_NEAREST, None, [bad]))
        assert e.value.code == B.RXR_ERR_UNSUPPORTED and "line end point" in str(e.value)
        got = scenes.render(build(product, B.SAMPLE_NEAREST, None, [5.0]))
        assert int(got[..., 3].min()) == 255
    finally:
        product.lib.rxh_set_device_projection(0)


@pytest.mark.parametrize("device_projection", [False, True])
def test_a_batch_with_an_infinite_box_is_not_drawn_at_all(oracle, product, device_projection):
    """-inf in a vertex: the batch's box is (x = -inf, width = inf), `x + width` is NaN and the reference's box test (:594-600) is false
    for every tile -- nothing of the batch is drawn, its unrepresentable end point included (no refusal)"""
    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(build(product, B.SAMPLE_NEAREST, None, [-INF]))
    finally:
        product.lib.rxh_set_device_projection(0)
    assert np.array_equal(got, scenes.render(build(oracle, B.SAMPLE_NEAREST, None, [-INF])))


@pytest.mark.parametrize("device_projection", [False, True])
@pytest.mark.parametrize("tile_size", [8, 40, 64, 300])
@pytest.mark.parametrize("huge", [-3.0e38, 3.0e38, -1.0e12, 6.0e7, -2.5e6])
def test_batches_with_huge_coordinates_are_drawn_in_the_reference_s_tiles_only(oracle, product, huge, tile_size, device_projection):
    """a batch box is Rect {x: min, width: max - min}, and the reference skips the batch for every tile that `x + width` does not reach
    (rasterizer.rs:594-600): with a vertex at -3e38 and the others on screen, `x + width` is 0 and only the leftmost tile column draws
    the batch; with 6e7 the sum is off by a few pixels.  Which pixels that is depends on the reference's tile size -- the device clips
    such a batch's primitives to the tiles that pass (rxr_device.h rxr_ref_tile_span; found by tools/fuzz_special2.py, seed 1048)"""
    def build(api):
        batches = [api.Batch2D.from_rectangle(0.0, 0.0, float(W), float(H)).source(B.PixelSource.Pixel((30, 60, 90, 255)))]
        tris = [[[35.5, 108.6], [huge, 133.5], [35.7, 43.5]], [[huge, 61.8], [111.2, 140.1], [127.9, 8.2]], [[200.0, 20.0], [210.0, huge], [150.0, 90.0]]]
        for k, t in enumerate(tris):
            v = np.array(t, np.float32)
            batches.append(api.Batch2D.new(v, np.array([[0, 1, 2]], np.uint32), np.array([[0, 0], [1, 0], [0, 1]], np.float32)).source(B.PixelSource.Pixel((250 - 60 * k, 40 + 70 * k, 120, 200))))
        ln = np.array([[20.0, 130.0], [huge, 135.0], [100.0, 10.0]], np.float32)
        if abs(huge) < 1.0e9:   # (segments end within +-2^30)
            batches.append(api.Batch2D.new(ln, np.array([[0, 1, 0], [0, 2, 0]], np.uint32), np.zeros_like(ln)).mode(B.MODE_LINES).source(B.PixelSource.Pixel((255, 255, 0, 255))))
        scene = api.Scene.from_static(batches, [])

        def setup():
            v_, p_ = api.D3OrbitCamera.new().matrices(float(W), float(H))
            return api.Rasterizer.setup(None, v_, p_).render_mode(B.RenderMode.render_2d()).background((5, 5, 5, 255))

        return scenes._result(api, scene, api.Assets.default(), setup, W, H, tile_size, "huge-2d")

    product.lib.rxh_set_device_projection.argtypes = [C.c_int]
    product.lib.rxh_set_device_projection(1 if device_projection else 0)
    try:
        got = scenes.render(build(product))
    finally:
        product.lib.rxh_set_device_projection(0)
    ref = scenes.render(build(oracle))
    d = (got != ref).any(axis=2)
    assert not d.any(), f"{huge}, tile {tile_size}: {int(d.sum())} pixels differ; first at {np.argwhere(d)[:3].tolist()}"
