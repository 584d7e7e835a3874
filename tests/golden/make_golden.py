#!/usr/bin/env python3
"""Regenerates tests/golden/frames.npz and tests/golden/crc.json from the CPU oracle.

The reference has no golden images for the rasterizer and cannot be built here, so
these vectors are produced by the oracle (the C++ restatement of the reference algorithm) and serve
two purposes: they freeze the oracle against accidental change, and they travel to the GPU box,
where the HIP path is compared with them.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from rusterix_amd import binding as B  # noqa: E402
from rusterix_amd import scenes  # noqa: E402

# name -> (builder, kwargs, lit)   lit frames are compared with tolerance 1 on the GPU
# (rect_size=12: at thumbnail size the drivers' 200 x 200 rectangle would cover the whole frame and hide the 3D part)
SMALL = {
    "cube_textured": (scenes.cube_scene, dict(rect_size=12, width=64, height=48, textured=True, distance=3.0, logo_size=32), False),
    "cube_off_source": (scenes.cube_scene, dict(rect_size=12, width=64, height=48, textured=False, distance=3.0, logo_size=32), False),
    "cube_linear": (scenes.cube_scene, dict(rect_size=12, width=64, height=48, textured=True, distance=2.0, logo_size=32, sample_mode=B.SAMPLE_LINEAR), False),
    "cube_near_clip": (scenes.cube_scene, dict(rect_size=12, width=64, height=48, textured=True, distance=0.7, logo_size=32), False),
    "teapot_ambient": (scenes.teapot_scene, dict(rect_size=12, width=64, height=36, logo_size=32), False),
    "teapot_lit": (scenes.teapot_scene, dict(rect_size=12, width=64, height=36, logo_size=32, with_light=True), True),
    "map_1_light": (scenes.map_scene, dict(rect_size=12, width=64, height=36, logo_size=32, n_lights=1), True),
    "map_16_lights": (scenes.map_scene, dict(rect_size=12, width=64, height=36, logo_size=32, n_lights=16), True),
    "box_grid": (scenes.box_grid_scene, dict(n=8, width=64, height=36), False),
    "grid_editor": (scenes.grid_editor_scene, dict(width=96, height=64, grid_size=17.0, subdivisions=3.0, offset=(5.5, -3.25)), False),
    "small_triangle_mesh": (scenes.small_triangle_mesh_scene, dict(width=96, height=64, n_triangles=900), False),
    "mesh_brush_preview": (scenes.small_triangle_mesh_scene, dict(width=96, height=64, n_triangles=300, brush=((0.0, 0.0, 0.0), 2.5, 0.5)), False),
}
LARGE = {
    "cube_800x600": (scenes.cube_scene, dict(width=800, height=600, textured=True, distance=3.0, logo_size=256), False),
    "teapot_480x270": (scenes.teapot_scene, dict(width=480, height=270, logo_size=128), False),
    "map16_640x360": (scenes.map_scene, dict(width=640, height=360, logo_size=64, n_lights=16), True),
    "box_grid_512x288": (scenes.box_grid_scene, dict(n=24, width=512, height=288), False),
    "grid_editor_640x400": (scenes.grid_editor_scene, dict(width=640, height=400), False),
    "small_triangle_mesh_640x400": (scenes.small_triangle_mesh_scene, dict(width=640, height=400, n_triangles=6000, brush=((0.5, 0.0, 0.2), 3.0, 0.7)), False),
}


def main():
    from tests.oracle_api import load_oracle

    orc = load_oracle()
    frames = {}
    for name, (builder, kw, _lit) in SMALL.items():
        frames[name] = scenes.render(builder(orc, **kw)).copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **frames)
    crc = {}
    for name, (builder, kw, _lit) in LARGE.items():
        img = scenes.render(builder(orc, **kw))
        crc[name] = dict(crc32=zlib.crc32(img.tobytes()), shape=list(img.shape), mean=[round(float(x), 4) for x in img.reshape(-1, 4).mean(0)])
    json.dump(crc, open(os.path.join(ROOT, "tests", "golden", "crc.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(frames), "frames and", len(crc), "checksums")


if __name__ == "__main__":
    main()
