#!/usr/bin/env python3
"""Per-phase wave-cycle breakdown of the raster kernel.  Needs a library built with -DRXR_PHASE_TIMING=1
(tuning build, see tools/try_variants.sh); renders the bench frame a few times and prints the share of
wave lifetime spent in each phase (s_memtime deltas summed over waves)."""
import argparse
import ctypes as C
import os

os.environ.setdefault("RXR_SHADER_JIT", "0")  # (measurements name their mode: interpreted unless asked otherwise)
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lights", type=int, default=16)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--scene", default="map", help="map | boxes | boxes_shader (the reduced C5 grid, 1920x1080) | c5 | c5_shader (the full 1 M-triangle grid)")
a = ap.parse_args()
prod = rusterix_amd.load()
rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
if a.scene == "map":
    cfg = scenes.map_scene(prod, width=a.width, height=a.height, n_lights=a.lights)
elif a.scene.startswith("c5"):
    cfg = scenes.box_grid_scene(prod, n=289, width=7680, height=4320, shader=a.scene == "c5_shader")
else:
    cfg = scenes.box_grid_scene(prod, n=96, width=1920, height=1080, shader=a.scene == "boxes_shader")
scenes.render(cfg)
out = (C.c_ulonglong * 16)()
assert rxr.rxr_debug_phase_read(out, 1) == 0, "library was not built with -DRXR_PHASE_TIMING=1"
import time
t0 = time.time()
for _ in range(a.frames):
    scenes.render(cfg)
print(f"wall per frame incl. host work: {(time.time() - t0) / a.frames * 1e3:.2f} ms")
assert rxr.rxr_debug_phase_read(out, 1) == 0
names = {0: "prologue (+ opacity pass)", 1: "lists: staging / walk", 7: "row mode (rows_round)", 10: "walk of a binned round", 9: "rows_resolve", 2: "shade begin", 3: "lights",
         4: "shade end", 5: "2D pass", 6: "store"}
tot = sum(out[k] for k in names)
waves = out[8]
print(f"scene={a.scene} lights={a.lights} waves={waves} cycles/wave={tot / max(waves, 1):.0f}")
for k, name in names.items():
    print(f"  {name:26s} {out[k] / max(waves, 1):9.0f} cyc/wave  {100.0 * out[k] / max(tot, 1):5.1f} %")
