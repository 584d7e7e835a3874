#!/usr/bin/env python3
"""Driver of tools/sanitize_cpu.sh: renders the seeded scenes of the test suite on the (sanitizer-built) oracle and projects them
through the (sanitizer-built) host mirror.  CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402
from tests import test_gpu_chunks as K  # noqa: E402
from tests import test_gpu_fuzz as F  # noqa: E402
from tests import test_gpu_rows as R  # noqa: E402
from tests import test_gpu_shaders as S  # noqa: E402

orc = load_oracle()
prod = rusterix_amd.load()
maps = open("/proc/self/maps").read()
assert "librusterix_oracle_asan.so" in maps and "librusterix_host_asan.so" in maps and "libasan" in maps, "the sanitizer builds are not the libraries in use"
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
done = 0


def both(build):
    """the oracle renders the scene; the host mirror builds and projects the same scene (its frame needs a GPU)"""
    global done
    scenes.render(build(orc))
    cfg = build(prod)
    cfg.setup().project(cfg.scene, cfg.width, cfg.height)
    done += 1


both(lambda api: scenes.cube_scene(api, width=160, height=100, distance=0.7))
both(lambda api: scenes.teapot_scene(api, width=240, height=135, logo_size=16))
both(lambda api: scenes.map_scene(api, width=320, height=180, logo_size=16, n_lights=4))
both(lambda api: scenes.box_grid_scene(api, n=48, width=320, height=180))           # (above the worker pool's threshold)
both(lambda api: scenes.box_grid_scene(api, n=12, width=160, height=90, shader=True))
both(lambda api: scenes.tile_map_2d_scene(api, width=320, height=200, nx=12, ny=8, stacked=30))
both(lambda api: scenes.grid_editor_scene(api))
both(lambda api: K.panes_of_one_chunk_scene(api, 6, True))
both(lambda api: K.nested_windows_scene(api, 4))
for s in range(100, 100 + n_seeds):
    both(lambda api: F.build(api, s, 160 + 16 * (s % 3), 100 + 7 * (s % 4)))
    both(lambda api: F.build_chunks(api, s, 168, 104))
    both(lambda api: F.build_chunks(api, s, 168, 104, dense=40))
    for variant in ("plain", "ties", "cutout", "mixed", "opacity"):
        both(lambda api: R.build(api, s, 203, 131, variant))
    rng = np.random.default_rng([0x52585231, 4242, s])
    prog = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3))).program()
    scenes.render(S.rect_scene(orc, prog, time=0.5))
    rng = np.random.default_rng([0x52585231, 777, s])
    prog = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3))).program()
    scenes.render(S.cube_scene(orc, prog))
    done += 2
print(f"sanitizers: clean ({done} scenes on the oracle and through the host mirror's projection)")
