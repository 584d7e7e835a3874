"""Rasterizers on several host threads (the reference's Rasterizer is an independent value; here they share the process's device
context, and every entry point of the host mirror holds one lock for its whole duration -- ADVICE round 1): frames rendered
concurrently from three threads must equal the frames rendered one after the other."""
import threading

import numpy as np
import pytest

from rusterix_amd import scenes

pytestmark = pytest.mark.gpu


def test_concurrent_rasterizers_give_the_serial_frames(product):
    builders = [
        lambda: scenes.map_scene(product, width=320, height=180, logo_size=16, n_lights=3),
        lambda: scenes.box_grid_scene(product, n=24, width=256, height=144),
        lambda: scenes.tile_map_2d_scene(product, width=240, height=150, nx=10, ny=6),
    ]
    want = [scenes.render(b()).copy() for b in builders]
    errors = []

    def work(k):
        try:
            for _ in range(12):
                got = scenes.render(builders[k]())
                if not np.array_equal(got, want[k]):
                    errors.append((k, int((got != want[k]).any(axis=2).sum())))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(builders))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
