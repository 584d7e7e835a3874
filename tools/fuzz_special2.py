#!/usr/bin/env python3
"""Wide sweep of LIT scenes with random poisoned parameters: the map room with up to six lights of random types whose fields are
poisoned at random (NaN, +-inf, zero, negative, huge), poisoned occluders / linedefs / sun, a few poisoned 2D triangles on top, both
light-loop modes.  Judged like every lit 3D frame: within one step per channel, few pixels off.
usage: python tools/fuzz_special2.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rusterix_amd  # noqa: E402
from rusterix_amd import binding as B  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tests.oracle_api import load_oracle  # noqa: E402

NAN, INF = float("nan"), float("inf")
POOL = [NAN, INF, -INF, 0.0, -0.0, -1.0, 1e-40, 3.0e38, -3.0e38, 1e30]
prod, orc = rusterix_amd.load(), load_oracle()


def build(api, seed, use_lights=True, use_2d=True, use_extras=True, log=None):
    rng = np.random.default_rng([0x52585231, 1313, seed])
    pick = lambda: POOL[int(rng.integers(0, len(POOL)))]   # noqa: E731
    cfg = scenes.map_scene(api, width=256, height=144, n_lights=int(rng.integers(1, 4)), logo_size=16)
    for _ in range(int(rng.integers(1, 6))):
        t = [B.LIGHT_POINT, B.LIGHT_SPOT, B.LIGHT_AREA, B.LIGHT_AMBIENT, B.LIGHT_DAYLIGHT, B.LIGHT_AMBIENT_DAYLIGHT][int(rng.integers(0, 6))]
        l = B.Light(t).with_position(tuple(float(x) for x in rng.uniform(1, 14, 3) * [1, 0.15, 1])).with_color(tuple(float(c) for c in rng.uniform(0.2, 1, 3)))
        l.with_intensity(float(rng.uniform(0.2, 2))).with_start_distance(float(rng.uniform(0.5, 3))).with_end_distance(float(rng.uniform(3, 10))).with_flicker(float(rng.uniform(0, 0.6)))
        l.direction, l.normal = tuple(float(x) for x in rng.normal(0, 1, 3)), tuple(float(x) for x in rng.normal(0, 1, 3))
        l.width, l.height, l.cone_angle, l.from_linedef = float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3)), float(rng.uniform(0.1, 1.5)), bool(rng.random() < 0.3)
        for _ in range(int(rng.integers(0, 3))):
            f = ["intensity", "start_distance", "end_distance", "flicker", "width", "height", "cone_angle", "position", "color", "direction", "normal"][int(rng.integers(0, 11))]
            if f in ("position", "color", "direction", "normal"):
                v = list(getattr(l, f))
                v[int(rng.integers(0, 3))] = pick()
                setattr(l, f, tuple(v))
            else:
                setattr(l, f, pick())
        if l.direction == (0.0, 0.0, 0.0) or l.normal == (0.0, 0.0, 0.0):
            pass   # (the zero vector normalises to NaN on both sides)
        if use_lights:
            cfg.scene.add_dynamic_light(l.compile())
        if log is not None:
            log.append(("light", t, l.position, l.color, l.intensity, l.start_distance, l.end_distance, l.flicker, l.direction, l.normal, l.width, l.height, l.cone_angle, l.from_linedef))
    for _ in range(int(rng.integers(0, 3))):
        v = rng.uniform(0, 250, (3, 2)).astype(np.float32)
        uv = rng.uniform(0, 1, (3, 2)).astype(np.float32)
        if rng.random() < 0.7:
            v[int(rng.integers(0, 3)), int(rng.integers(0, 2))] = pick()
        if rng.random() < 0.5:
            uv[int(rng.integers(0, 3)), int(rng.integers(0, 2))] = pick()
        if log is not None:
            log.append(("2d", v.tolist(), uv.tolist()))
        if use_2d:
          cfg.scene.add_d2_static(api.Batch2D.new(v, np.array([[0, 1, 2]], np.uint32), uv).source(B.PixelSource.Pixel(tuple(int(c) for c in rng.integers(0, 256, 3)) + (int(rng.integers(30, 256)),))))
    base = cfg.setup
    extras = []
    for _ in range(int(rng.integers(0, 3))):
        box = [float(x) for x in rng.uniform(0, 15, 4)]
        occ = float(rng.uniform(0, 1))
        if rng.random() < 0.6:
            k = int(rng.integers(0, 5))
            if k == 4:
                occ = pick()
            else:
                box[k] = pick()
        extras.append(("occ", (box[0], box[1]), (box[2], box[3]), occ))
    if rng.random() < 0.5:
        d = [float(x) for x in rng.normal(0, 1, 3)]
        df = float(rng.uniform(0, 1))
        if rng.random() < 0.6:
            k = int(rng.integers(0, 4))
            if k == 3:
                df = pick()
            else:
                d[k] = pick()
        extras.append(("sun", tuple(d), df))
    if rng.random() < 0.4:
        a, b = [float(x) for x in rng.uniform(0, 15, 2)], [float(x) for x in rng.uniform(0, 15, 2)]
        if rng.random() < 0.6:
            a[int(rng.integers(0, 2))] = pick()
        extras.append(("line", tuple(a), tuple(b)))

    if log is not None:
        log.extend(extras)

    def setup():
        r = base()
        for e in (extras if use_extras else []):
            if e[0] == "occ":
                r.mapmini_add_occluder(e[1], e[2], e[3])
            elif e[0] == "sun":
                r.sun(e[1], e[2])
            else:
                r.mapmini_add_linedef(e[1], e[2])
        return r

    cfg.setup = setup
    return cfg


if __name__ != "__main__":
    raise SystemExit
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = []
for s in range(first, first + n):
    ref = scenes.render(build(orc, s))
    for exact in (0, 1):
        prod.lib.rxh_set_light_math_exact(exact)
        try:
            got = scenes.render(build(prod, s))
        except Exception as e:
            bad.append((s, exact, str(e)[:90]))
            continue
        finally:
            prod.lib.rxh_set_light_math_exact(0)
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
        if d.max() > 1 or (d > 0).sum() > 200:
            y, x = np.argwhere(d == d.max())[0]
            bad.append((s, exact, int((d > 1).sum()), int((d > 0).sum()), int(d.max()), (int(y), int(x)), got[y, x].tolist(), ref[y, x].tolist()))
    if (s - first) % 50 == 49:
        print(f"... {s - first + 1} seeds, {len(bad)} failures so far", flush=True)
print("lit special-value sweep seeds", first, "..", first + n - 1, "failures:", len(bad))
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
