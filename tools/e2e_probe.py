#!/usr/bin/env python3
"""Where the end-to-end time of the drop-in call goes for a large host-projected scene (default: configuration C5, 1 M triangles
at 7680x4320): host projection (Scene::project on the worker pool), the hand-over (validation + flatten into pinned memory + H2D),
render + download, and the whole call.  Median of --frames calls each.

    python tools/e2e_probe.py [--config C5] [--frames 7] [--out profiles/r03/c5_e2e_breakdown.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5")
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import rusterix_amd
    from rusterix_amd import scenes
    from run_configs import config

    prod = rusterix_amd.load()
    host = prod.lib
    rxr = rusterix_amd.rxr_abi()
    host.rxh_scene_project.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    cfg = config(prod, args.config)
    W, H = cfg.width, cfg.height
    out = np.zeros(W * H * 4, np.uint8)
    scenes.render(cfg, out)
    scenes.render(cfg, out)

    def med(fn):
        ts = []
        for _ in range(args.frames):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return round(float(np.median(ts)), 3)

    r = cfg.setup()
    res = {"config": args.config, "resolution": [W, H], "host_threads": os.environ.get("RXR_HOST_THREADS", "default"), "upload_mode": os.environ.get("RXR_UPLOAD_PIPELINE", "default")}
    res["project_ms"] = med(lambda: host.rxh_scene_project(r._h, cfg.scene._h, W, H))
    res["project_plus_handover_ms"] = med(lambda: host.rxh_rasterizer_upload(r._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h))
    ctx = host.rxh_context()
    res["render_download_ms"] = med(lambda: rxr.rxr_render_download(ctx, out.ctypes.data))
    res["whole_call_ms"] = med(lambda: scenes.render(cfg, out))
    from rusterix_amd.binding import pinned_pixels

    locked, free_locked = pinned_pixels(rxr, out.nbytes)   # page-locked pixels from the library's allocator (nothing of the heap is locked: rxr.h)
    if locked is not None:
        res["render_download_pinned_ms"] = med(lambda: rxr.rxr_render_download(ctx, locked.ctypes.data))
        res["whole_call_pinned_ms"] = med(lambda: scenes.render(cfg, locked))
        del locked
        free_locked()
    res["handover_ms"] = round(res["project_plus_handover_ms"] - res["project_ms"], 3)
    print(json.dumps(res), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
