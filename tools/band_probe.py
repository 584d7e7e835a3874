#!/usr/bin/env python3
"""What do the EMPTY tiles of a sparse frame cost?  Renders configuration C5 (the 1 M-triangle grid covers 45 % of the 8K frame's tiles; the
rows above it are empty) device-resident as the whole frame and as the band of rows that holds every non-empty tile, and prints both times.
    python tools/band_probe.py [--frames 30]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import rusterix_amd
from run_configs import config

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--config", default="C5")
a = ap.parse_args()
os.environ.setdefault("RXR_SHADER_JIT", "0")
prod = rusterix_amd.load(); host = prod.lib
rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
host.rxh_context.restype = C.c_void_p
host.rxh_rasterizer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
rxr.rxr_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
rxr.rxr_synchronize.argtypes = [C.c_void_p]
rxr.rxr_download_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
cfg = config(prod, a.config)
W, H = cfg.width, cfg.height
r = cfg.setup()
assert host.rxh_rasterizer_upload(r._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h) == 0
ctx = host.rxh_context()
img = np.zeros((H, W, 4), np.uint8)
assert rxr.rxr_render_rows(ctx, 0, H) == 0 and rxr.rxr_download_rows(ctx, img.ctypes.data_as(C.POINTER(C.c_uint8)), 0, H) == 0
hit_rows = np.nonzero((img[..., :3].max(axis=2) > 0).any(axis=1))[0]
r0, r1 = int(hit_rows.min()) // 16 * 16, min(H, (int(hit_rows.max()) // 16 + 1) * 16)
def t(row0, row1):
    for _ in range(3): rxr.rxr_render_rows(ctx, row0, row1)
    rxr.rxr_synchronize(ctx)
    t0 = time.perf_counter()
    for _ in range(a.frames): rxr.rxr_render_rows(ctx, row0, row1)
    rxr.rxr_synchronize(ctx)
    return (time.perf_counter() - t0) / a.frames * 1e3
res = {"config": a.config, "rows_with_hits": [r0, r1], "tile_rows_total": (H + 15) // 16, "tile_rows_in_band": (r1 - r0) // 16}
for k in range(2):
    res[f"whole_frame_ms_{k}"] = round(t(0, H), 4)
    res[f"band_ms_{k}"] = round(t(r0, r1), 4)
print(json.dumps(res))
