#!/usr/bin/env python3
"""End-to-end time of the drop-in call (Rasterizer::rasterize into host pixels) on a multi-device context with N members.
On a 1-GPU box the members are logical (all on device 0): that measures the overhead of the multi-device path (worker
threads, per-member uploads, strided downloads), not a speed-up.   usage: tools/e2e_multi.py [--members 1,2,4,8] [--config C4]"""
import argparse
import ctypes as C
import json
import os

os.environ.setdefault("RXR_SHADER_JIT", "0")  # (measurements name their mode: interpreted unless asked otherwise)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusterix_amd  # noqa: E402
from rusterix_amd import scenes  # noqa: E402
from tools.run_configs import config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", default="1,2,4,8")
    ap.add_argument("--config", default="C4")
    ap.add_argument("--devices", default="", help="comma-separated device ids to cycle through (default: all visible)")
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--device-projection", action="store_true")
    args = ap.parse_args()
    prod = rusterix_amd.load()
    rxr = rusterix_amd.rxr_abi()
    n_dev = rxr.rxr_device_count()
    devs = [int(x) for x in args.devices.split(",")] if args.devices else list(range(max(1, n_dev)))
    prod.lib.rxh_set_device_projection(1 if args.device_projection else 0)
    cfg = config(prod, args.config)
    W, H = cfg.width, cfg.height
    out = np.zeros(W * H * 4, np.uint8)
    if args.pinned:   # page-locked pixels from the library's allocator (nothing of the malloc heap is locked: include/rxr.h)
        from rusterix_amd.binding import pinned_pixels

        out, free_locked = pinned_pixels(rxr, W * H * 4)
        assert out is not None
    ref = None
    for n in [int(x) for x in args.members.split(",")]:
        ids = (C.c_int * n)(*[devs[i % len(devs)] for i in range(n)])
        prod.lib.rxh_set_devices(ids, n)
        ctx = prod.lib.rxh_context()
        for _ in range(3):
            scenes.render(cfg, out)
        ts = []
        t_start = time.perf_counter()
        while len(ts) < 20 or time.perf_counter() - t_start < 0.5:
            t0 = time.perf_counter()
            scenes.render(cfg, out)
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = out.copy()
        ms = float(np.median(ts)) * 1e3
        print(json.dumps(dict(config=args.config, members=n, devices=[int(i) for i in ids], pinned=args.pinned, shard=os.environ.get("RXR_MULTI_SHARD", "stripes"),
                              copy=os.environ.get("RXR_MULTI_COPY", "2d"), e2e_ms=round(ms, 4), e2e_mpix_s=round(W * H / ms / 1e3, 1), frames=len(ts),
                              identical_to_first=bool(np.array_equal(out, ref)))), flush=True)


if __name__ == "__main__":
    main()
