#!/usr/bin/env python3
"""What ONE GPU of an N-GPU node has to do per frame, measured on one GPU: a loop of rxr_render_stripes_to(first, stride=N) over
the bench frame (map scene, 3840x2160, 16 point lights), no exchange.  Per share it reports

  gpu_us     kernel time per frame from HIP events on the launch stream (every 8th frame; set-up + raster),
  wall_us    wall time per frame of the back-to-back loop, synchronized once at the end (= max(host issue, GPU)),
  issue_us   host time per call while the queue is far from full (the first frames of an idle stream),
  batch_*    the same through rxr_render_stripes_batch (one host call issues K frames), when the library has it.

The north-star's >= 6x at N = 8 leaves 139 us / 6 = 23 us per frame and GPU.

    python tools/share_probe.py [--strides 1,2,4,8] [--frames 2000] [--out profiles/r03/share8.json]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--strides", default="1,2,4,8")
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=16)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lanes", type=int, default=1,
                    help="frames in flight on the GPU: L > 1 renders frame i through member i mod L of a multi-device context whose members all sit "
                         "on GPU 0 (each has its own resident frame and scratch), on a stream of its own")
    ap.add_argument("--group-batch", action="store_true", help="hand the lane group to rxr_render_stripes_batch (lanes alternate per frame, fork / join per call)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch

    import rusterix_amd
    from rusterix_amd import distributed as D
    from rusterix_amd import scenes

    prod = rusterix_amd.load()
    host = prod.lib
    rxr = rusterix_amd.rxr_abi()
    L = max(1, args.lanes)
    if L > 1:
        host.rxh_set_devices((C.c_int * L)(*([0] * L)), L)
    else:
        host.rxh_set_device(0)
    W, H = args.width, args.height
    cfg = scenes.map_scene(prod, width=W, height=H, n_lights=args.lights)
    rast = cfg.setup()
    rc = host.rxh_rasterizer_upload(rast._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h)
    assert rc == 0, host.rxh_last_error()
    ctx = host.rxh_context()

    def check(rc_):
        if rc_ != 0:
            raise SystemExit(f"rxr call failed: {rc_} {rxr.rxr_last_error(ctx)}")

    group = ctx
    lanes = [C.c_void_p(rxr.rxr_member(group, k)) for k in range(L)] if L > 1 else [C.c_void_p(ctx)]
    ctx = lanes[0]
    streams = [torch.cuda.Stream() for _ in range(L)]
    sptrs = [C.c_void_p(st.cuda_stream) for st in streams]
    stream = streams[0]
    torch.cuda.set_stream(stream)
    sptr = sptrs[0]
    full = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    check(rxr.rxr_render_rows_to(ctx, 0, H, C.c_void_p(full.data_ptr()), sptr))
    check(rxr.rxr_synchronize(ctx))
    has_batch = hasattr(rxr, "rxr_render_stripes_batch")
    rows = []
    for stride in [int(s) for s in args.strides.split(",")]:
        spr = D.stripes_per_rank(H, stride)
        bufs = [torch.zeros((spr * D.TILE_H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2 * L)]
        ptrs = [C.c_void_p(b.data_ptr()) for b in bufs]
        first = 0

        def loop(n):
            if L == 1:
                for i in range(n):
                    rc_ = rxr.rxr_render_stripes_to(ctx, first, stride, ptrs[i & 1], sptr)
                    if rc_ != 0:
                        check(rc_)
            else:
                nb = 2 * L
                for i in range(n):
                    rc_ = rxr.rxr_render_stripes_to(lanes[i % L], first, stride, ptrs[i % nb], sptrs[i % L])
                    if rc_ != 0:
                        check(rc_)

        loop(200)
        torch.cuda.synchronize()
        for ln in lanes:
            check(rxr.rxr_synchronize(ln))
        # byte identity of the share with the full frame's stripes
        ref = torch.from_numpy(D.extract_stripes(full.cpu().numpy(), stride, first))
        for b in bufs:
            assert torch.equal(b.cpu(), ref), f"stripes of stride {stride} differ from the full frame"
        # wall: back-to-back, one synchronize at the end; median of 5 runs
        walls = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(args.frames)
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) / args.frames * 1e6)
        # host issue cost: 64 calls into an idle stream, not waited for
        issues = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(64)
            issues.append((time.perf_counter() - t0) / 64 * 1e6)
            torch.cuda.synchronize()
        # kernel time from events on every 8th frame
        check(rxr.rxr_profile_stride(ctx, 8))
        ring = args.frames // 8 + 8
        check(rxr.rxr_profile_begin(ctx, ring))
        loop(args.frames)
        su = (C.c_float * ring)()
        ru = (C.c_float * ring)()
        n = C.c_uint32(0)
        check(rxr.rxr_profile_read(ctx, su, ru, ring, C.byref(n)))
        check(rxr.rxr_profile_begin(ctx, 0))
        row = {"stride": stride, "lanes": L, "share_rows": int(min(spr * D.TILE_H, H)), "frames": args.frames,
               "gpu_setup_us": round(float(np.median(su[: n.value])), 2), "gpu_raster_us": round(float(np.median(ru[: n.value])), 2),
               "wall_us": round(float(np.median(walls)), 2), "wall_us_min_max": [round(min(walls), 2), round(max(walls), 2)],
               "issue_us": round(float(np.median(issues)), 2)}
        row["gpu_us"] = round(row["gpu_setup_us"] + row["gpu_raster_us"], 2)
        if has_batch:
            K = args.batch
            big = [torch.zeros((K, spr * D.TILE_H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2 * L)]
            frame_bytes = spr * D.TILE_H * W * 4

            # lanes alternate per CALL: batch i goes to member i mod L on stream i mod L (bench.py's sharded loop), nothing to fork or
            # join.  (The group handle alternates the lanes per frame inside one call instead and pays fork / join events per call:
            # measured with --group-batch.)
            cs = [torch.cuda.Stream() for _ in range(max(2, L))]
            csp = [C.c_void_p(st.cuda_stream) for st in cs]
            nbig = len(big)

            def bloop(n):
                for i in range(n // K):
                    if args.group_batch or L == 1:
                        rc_ = rxr.rxr_render_stripes_batch(C.c_void_p(group) if L > 1 else ctx, first, stride, K, C.c_void_p(big[i % nbig].data_ptr()), C.c_size_t(frame_bytes), csp[i & 1])
                    else:
                        rc_ = rxr.rxr_render_stripes_batch(lanes[i % L], first, stride, K, C.c_void_p(big[i % nbig].data_ptr()), C.c_size_t(frame_bytes), csp[i % L])
                    if rc_ != 0:
                        check(rc_)

            bloop(10 * K)
            torch.cuda.synchronize()
            check(rxr.rxr_synchronize(C.c_void_p(group) if L > 1 else ctx))
            for bb in big:
                for k in range(K):
                    assert torch.equal(bb[k].cpu(), ref), "batched stripes differ from the full frame"
            bw = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bloop(args.frames)
                torch.cuda.synchronize()
                bw.append((time.perf_counter() - t0) / (args.frames // K * K) * 1e6)
            row["batch_frames_per_call"] = K
            bi = []
            for _ in range(9):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bloop(4 * K)
                bi.append((time.perf_counter() - t0) / (4 * K) * 1e6)
                torch.cuda.synchronize()
            row["batch_issue_us"] = round(float(np.median(bi)), 2)
            row["batch_lanes_alternate_per"] = "frame (group handle, fork / join per call)" if args.group_batch and L > 1 else "call"
            row["batch_wall_us"] = round(float(np.median(bw)), 2)
            row["batch_wall_us_min_max"] = [round(min(bw), 2), round(max(bw), 2)]
        rows.append(row)
        print(json.dumps(row), flush=True)
    info = {"what": "per-GPU cost of one frame share on ONE MI355X (tools/share_probe.py): map scene %dx%d, %d point lights, light loop %s, "
                    "RXR_SMALL_MODE=%s, RXR_SHARE_FUSED=%s" % (W, H, args.lights, os.environ.get("RXR_LIGHT_MATH", "default"),
                                                               os.environ.get("RXR_SMALL_MODE", "default"), os.environ.get("RXR_SHARE_FUSED", "default")),
            "budget_us_for_6x_at_8": 23.0, "rows": rows}
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "w") as f:
            json.dump(info, f, indent=1)


if __name__ == "__main__":
    main()
