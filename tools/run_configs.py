#!/usr/bin/env python3
"""Runs the BASELINE.json configurations at full size on one GPU: kernel-only and end-to-end timings
through the C ABI, plus a full-size comparison with the CPU oracle (the checker; never the thing
measured).  Writes one JSON line per configuration; used to fill the table in BASELINE.md.

    python tools/run_configs.py [--configs C2,C3,C4,C5] [--frames 30] [--oracle C2,C3,C4]
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

import rusterix_amd  # noqa: E402
from rusterix_amd import binding as B  # noqa: E402
from rusterix_amd import scenes  # noqa: E402


def config(api, name):
    if name == "C1":
        return scenes.cube_scene(api, width=800, height=600, tile_size=200, textured=True, distance=3.0)
    if name == "C2":
        return scenes.teapot_scene(api, width=1920, height=1080)
    if name == "C3":
        return scenes.map_scene(api, width=1920, height=1080, n_lights=1)
    if name == "C4":
        return scenes.map_scene(api, width=3840, height=2160, n_lights=16)
    if name == "C5":
        return scenes.box_grid_scene(api, n=289, width=7680, height=4320)
    if name == "C5_4k":  # ... and at 3840 x 2160: 31 triangles per 16 x 16 tile on average, hundreds at the horizon
        return scenes.box_grid_scene(api, n=289, width=3840, height=2160)
    if name == "C5_16k":  # the 1 M-triangle grid at 15360 x 8640 (132.7 Mpixel: beyond k_blockscan's slot budget, a 531 MB frame)
        return scenes.box_grid_scene(api, n=289, width=15360, height=8640)
    if name == "C5s":  # reduced C5 for a full oracle comparison
        return scenes.box_grid_scene(api, n=96, width=1920, height=1080)
    if name == "C5shader":  # configs[4] as named: Linear sampling + per-batch shader
        return scenes.box_grid_scene(api, n=289, width=7680, height=4320, shader=True)
    if name == "C5s_shader":
        return scenes.box_grid_scene(api, n=96, width=1920, height=1080, shader=True)
    if name.startswith("C5s_vm:"):  # interpreter cost probes: C5s_shader's scene with a synthetic program (empty | u40 | b20 | g20 | m20)
        kind = name.split(":", 1)[1]
        body = {"empty": [],
                "u40": ["Color"] + ["Abs"] * 40 + ["SetColor"],
                "b20": ["Color"] + [("Push", 1.0), "Mul"] * 20 + ["SetColor"],
                "g20": ["Color", "SetColor"] * 20,
                "m20": ["Color"] + ["Color", "Mul"] * 20 + ["SetColor"]}[kind]
        scenes.box_grid_shader = lambda: B.Program([body])
        return scenes.box_grid_scene(api, n=96, width=1920, height=1080, shader=True)
    if name.startswith("near:"):  # the box lattice seen from among its boxes: binned scenes of LARGE triangles (row mode's other end); near:<distance>
        return scenes.box_grid_scene(api, n=48, width=1920, height=1080, distance=float(name.split(":", 1)[1]))
    if name.startswith("C5s_panes:"):  # ... with k small translucent panes scattered over the lattice in ONE opacity batch (particles, glass)
        k = int(name.split(":", 1)[1])
        cfg = scenes.box_grid_scene(api, n=96, width=1920, height=1080)
        rng = np.random.default_rng(77)
        tmpl = api.Batch3D.from_box(0.0, 0.0, 0.0, 0.3, 0.3, 0.01)
        tv, ti, tuv, _ = tmpl.geometry()
        vs, is_, uvs = [], [], []
        for j in range(k):
            v = tv.copy()
            v[:, 0] += np.float32(rng.uniform(0.0, 19.0)); v[:, 1] += np.float32(rng.uniform(0.5, 1.5)); v[:, 2] += np.float32(rng.uniform(0.0, 19.0))
            vs.append(v); is_.append(ti + np.uint32(24 * j)); uvs.append(tuv)
        chunk = cfg.scene.add_chunk()
        chunk.add_batch3d_opacity(api.Batch3D.new(np.concatenate(vs), np.concatenate(is_), np.concatenate(uvs)).with_computed_normals()
                                  .source(B.PixelSource.Pixel((90, 160, 250, 120))))
        return cfg
    if name == "C5s_pane":  # the reduced box grid + ONE small translucent pane in a chunk's opacity list: an opacity pass in a binned frame
        cfg = scenes.box_grid_scene(api, n=96, width=1920, height=1080)
        chunk = cfg.scene.add_chunk()
        chunk.add_batch3d_opacity(api.Batch3D.from_box(9.0, 1.0, 9.0, 1.2, 1.2, 0.02).with_computed_normals()
                                  .source(B.PixelSource.Pixel((90, 160, 250, 120))).profile_id(10))
        return cfg
    if name.startswith("C5s_lit:"):  # the reduced box grid under k point lights spread over the lattice (binned + the light loop: k_raster_rows_rl)
        k = int(name.split(":", 1)[1])
        cfg = scenes.box_grid_scene(api, n=96, width=1920, height=1080)
        rng = np.random.default_rng(11)
        cfg.scene.lights([B.Light(B.LIGHT_POINT).with_position((float(rng.uniform(0, 19.2)), 1.2, float(rng.uniform(0, 19.2))))
                          .with_color((1.0, 0.9, 0.7)).with_intensity(1.5).with_start_distance(1.0).with_end_distance(6.0).compile() for _ in range(k)])
        return cfg
    if name.startswith("many_batches:"):  # the reduced box grid's 9216 boxes as k x k batches of (96 / k)^2 boxes each (k = 96: one box per batch)
        k = int(name.split(":", 1)[1])
        cfg = scenes.box_grid_scene(api, n=96, width=1920, height=1080)
        rng = np.random.default_rng(200)
        tmpl = api.Batch3D.from_box(0.0, 0.0, 0.0, 0.16, 0.16, 0.16)
        tv, ti, tuv, _ = tmpl.geometry()
        ys = rng.random((96, 96)).astype(np.float32) * np.float32(0.4)
        scene = api.Scene.empty()
        per = 96 // k
        for bz in range(k):
            for bx in range(k):
                vs, is_, uvs = [], [], []
                for j in range(per):
                    for i in range(per):
                        v = tv.copy()
                        v[:, 0] += np.float32((bx * per + i) * 0.2)
                        v[:, 1] += ys[bz * per + j, bx * per + i]
                        v[:, 2] += np.float32((bz * per + j) * 0.2)
                        vs.append(v)
                        is_.append(ti + np.uint32(24 * (j * per + i)))
                        uvs.append(tuv)
                b = api.Batch3D.new(np.concatenate(vs), np.concatenate(is_), np.concatenate(uvs))
                scene.add_d3_static(b.source(B.PixelSource.StaticTileIndex((bz + bx) % 16)).repeat_mode(B.REPEAT_REPEAT_XY).with_computed_normals())
        cfg.scene = scene
        return cfg
    if name.startswith("C5s_shader_cutout:"):  # ... with the per-batch program on every batch as well (the compiled kernel's cut variant)
        return scenes.box_grid_scene(api, n=96, width=1920, height=1080, shader=True, cutout_every=int(name.split(":", 1)[1]))
    if name.startswith("C5s_cutout:"):  # the reduced box grid with every k-th batch textured with holes (a fence): cut-out candidates in binned rounds
        return scenes.box_grid_scene(api, n=96, width=1920, height=1080, cutout_every=int(name.split(":", 1)[1]))
    if name == "D2":  # 2D tile map: 60 x 34 textured / translucent rectangles + overlays + lines, render_2d mode
        return scenes.tile_map_2d_scene(api, width=1920, height=1080, nx=60, ny=34)
    raise SystemExit(f"unknown config {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3,C4,C5s,C5")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--oracle", default="C2,C3,C4,C5s")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--jit", choices=["0", "1"], default=None,
                    help="RXR_SHADER_JIT for this run: 1 = program sets compiled at run time (the caller waits), 0 = interpreted; default: the environment")
    ap.add_argument("--no-e2e", action="store_true",
                    help="profiling runs: skip the end-to-end leg (its pipelined download rasters the frame in four bands: four shorter raster launches "
                         "per call would mix into the per-kernel averages of the device-resident loop)")
    ap.add_argument("--device-projection", action="store_true",
                    help="N1: clip_and_project + Edges on the GPU (geometry registered once, matrices per frame)")
    args = ap.parse_args()
    # measurements name their mode: interpreted unless --jit 1 (the library's own default, a background compilation that switches over
    # in mid-run, would make the numbers depend on timing)
    os.environ["RXR_SHADER_JIT"] = args.jit if args.jit is not None else os.environ.get("RXR_SHADER_JIT", "0")
    if os.environ["RXR_SHADER_JIT"] not in ("0", "1"):
        os.environ["RXR_SHADER_JIT"] = "0"

    prod = rusterix_amd.load()
    host = prod.lib
    rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
    host.rxh_context.restype = C.c_void_p
    host.rxh_last_error.restype = C.c_char_p
    host.rxh_rasterizer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rxr.rxr_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    rxr.rxr_synchronize.argtypes = [C.c_void_p]
    rxr.rxr_download_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
    rxr.rxr_profile_begin.argtypes = [C.c_void_p, C.c_uint32]
    rxr.rxr_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]

    class Stats(C.Structure):
        _fields_ = [("setup_us", C.c_float), ("raster_us", C.c_float), ("total_us", C.c_float), ("n_triangles3d", C.c_uint32),
                    ("n_triangles2d", C.c_uint32), ("n_bin_entries", C.c_uint32), ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32)]

    rxr.rxr_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    host.rxh_set_device_projection.argtypes = [C.c_int]
    host.rxh_set_device_projection(1 if args.device_projection else 0)

    for name in args.configs.split(","):
        t0 = time.perf_counter()
        cfg = config(prod, name)
        t_build = time.perf_counter() - t0
        W, H = cfg.width, cfg.height
        out = np.zeros((H, W, 4), np.uint8)
        # end-to-end: Rasterizer::setup(..).rasterize(..) = host projection + upload + kernels + download
        if args.no_e2e:
            os.environ["RXR_NO_DOWNLOAD_PIPELINE"] = "1"  # (the warm-up calls below render the frame in one launch as well)
        for _ in range(3):  # warm-up (uploads the textures; the page-locked pools of the host mirror and the library reach their size)
            scenes.render(cfg, out.reshape(-1))
        e2e = []
        for _ in range(0 if args.no_e2e else max(5, args.frames // 3)):
            t0 = time.perf_counter()
            scenes.render(cfg, out.reshape(-1))
            e2e.append(time.perf_counter() - t0)
        # device-resident: upload once, then kernels only
        r = cfg.setup()
        t0 = time.perf_counter()
        rc = host.rxh_rasterizer_upload(r._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h)
        t_upload = time.perf_counter() - t0
        assert rc == 0, host.rxh_last_error()
        ctx = host.rxh_context()
        for _ in range(3):
            rxr.rxr_render_rows(ctx, 0, H)
        rxr.rxr_synchronize(ctx)
        rxr.rxr_profile_begin(ctx, args.frames)
        t0 = time.perf_counter()
        for _ in range(args.frames):
            rxr.rxr_render_rows(ctx, 0, H)
        rxr.rxr_synchronize(ctx)
        t_loop = (time.perf_counter() - t0) / args.frames
        su = (C.c_float * args.frames)()
        ru = (C.c_float * args.frames)()
        n = C.c_uint32()
        rxr.rxr_profile_read(ctx, su, ru, args.frames, C.byref(n))
        # the same loop without per-kernel event records (what a caller that does not profile gets)
        rxr.rxr_profile_begin(ctx, 0)
        for _ in range(3):
            rxr.rxr_render_rows(ctx, 0, H)
        rxr.rxr_synchronize(ctx)
        t0 = time.perf_counter()
        for _ in range(args.frames):
            rxr.rxr_render_rows(ctx, 0, H)
        rxr.rxr_synchronize(ctx)
        t_loop_untimed = (time.perf_counter() - t0) / args.frames
        st = Stats()
        rxr.rxr_get_stats(ctx, C.byref(st))
        light_math = os.environ.get("RXR_LIGHT_MATH", "exact" if host.rxh_get_light_math_exact() else "relaxed")
        rec = dict(config=name, scene=cfg.name, shader_jit=os.environ.get("RXR_SHADER_JIT", "0"), light_math=light_math, device_projection=bool(args.device_projection), resolution=[W, H], triangles_3d=st.n_triangles3d, bin_entries=st.n_bin_entries,
                   scene_build_s=round(t_build, 2), upload_ms=round(t_upload * 1e3, 2),
                   setup_kernels_us=round(float(np.median(su[: n.value])), 1), raster_kernel_us=round(float(np.median(ru[: n.value])), 1),
                   frame_ms_device_resident=round(t_loop * 1e3, 4), mpix_per_s_device_resident=round(W * H / t_loop / 1e6, 1),
                   frame_ms_device_resident_no_events=round(t_loop_untimed * 1e3, 4),
                   frame_ms_end_to_end=round(float(np.median(e2e)) * 1e3, 3) if e2e else None,
                   mpix_per_s_end_to_end=round(W * H / float(np.median(e2e)) / 1e6, 1) if e2e else None)
        if name in args.oracle.split(","):
            from tests.oracle_api import load_oracle

            orc = load_oracle()
            ocfg = config(orc, name)
            ref = np.zeros((H, W, 4), np.uint8)
            threads = args.threads or os.cpu_count()
            t0 = time.perf_counter()
            orc.set_threads(ocfg.setup(), threads).rasterize(ocfg.scene, ref.reshape(-1), W, H, ocfg.tile_size, ocfg.assets)
            t_cpu = time.perf_counter() - t0
            diff = np.abs(out.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
            rec.update(cpu_port_ms=round(t_cpu * 1e3, 1), cpu_port_mpix_per_s=round(W * H / t_cpu / 1e6, 2), cpu_threads=threads,
                       parity=dict(pixels=int(W * H), differing=int((diff > 0).sum()), off_by_more_than_1=int((diff > 1).sum()),
                                   max_abs_diff=int(diff.max())))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
