#!/usr/bin/env python3
"""bench.py -- Mpixels/s of one `rasterize` pass over the map scene at 3840x2160 with 16 point lights
(BASELINE.json configs[3], the configuration the metric is quoted on), on N MI355X of one node.

A "step" is one full frame: triangle set-up + binning + the tile raster kernel over every pixel,
inputs (projected batches, textures, lights) already resident in HBM, output left in HBM.  At N > 1
the frame is sharded by interleaved 16-row stripes (rank r renders stripes r, r+N, ...), the compact
per-rank stripe buffers are gathered to rank 0 with RCCL over xGMI and de-interleaved there, so one
step still produces the whole frame (strong scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_STRIDE = int(os.environ.get("RXR_BENCH_PROFILE_STRIDE", "8"))             # every 8th frame of the timed region carries HIP events around its kernels
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (FMA counted as 2)


def algorithmic_bytes(width, height, n_vertices, n_triangles, texture_bytes, n_lights):
    """framebuffer written once + projected geometry, touched textures and lights
    read once.  Per vertex 16+8+12 B (projected xyzw, uv, normal), per triangle 12+40 B (indices, Edges)."""
    return width * height * 4 + n_vertices * 36 + n_triangles * 52 + texture_bytes + n_lights * 80


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=16)
    ap.add_argument("--cpu-frames", type=int, default=2, help="frames of the same workload timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--exchange", choices=["gather", "allgather"], default="gather",
                    help="N>1: gather the stripes to rank 0 (default, what the north-star asks for) or all-gather them")
    ap.add_argument("--force-gather", action="store_true",
                    help="debug: run the stripe -> gather -> assemble path even at N=1 (never used by the driver)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import rusterix_amd
    from rusterix_amd import distributed as D
    from rusterix_amd import scenes

    prod = rusterix_amd.load()
    host = prod.lib
    rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
    host.rxh_set_device.argtypes = [C.c_int]
    host.rxh_context.restype = C.c_void_p
    host.rxh_last_error.restype = C.c_char_p
    host.rxh_rasterizer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rxr.rxr_render_stripes_to.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    rxr.rxr_render_rows_to.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    rxr.rxr_synchronize.argtypes = [C.c_void_p]
    rxr.rxr_profile_begin.argtypes = [C.c_void_p, C.c_uint32]
    rxr.rxr_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
    rxr.rxr_last_error.restype = C.c_char_p
    rxr.rxr_last_error.argtypes = [C.c_void_p]
    host.rxh_set_device(local_rank)

    W, H = args.width, args.height
    cfg = scenes.map_scene(prod, width=W, height=H, n_lights=args.lights)
    rast = cfg.setup()
    rc = host.rxh_rasterizer_upload(rast._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h)
    if rc != 0:
        raise SystemExit(f"upload failed: {rc} {host.rxh_last_error()}")
    ctx = host.rxh_context()

    def check(rc_):
        if rc_ != 0:
            raise SystemExit(f"rxr call failed: {rc_} {rxr.rxr_last_error(ctx)}")

    # one explicit (non-default) stream for everything: the raster launches, the dependency of the RCCL
    # exchange and the assemble copy.  (torch's default stream has handle 0, which the C ABI would read
    # as "use the context's own stream" -- the collective would then not be ordered behind the render.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    sptr = C.c_void_p(stream.cuda_stream)
    spr = D.stripes_per_rank(H, world)
    stripe_rows = spr * D.TILE_H
    # double-buffered outputs so that frame i+1 can render while frame i is being gathered
    NBUF = 2
    sharded = world > 1 or args.force_gather
    if not sharded:
        frames = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
    else:
        gather = D.StripeGather(H, W, world, rank, device="cuda", nbuf=NBUF, mode=args.exchange)
        frames = gather.frames  # None on ranks that do not own the frame (gather mode: only rank 0 does)

    def run(n):
        """n complete frames.  N > 1 is software-pipelined: while frame i-1 travels to rank 0 on the RCCL
        stream, frame i renders; every frame is gathered and assembled before run() returns."""
        for i in range(n):
            b = i % NBUF
            if not sharded:
                check(rxr.rxr_render_rows_to(ctx, 0, H, C.c_void_p(frames[b].data_ptr()), sptr))
            else:
                # this rank's stripes -> compact band -> RCCL gather to rank 0 over xGMI -> de-interleave
                check(rxr.rxr_render_stripes_to(ctx, rank, world, C.c_void_p(gather.band(i).data_ptr()), sptr))
                gather.exchange_begin(i)
                if i > 0:
                    gather.exchange_end(i - 1)
        if sharded and n > 0:
            gather.exchange_end(n - 1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    # kernel durations are measured live with HIP events on the launch stream, on every PROFILE_STRIDE-th frame of the
    # timed region: three event records per frame idle the GPU for 10-25 us, a tenth of this frame
    rxr.rxr_profile_stride.argtypes = [C.c_void_p, C.c_uint32]
    check(rxr.rxr_profile_stride(ctx, PROFILE_STRIDE))
    check(rxr.rxr_profile_begin(ctx, args.steps))
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch kernel durations measured with HIP events on the launch stream during the timed region
    setup_us = (C.c_float * args.steps)()
    raster_us = (C.c_float * args.steps)()
    n_prof = C.c_uint32(0)
    check(rxr.rxr_profile_read(ctx, setup_us, raster_us, args.steps, C.byref(n_prof)))
    raster_avg_us = float(np.mean(raster_us[: n_prof.value])) if n_prof.value else float("nan")
    setup_avg_us = float(np.mean(setup_us[: n_prof.value])) if n_prof.value else float("nan")

    # sanity (on the rank that owns the assembled frame): not empty, every pixel resolved
    if frames is not None:
        final = frames[(args.steps - 1) % NBUF][:H]
        assert int(final[..., 3].min().item()) == 255 and int(final[..., :3].max().item()) > 0, "benchmark frame is not a rendered frame"
        if sharded:
            # the sharded + gathered frame must be byte-identical to a single-launch frame
            direct = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
            check(rxr.rxr_render_rows_to(ctx, 0, H, C.c_void_p(direct.data_ptr()), sptr))
            torch.cuda.synchronize()
            assert torch.equal(direct, final), "gathered frame differs from the single-launch frame"

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = W * H * args.steps / dt / 1e6
        # algorithmic bytes of ONE raster launch: this rank's share of the framebuffer + the scene read once
        n_verts, n_tris, tex_bytes = scene_counts(cfg, prod)
        rows_this_rank = min(stripe_rows, H) if sharded else H
        alg = algorithmic_bytes(W, rows_this_rank, n_verts, n_tris, tex_bytes, args.lights)
        achieved = alg / (raster_avg_us * 1e-6) / 1e9
        default_workload = (W, H, args.lights, world) == (3840, 2160, 16, 1)
        out = {
            "metric": "Mpixels/s (+ ms/frame) on rasterize_map @3840x2160, 1/2/4/8 MI355X vs CPU",
            "value": round(value, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"map scene (minigame room), {W}x{H}, {args.lights} point lights, Nearest sampling, "
                            "fence cut-outs, 2D logo rectangle; UV jitter absent in the reference snapshot",
                "resolution": [W, H],
                "triangles_3d": n_tris,
                "sharding": "single GPU" if world == 1 else f"interleaved 16-row stripes over {world} GPUs + RCCL {args.exchange} to rank 0, "
                                                           "pipelined with the next frame's render",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_raster",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                # HBM bytes of one launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE with the
                # gfx950 x2 correction + WRITE_SIZE; profiles/r01/h_bench_pmc_summary.json); only known for the
                # default workload
                "traffic": MEASURED_TRAFFIC_4K["fetch_x2"] + MEASURED_TRAFFIC_4K["write"] if default_workload else None,
                "algorithmic_bytes_per_launch": int(alg),
                "kernel_avg_us": round(raster_avg_us, 2),
                "kernel_samples": int(n_prof.value),
                "setup_kernels_avg_us": round(setup_avg_us, 2),
                "measured_traffic_bytes_4k_1gpu": MEASURED_TRAFFIC_4K,
                "valu": {
                    "note": "the kernel is fp32-VALU bound, not HBM bound: algorithmic HBM traffic is ~4.5 B/pixel "
                            "(DESIGN.md section 6)",
                    "wave_instructions_per_launch_4k": MEASURED_VALU_4K,
                    "issue_cycles_per_launch_4k": VALU_ISSUE_CYCLES_4K,
                    "frac_of_simd_cycles": round(min(1.0, VALU_ISSUE_CYCLES_4K / (1024 * 2.4e3 * raster_avg_us)), 3) if default_workload else None,
                },
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(W, H, args.lights, args.cpu_frames)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# rocprofv3 PMC passes of `python3 bench.py` (3840x2160, 16 lights, 1 GPU), per k_raster launch:
# profiles/r01/h_bench_pmc_summary.json (WRITE_SIZE includes 2.4 MB of register spills: k_raster is bounded to 64 VGPRs)
MEASURED_TRAFFIC_4K = {"write": 35542191, "fetch_x2": 2248244, "source": "profiles/r01/i_bench_pmc_summary.json"}
MEASURED_VALU_4K = 156887443
# the same instructions priced with the measured issue costs of tools/microbench/valu_rates.hip (profiles/r01/valu_issue_rates.txt:
# fma / mul / add 2.6 cycles per wave64 instruction, rcp / sqrt / rsq / log / exp 8.3, the rest -- compares, selects,
# conversions, min / max at 4.3, moves and integer adds at 2.5 -- taken at 3.4): an estimate, good to about +-10 %
VALU_ISSUE_CYCLES_4K = int((21.84e6 + 31.40e6 + 37.63e6) * 2.6 + 4.94e6 * 8.3 + (156.56e6 - 90.87e6 - 4.94e6) * 3.4)


def scene_counts(cfg, api):
    n_verts = n_tris = 0
    for i in range(64):
        try:
            b = cfg.scene.projected_batch3d(3, i)  # RXR_LIST_STATIC
        except IndexError:
            break
        n_verts += b["projected_vertices"].shape[0]
        n_tris += b["clipped_indices"].shape[0]
    # textures touched by the map scene: five 64x64 + the 64x80 fence + the logo tile
    tex_bytes = 4 * (4 * 64 * 64 + 64 * 80) + 1024 * 1024 * 4
    return n_verts, n_tris, tex_bytes


def cpu_baseline(W, H, n_lights, n_frames):
    """The CPU oracle (C++ restatement of the reference algorithm, threaded over tiles like rayon) timed
    on this host's cores on the SAME workload, for a bounded number of frames."""
    from rusterix_amd import scenes
    from tests.oracle_api import load_oracle

    orc = load_oracle()
    cfg = scenes.map_scene(orc, width=W, height=H, n_lights=n_lights)
    out = np.zeros(W * H * 4, np.uint8)
    threads = os.cpu_count() or 1
    r = orc.set_threads(cfg.setup(), threads)
    r.rasterize(cfg.scene, out, W, H, cfg.tile_size, cfg.assets)  # warm-up frame (not timed)
    # bounded sample: at least `n_frames` frames and at least ~10 s of wall time, at most 30 s
    t0 = time.perf_counter()
    done = 0
    while True:
        r = orc.set_threads(cfg.setup(), threads)
        r.rasterize(cfg.scene, out, W, H, cfg.tile_size, cfg.assets)
        done += 1
        el = time.perf_counter() - t0
        if (done >= n_frames and el >= 10.0) or el >= 30.0:
            break
    n_frames = done
    dt = time.perf_counter() - t0
    return {
        "value": round(W * H * n_frames / dt / 1e6, 3),
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": "port",
        "ms_per_frame": round(dt / n_frames * 1e3, 1),
        "sample": f"{n_frames} full frames of the same workload ({W}x{H}, {n_lights} lights, tile_size 40) after one warm-up frame; "
                  "C++ restatement of the reference algorithm, std::thread pool over tiles, includes Scene::project",
    }


if __name__ == "__main__":
    main()
