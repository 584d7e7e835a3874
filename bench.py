#!/usr/bin/env python3
"""bench.py -- Mpixels/s of one `rasterize` pass over the map scene at 3840x2160 with 16 point lights
(BASELINE.json configs[3], the configuration the metric is quoted on), on N MI355X of one node.

A "step" is one full frame: triangle set-up (+ binning for larger scenes) + the tile raster kernel over every pixel,
inputs (projected batches, textures, lights) already resident in HBM, output left in HBM.  At N > 1 the frame is sharded
by interleaved 16-row stripes (rank r renders stripes r, r+N, ...), the compact per-rank stripe buffers are gathered to
rank 0 with RCCL over xGMI and de-interleaved there, so one step still produces the whole frame (strong scaling).  A rank's
share of a frame is tens of microseconds of GPU work, so at N > 1 every rank keeps --lanes frames in flight on its GPU (member
contexts of one device, rxr_render_stripes_batch) and moves --bucket frames per collective; `value` is measured on the exchange
BASELINE.json names (gather to rank 0), and the SAME run also times the variants that lift that exchange's link bound (rotating
root, several communicators) under `exchange_variants`, plus every rank's render-only and exchange-only time (`per_rank`).

Timing: W untimed warm-up steps, then batches of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
on both sides and timed on its own (max over ranks); batches are repeated until at least MIN_TIMED_S seconds have been
timed (a 4K frame takes 0.2 ms: K = 20 steps alone would be a 4 ms measurement) and the MEDIAN batch is reported.

Launch: `python bench.py --gpus N` starts its own N ranks (child processes, before anything touches a GPU) unless a
launcher (torchrun: RANK / WORLD_SIZE in the environment) already did; the number of ranks observed must equal --gpus.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline` and, beside the
device-resident `value`, the end-to-end time of the drop-in call (`e2e_ms`: host projection + upload + kernels + download
into the caller's pixels; at N > 1 through a multi-device context, every GPU downloading its own stripes).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# every PROFILE_STRIDE-th frame of the timed region has its kernels timed: a (start, stop) HIP event pair per DISPATCH (hipExtLaunchKernel:
# the dispatch's own begin / end timestamps, what rocprofv3's kernel trace reports), no event record on the stream -- rounds 1-3 recorded
# events between the launches, which idled the GPU and counted launch latency as kernel time (set-up 8.6 us against 5.3 in the trace)
PROFILE_STRIDE = int(os.environ.get("RXR_BENCH_PROFILE_STRIDE", "4"))
MIN_TIMED_S = float(os.environ.get("RXR_BENCH_MIN_TIMED_S", "1.0"))
MAX_BATCHES = 5000
# Rehearsal of the N > 1 code path on a box with ONE GPU (tests/test_gpu_bench_rehearsal.py): every rank uses GPU 0 and the exchange
# goes through gloo with host staging, because RCCL refuses two ranks on one device.  Everything else -- rank bookkeeping, stripes,
# pipelining indices, the byte-identity check, the multi-device end-to-end leg, the JSON line -- is the code the driver runs.  The
# line says "rehearsal": true and its numbers mean nothing.
REHEARSAL = os.environ.get("RXR_BENCH_REHEARSAL") == "1"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (FMA counted as 2)
N_SIMDS = 1024                 # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md: peak engine clock
VALU_ISSUE_CYCLES = 2          # nominal: one wave64 VALU instruction per SIMD every 2 cycles (the measured costs are higher: DESIGN.md section 6)
XGMI_LINK_GBS = (64.0, 77.0, 153.0)  # per direction and link: what RCCL point-to-point sustains on MI300-class parts (low, high), and the link's peak

# rocprofv3 passes of this same command (3840x2160, 16 lights, 1 GPU), per raster-kernel launch, for the two light-loop modes.
# NOT measured in this run: the PMC passes need the profiler (tools/profile_bench.sh).  The numbers are READ from the committed
# summary -- never copied into this file -- together with the hash of the kernel source they were taken on, so that a line
# quoting a profile of an older kernel says so (`from_profiles.stale`; tests/test_bench_profiles.py).
PROFILE_SUMMARY = "profiles/r04/bench_pmc_summary.json"
PROFILE_KERNEL_STATS = "profiles/r04/bench_kernel_stats.csv"
PROFILE_BENCH_LINE = "profiles/r04/bench_line.json"   # the line of a plain (unprofiled) run of the build the summary was taken on
# rxr_kernels.hip and everything it includes (the advisor's round-3 finding: rxr_vm.h and rxr_project.h were missing)
KERNEL_SOURCES = ["rusterix_amd/csrc/rxr_kernels.hip", "rusterix_amd/csrc/rxr_device.h", "rusterix_amd/csrc/rxr_exact_math.h", "rusterix_amd/csrc/rxr_project.h",
                  "rusterix_amd/csrc/rxr_vm.h", "rusterix_amd/csrc/rxr_launch.h", "include/rxr.h", "include/rusterix_vek.hpp"]


def kernel_sources_sha():
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def load_profiles(path=None):
    """{"relaxed": {...}, "exact": {...}} from the committed rocprofv3 summary (k_raster_rl / k_raster of the same passes: bench.py
    times the other mode as well), or {} when the file is missing"""
    path = os.path.join(ROOT, path or PROFILE_SUMMARY)
    if not os.path.exists(path):
        return {}
    d = json.load(open(path))
    out = {}
    # average duration per kernel in the rocprofv3 --kernel-trace --stats pass of the same command
    traced = {}
    stats = os.path.join(ROOT, PROFILE_KERNEL_STATS)
    if os.path.exists(stats):
        import csv

        for row in csv.DictReader(open(stats)):
            try:
                traced[row["Name"].split("(")[0]] = float(row["AverageNs"]) / 1e3
            except (KeyError, ValueError):
                pass
    for mode, kernel in (("relaxed", "k_raster_rl"), ("exact", "k_raster")):
        k = d.get("kernels", {}).get(kernel)
        if not k or "write_bytes" not in k or "fetch_bytes_x2_gfx950" not in k or "SQ_INSTS_VALU" not in k:
            continue
        out[mode] = {
            "source": f"{PROFILE_SUMMARY} (tools/profile_bench.sh; rocprofv3 --pmc, separate passes; kernel {kernel})",
            "write_bytes": int(round(k["write_bytes"])),          # WRITE_SIZE
            "fetch_bytes_x2": int(round(k["fetch_bytes_x2_gfx950"])),  # FETCH_SIZE with the gfx950 x2 correction
            "valu_wave_instructions": int(round(k["SQ_INSTS_VALU"])),
            # fp32 flops of one launch: (add + mul + 2 fma) wave-instructions x 64 lanes x the measured lane utilisation
            "fp32_wave_instructions": {c: int(round(k.get("SQ_INSTS_VALU_" + c + "_F32", 0))) for c in ("ADD", "MUL", "FMA")},
            "valu_lane_utilisation": round(k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64.0), 4) if k.get("SQ_ACTIVE_INST_VALU") else None,
            "traced_kernel_avg_us": round(traced[kernel], 2) if kernel in traced else None,
            "traced_setup_avg_us": round(traced["k_setup3d"], 2) if "k_setup3d" in traced else None,
            "kernel_sources_sha": d.get("kernel_sources_sha"),
        }
    return out


PROFILES = load_profiles()


def algorithmic_bytes(width, height, n_vertices, n_triangles, texture_bytes, n_lights):
    """framebuffer written once + projected geometry, touched textures and lights
    read once.  Per vertex 16+8+12 B (projected xyzw, uv, normal), per triangle 12+40 B (indices, Edges)."""
    return width * height * 4 + n_vertices * 36 + n_triangles * 52 + texture_bytes + n_lights * 80


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=16)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--exchange", choices=["gather", "allgather", "rotate"], default="gather",
                    help="N>1: gather the stripes to rank 0 (default, what the north-star asks for), all-gather them, or gather frame i "
                         "to rank i mod N (every frame still whole on one GPU, but no single GPU's links carry every frame)")
    ap.add_argument("--comms", type=int, default=1,
                    help="N > 1: communicators over all ranks, frame i's exchange on communicator i mod comms (with --exchange rotate the gathers of "
                         "consecutive frames go to different roots over disjoint xGMI links and can overlap); default 1")
    ap.add_argument("--depth", type=int, default=1,
                    help="N > 1: frames between the start of a frame's exchange and the wait for it (depth + 1 stripe / frame buffers); default 1")
    ap.add_argument("--lanes", type=int, default=0,
                    help="frames in flight on each GPU (member contexts on one device, each with its own resident frame and scratch, frame k on lane "
                         "k mod L): the tail of one share's launches runs under the head of the next.  Default: 2 at N > 1, 1 at N = 1")
    ap.add_argument("--bucket", type=int, default=0,
                    help="N > 1: frames per exchange (one rxr_render_stripes_batch call and one collective per bucket); default 4")
    ap.add_argument("--no-in-flight", action="store_true",
                    help="N = 1: skip the two-frames-in-flight leg (profiling runs: its overlapped launches would mix into the per-kernel averages)")
    ap.add_argument("--no-variants", action="store_true", help="N > 1: skip the exchange variants and the per-rank render-only / exchange-only legs")
    ap.add_argument("--force-gather", action="store_true",
                    help="debug: run the stripe -> gather -> assemble path even at N=1 (never used by the driver)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process never touches a GPU (it
    only counts devices) and never execs: it starts children and exits with their status."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not REHEARSAL:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if REHEARSAL else str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                st = p.poll()
                if st is None:
                    continue
                procs.remove(p)
                if st != 0:
                    rc = rc or st
                    for q in procs:  # one rank failed: the others would wait for it forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    raise SystemExit(rc)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) were launched (WORLD_SIZE={world})")

    import datetime

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (a variant that fails on one rank only would leave the others waiting in a collective: bounded)
        tmo = datetime.timedelta(seconds=int(os.environ.get("RXR_BENCH_PG_TIMEOUT_S", "300")))
        if REHEARSAL:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)
    ctl = "cpu" if REHEARSAL else "cuda"  # where the control-plane tensors (timings, batch count) live

    import rusterix_amd
    from rusterix_amd import distributed as D
    from rusterix_amd import scenes

    prod = rusterix_amd.load()
    host = prod.lib
    rxr = rusterix_amd.rxr_abi()
    sharded = world > 1 or args.force_gather
    lanes = args.lanes if args.lanes > 0 else (2 if sharded else 1)
    bucket = args.bucket if args.bucket > 0 else 4
    if not sharded:
        lanes = 1  # (the N = 1 `value` is the serial loop of earlier rounds: one frame at a time; two frames in flight are reported beside it)

    def use_lanes(n):
        """the process-wide context of the host mirror: a plain context on this rank's GPU, or n member contexts on it"""
        if n > 1:
            host.rxh_set_devices((C.c_int * n)(*([local_rank] * n)), n)
        else:
            host.rxh_set_device(local_rank)

    use_lanes(lanes)
    W, H = args.width, args.height
    cfg = scenes.map_scene(prod, width=W, height=H, n_lights=args.lights)
    rast = cfg.setup()

    def upload():
        rc_ = host.rxh_rasterizer_upload(rast._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h)
        if rc_ != 0:
            raise SystemExit(f"upload failed: {rc_} {host.rxh_last_error()}")
        return host.rxh_context()

    ctx = upload()           # plain context, or the handle of the lanes
    m0 = C.c_void_p(rxr.rxr_member(ctx, 0))  # the member that renders whole frames (identity check) and carries the kernel timing

    def check(rc_):
        if rc_ != 0:
            raise SystemExit(f"rxr call failed: {rc_} {rxr.rxr_last_error(ctx)}")

    # explicit (non-default) streams for everything: the raster launches, the dependency of the RCCL exchange and the
    # assemble copy.  (torch's default stream has handle 0, which the C ABI would read as "use the context's own
    # stream" -- the collective would then not be ordered behind the render.)  Sharded runs use one stream per lane: bucket j is
    # rendered by lane (member context) j mod L on stream j mod L, one rxr_render_stripes_batch call for its frames, and its exchange
    # is queued on the same stream -- so the buckets of different lanes overlap on the GPU (the tail of one share's launches under
    # the head of another's) without a single cross-stream wait between a render and its exchange.  (Handing the whole lane group
    # to the batch call alternates the lanes per FRAME instead; its fork / join events cost ~30 us per call on this runtime:
    # tools/share_probe.py, profiles/r03/.)
    cstreams = [torch.cuda.Stream() for _ in range(lanes if sharded else 1)]
    members = [C.c_void_p(rxr.rxr_member(ctx, k)) for k in range(lanes)]
    stream = cstreams[0]
    torch.cuda.set_stream(stream)
    assert all(st.cuda_stream != 0 for st in cstreams)
    sptr = C.c_void_p(stream.cuda_stream)
    spr = D.stripes_per_rank(H, world)
    stripe_rows = spr * D.TILE_H
    share_bytes = stripe_rows * W * 4
    depth = max(1, args.depth)
    NBUF = depth + 1
    frames = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(NBUF)] if not sharded else None

    # every communicator any configuration of this run uses, created once, in the same order on every rank
    vk = min(world, 4)
    n_groups = max(max(1, args.comms), 1 if args.no_variants else vk)
    groups = [None] if world == 1 or n_groups == 1 else [dist.new_group(list(range(world))) for _ in range(n_groups)]

    class Pipeline:
        """One exchange configuration: --lanes frames in flight per GPU, `bk` frames per collective, `dp` exchanges between
        begin and end, exchange i on communicator i mod `cm`."""

        def __init__(self, mode, cm, dp, bk):
            self.mode, self.cm, self.dp, self.bk = mode, cm, dp, bk
            self.gather = D.StripeGather(H, W, world, rank, device="cuda", nbuf=dp + 1, mode=mode, host_staged=REHEARSAL, groups=groups[:max(1, cm)], bucket=bk)
            self.free = [torch.cuda.Event() for _ in range(dp + 1)]  # buffer b's last exchange has been consumed
            for e in self.free:
                e.record(stream)
            self.last = None  # (exchange index, frames in it) of the last bucket run() assembled

        def render(self, j, k, st):
            # this rank's stripes of k frames -> compact bands (frame f of the bucket at band(j)[f]), on lane j mod L
            check(rxr.rxr_render_stripes_batch(members[j % lanes], rank, world, k, C.c_void_p(self.gather.band_ptr(j, 0)), C.c_size_t(share_bytes),
                                               C.c_void_p(st.cuda_stream)))

        def end(self, i, st):
            self.gather.exchange_end(i)
            self.free[i % (self.dp + 1)].record(st)

        def run(self, n, do_render=True, do_exchange=True):
            """n complete frames in buckets of bk: render bucket j -> RCCL gather over xGMI -> de-interleave on the root, software
            pipelined (bucket j renders while bucket j - dp travels); every frame is assembled before run() returns."""
            nb = (n + self.bk - 1) // self.bk
            for j in range(nb):
                k = min(self.bk, n - j * self.bk)
                st = cstreams[j % len(cstreams)]
                with torch.cuda.stream(st):
                    st.wait_event(self.free[j % (self.dp + 1)])  # the bands / gather targets of this buffer are free again
                    if do_render:
                        self.render(j, k, st)
                    if do_exchange:
                        self.gather.exchange_begin(j)
                        if j >= self.dp:
                            self.end(j - self.dp, st)
            if do_exchange:
                with torch.cuda.stream(stream):
                    for j in range(max(0, nb - self.dp), nb):
                        self.end(j, stream)
                self.last = (nb - 1, n - (nb - 1) * self.bk)

        def describe(self):
            return (f"interleaved 16-row stripes over {world} GPUs (one process per GPU, {lanes} frame(s) in flight per GPU) + RCCL {self.mode} "
                    + ("to rank (exchange mod N)" if self.mode == "rotate" else "to rank 0")
                    + f" over xGMI, {self.bk} frame(s) per collective, pipelined with the next renders ({max(1, self.cm)} communicator(s), {self.dp} exchange(s) in flight)")

    pipe = Pipeline(args.exchange, max(1, args.comms), depth, bucket) if sharded else None

    def run(n):
        if sharded:
            pipe.run(n)
        else:
            for i in range(n):
                check(rxr.rxr_render_rows_to(ctx, 0, H, C.c_void_p(frames[i % NBUF].data_ptr()), sptr))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        """n steps of fn between two fences; seconds, max over ranks"""
        fence()
        t0 = time.perf_counter()
        fn(n)
        fence()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    def timed_batch():
        return timed(run, args.steps)

    def agree(n):
        """rank 0's batch count on every rank"""
        if world > 1:
            nb_ = torch.tensor([n], dtype=torch.int64, device=ctl)
            dist.broadcast(nb_, src=0)
            n = int(nb_.item())
        return n

    run(args.warmup)
    fence()
    check(rxr.rxr_synchronize(ctx))  # (every warm-up frame complete: no list overflowed, no program faulted)
    # kernel durations are measured live with HIP events on the launch stream, on every PROFILE_STRIDE-th frame of the
    # timed region: three event records per frame idle the GPU for 10-25 us, a tenth of this frame
    first = timed_batch()  # sizes the run (timed like the others, not discarded)
    n_batches = agree(int(min(MAX_BATCHES, max(1, np.ceil(1.25 * MIN_TIMED_S / max(first, 1e-6))))))
    ring = min(65536, n_batches * args.steps // PROFILE_STRIDE + args.steps + 8)
    check(rxr.rxr_profile_stride(ctx, PROFILE_STRIDE))
    check(rxr.rxr_profile_begin(ctx, ring))
    batch_s = [timed_batch() for _ in range(n_batches)]
    # the asynchronous loop above never asked whether its frames were complete: do it now (sticky status, rxr.h)
    check(rxr.rxr_synchronize(ctx))
    dt = float(np.median(batch_s))

    # per-launch kernel durations measured with HIP events on the launch stream during the timed region (lanes: member 0's launches)
    setup_us = (C.c_float * ring)()
    raster_us = (C.c_float * ring)()
    n_prof = C.c_uint32(0)
    check(rxr.rxr_profile_read(ctx, setup_us, raster_us, ring, C.byref(n_prof)))
    check(rxr.rxr_profile_begin(ctx, 0))
    raster_avg_us = float(np.mean(raster_us[: n_prof.value])) if n_prof.value else float("nan")
    setup_avg_us = float(np.mean(setup_us[: n_prof.value])) if n_prof.value else float("nan")

    def check_assembled(p):
        """on the rank that owns the last assembled frame: not empty, every pixel resolved, byte-identical to a single-launch frame
        """
        i_last, k_last = p.last
        if p.gather.frames is None or (p.mode == "rotate" and rank != p.gather.root_of(i_last)):
            return
        fr = p.gather.frames[i_last % (p.dp + 1)]
        final = (fr if p.bk == 1 else fr[k_last - 1])[:H]
        assert int(final[..., 3].min().item()) == 255 and int(final[..., :3].max().item()) > 0, "benchmark frame is not a rendered frame"
        direct = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
        check(rxr.rxr_render_rows_to(m0, 0, H, C.c_void_p(direct.data_ptr()), sptr))
        check(rxr.rxr_synchronize(ctx))
        torch.cuda.synchronize()
        assert torch.equal(direct, final), "gathered frame differs from the single-launch frame"

    if sharded:
        check_assembled(pipe)
    else:
        final = frames[(args.steps - 1) % NBUF][:H]
        assert int(final[..., 3].min().item()) == 255 and int(final[..., :3].max().item()) > 0, "benchmark frame is not a rendered frame"

    def measure(fn, t_min=min(0.3, MIN_TIMED_S), n_max=200):
        """median seconds per `steps` steps of fn over batches that add up to >= t_min seconds (the count agreed on by all ranks)"""
        fn(args.warmup)
        t_first = timed(fn, args.steps)
        nb_ = agree(int(min(n_max, max(2, np.ceil(t_min / max(t_first, 1e-6))))))
        ts = [timed(fn, args.steps) for _ in range(nb_)]
        return float(np.median(ts)), nb_

    # N > 1, the SAME run: (a) what each rank spends on rendering alone and on the exchange alone (max-over-ranks timing cannot say
    # which of the two bounds the curve), (b) the exchange variants that lift the fixed root's link bound.  Never `value`.
    per_rank = None
    variants = None
    if sharded and not args.no_variants:
        def local_time(fn, n_rep=5):
            fn(args.warmup)
            ts = []
            for _ in range(n_rep):
                fence()
                t0 = time.perf_counter()
                fn(args.steps)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                fence()
            return float(np.median(ts)) / args.steps * 1e3

        def all_ranks(x):
            if world == 1:
                return [round(x, 4)]
            t = torch.tensor([x], dtype=torch.float64, device=ctl)
            out_ = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(out_, t)
            return [round(float(o.item()), 4) for o in out_]

        # the WHOLE frame on each rank's own GPU, same run: one frame at a time on one lane (what the N = 1 `value` times) and alternating
        # over this rank's lanes (the intra-GPU overlap the N > 1 pipeline also enjoys) -- scaling ratios against either figure can be
        # formed from this one line (the advisor's round-3 finding: N > 1 with lanes against a serial N = 1 credits sharding with overlap)
        whole = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")

        def whole_serial(n):
            for _ in range(n):
                check(rxr.rxr_render_rows_to(members[0], 0, H, C.c_void_p(whole.data_ptr()), sptr))

        def whole_lanes(n):
            for i in range(n):
                check(rxr.rxr_render_rows_to(members[i % lanes], 0, H, C.c_void_p(whole.data_ptr()), C.c_void_p(cstreams[i % len(cstreams)].cuda_stream)))

        per_rank = {
            "render_only_ms": all_ranks(local_time(lambda n: pipe.run(n, do_exchange=False))),
            "exchange_only_ms": all_ranks(local_time(lambda n: pipe.run(n, do_render=False))),
            "whole_frame_serial_ms": all_ranks(local_time(whole_serial)),
            "whole_frame_lanes_ms": all_ranks(local_time(whole_lanes)),
            "what": f"per step and rank, in rank order: this rank's stripes through {lanes} lane(s) without any exchange; the exchange "
                    "(collective + de-interleave on the root) of already rendered stripes without any render; the whole frame on this rank's GPU "
                    f"alone, one at a time and over its {lanes} lane(s)",
        }
        del whole

        def all_ok(ok):
            """every rank's verdict on a stage, before anybody enters the next collective (a failure on one rank only would leave the others
            waiting in it until the process-group timeout)"""
            if world == 1:
                return ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        variants = {}
        todo = [("rotate", vk, vk, bucket), ("rotate", vk, vk, 1), ("gather", 1, 1, 1)]
        for mode, cm, dp, bk in todo:
            name = f"{mode}_comms{cm}_depth{dp}_bucket{bk}"
            if (mode, cm, dp, bk) == (pipe.mode, pipe.cm, pipe.dp, pipe.bk):
                continue
            pv, err = None, None
            try:
                pv = Pipeline(mode, cm, dp, bk)
            except Exception as ex:  # a variant must never cost the bench line
                err = repr(ex)
            if not all_ok(err is None):  # (nobody starts a collective of a configuration that some rank could not build)
                variants[name] = {"error": err or "another rank could not build this configuration"}
                continue
            try:
                sec, nb_ = measure(pv.run)
                check(rxr.rxr_synchronize(ctx))
                check_assembled(pv)
                variants[name] = {"mpix_s": round(W * H * args.steps / sec / 1e6, 2), "ms_per_step": round(sec / args.steps * 1e3, 4), "batches": nb_,
                                  "sharding": pv.describe()}
            except (Exception, SystemExit) as ex:
                err = repr(ex)
                variants[name] = {"error": err}
            if not all_ok(err is None) and "error" not in variants[name]:
                variants[name] = {"error": "another rank failed in this configuration", **variants[name]}
            del pv

    # the same frame in the other light-loop arithmetic (rxr_set_light_math, include/rxr.h), N = 1 only: a few batches, reported
    # beside the default mode's `value`, never as it
    relaxed = os.environ.get("RXR_LIGHT_MATH", "exact" if host.rxh_get_light_math_exact() else "relaxed")[0] == "r"
    other_mode = None
    in_flight = None
    if world == 1 and not sharded and "RXR_LIGHT_MATH" not in os.environ:
        host.rxh_set_light_math_exact(1 if relaxed else 0)
        rc = host.rxh_rasterizer_upload(rast._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h)
        if rc == 0:
            run(args.warmup)
            other_s = [timed_batch() for _ in range(5)]
            check(rxr.rxr_synchronize(ctx))
            other_dt = float(np.median(other_s))
            other_mode = {"light_math": "exact" if relaxed else "relaxed", "ms_per_step": round(other_dt / args.steps * 1e3, 4),
                          "mpix_s": round(W * H * args.steps / other_dt / 1e6, 2), "batches": len(other_s)}
        host.rxh_set_light_math_exact(0 if relaxed else 1)
        upload()
    if world == 1 and not sharded and not args.no_in_flight:
        # two frames in flight on the one GPU (two member contexts, frame i on lane i mod 2, each on its own stream): throughput
        # of the same K whole frames when the tail of one frame's launches runs under the head of the next.  Reported beside
        # `value` (the serial loop, one frame at a time, as a caller of rasterize() sees it), never as it.
        try:
            serial = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")  # (the buffers of the timed loop may hold the other mode's frame)
            check(rxr.rxr_render_rows_to(ctx, 0, H, C.c_void_p(serial.data_ptr()), sptr))
            check(rxr.rxr_synchronize(ctx))
            torch.cuda.synchronize()
            use_lanes(2)
            g2 = upload()
            mem = [C.c_void_p(rxr.rxr_member(g2, k)) for k in range(2)]
            st2 = [torch.cuda.Stream() for _ in range(2)]
            fb2 = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]

            def run2(n):
                for i in range(n):
                    rc_ = rxr.rxr_render_rows_to(mem[i & 1], 0, H, C.c_void_p(fb2[i & 3].data_ptr()), C.c_void_p(st2[i & 1].cuda_stream))
                    if rc_ != 0:
                        raise RuntimeError(f"lane render failed: {rc_} {rxr.rxr_last_error(mem[i & 1])}")

            sec, nb_ = measure(run2)
            if rxr.rxr_synchronize(g2) != 0:
                raise RuntimeError(str(rxr.rxr_last_error(g2)))
            torch.cuda.synchronize()
            same = bool(torch.equal(fb2[0], serial) and torch.equal(fb2[3], serial))
            in_flight = {"lanes": 2, "ms_per_step": round(sec / args.steps * 1e3, 4), "mpix_s": round(W * H * args.steps / sec / 1e6, 2), "batches": nb_,
                         "identical_to_serial_frame": same,
                         "what": "the same K whole frames with two in flight on the GPU (two member contexts of one device, alternating streams)"}
            del fb2
        except Exception as ex:
            in_flight = {"error": repr(ex)}
        use_lanes(1)
        ctx = upload()

    fence()
    e2e = None
    if rank == 0 and not args.no_e2e:
        try:
            e2e = end_to_end(prod, host, rxr, cfg, W, H, world)
        except Exception as ex:  # the end-to-end leg must never cost the bench line
            e2e = {"error": repr(ex)}
    fence()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = W * H * args.steps / dt / 1e6
        # algorithmic bytes of ONE raster launch: this rank's share of the framebuffer + the scene read once
        n_verts, n_tris, tex_bytes = scene_counts(cfg, prod)
        rows_this_rank = min(stripe_rows, H) if sharded else H
        alg = algorithmic_bytes(W, rows_this_rank, n_verts, n_tris, tex_bytes, args.lights)
        achieved = alg / (raster_avg_us * 1e-6) / 1e9
        default_workload = (W, H, args.lights, world) == (3840, 2160, 16, 1) and not sharded
        PROFILE = PROFILES.get("relaxed" if relaxed else "exact")
        default_workload = default_workload and PROFILE is not None
        # the compute-side rooflines of the same kernel ("state both fractions"): instruction and flop counts from the
        # committed PMC passes, the duration measured live
        valu_issue_frac = fp32_frac = fp32_tflops = None
        if default_workload and raster_avg_us == raster_avg_us:
            t_s = raster_avg_us * 1e-6
            valu_issue_frac = PROFILE["valu_wave_instructions"] * VALU_ISSUE_CYCLES / (N_SIMDS * CLOCK_GHZ * 1e9 * t_s)
            f = PROFILE["fp32_wave_instructions"]
            if PROFILE["valu_lane_utilisation"] and (f["ADD"] + f["MUL"] + f["FMA"]):
                fp32_tflops = (f["ADD"] + f["MUL"] + 2 * f["FMA"]) * 64 * PROFILE["valu_lane_utilisation"] / t_s / 1e12
                fp32_frac = fp32_tflops / FP32_VALU_PEAK_TFLOPS
        out = {
            "metric": "Mpixels/s (+ ms/frame) on rasterize_map @3840x2160, 1/2/4/8 MI355X vs CPU",
            "value": round(value, 2),
            "unit": "Mpixels/s",
            # inputs resident in HBM when the timed region starts, the frame left in HBM; the drop-in call into host pixels (host
            # projection + upload + kernels + PCIe download) is `e2e_ms` below, never `value`
            "value_semantics": "device-resident",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            **({"rehearsal": True} if REHEARSAL else {}),
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "timing": {
                "batches": n_batches + 1,
                "batch_steps": args.steps,
                "timed_s": round(float(np.sum(batch_s)) + first, 3),
                "reported": "median batch of `steps` steps, each batch between barrier + synchronize on both sides",
                "batch_ms_min_median_max": [round(float(np.min(batch_s)) * 1e3, 4), round(dt * 1e3, 4), round(float(np.max(batch_s)) * 1e3, 4)],
            },
            "config": {
                "workload": f"map scene (minigame room), {W}x{H}, {args.lights} point lights, Nearest sampling, "
                            "fence cut-outs, 2D logo rectangle; UV jitter absent in the reference snapshot",
                "resolution": [W, H],
                "triangles_3d": n_tris,
                # arithmetic of the 3D light loop (rxr_set_light_math): "relaxed" = the library's default, point lights within
                # BASELINE.json's 1-per-channel tolerance for lit 3D fragments (tests/test_gpu_light_math.py: 112 of 8 294 400
                # pixels of this frame differ from the CPU oracle, each by 1); "exact" = correctly rounded throughout
                "light_math": "relaxed" if relaxed else "exact",
                "sharding": "single GPU, one frame at a time" if not sharded else pipe.describe(),
                **({"frames_in_flight_per_gpu": lanes, "frames_per_exchange": bucket} if sharded else {}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_raster_rl" if relaxed else "k_raster",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                # the bound that binds: VALU wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time),
                # and fp32 flops against the 157.3 TFLOP/s vector peak
                "valu_issue_frac": round(valu_issue_frac, 4) if valu_issue_frac is not None else None,
                "fp32_frac": round(fp32_frac, 4) if fp32_frac is not None else None,
                "fp32_tflops": round(fp32_tflops, 2) if fp32_tflops is not None else None,
                # HBM bytes of one launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE with the gfx950 x2
                # correction + WRITE_SIZE): from the committed profile, NOT measured in this run; only known for the default workload
                "traffic": PROFILE["fetch_bytes_x2"] + PROFILE["write_bytes"] if default_workload else None,
                "traffic_from": PROFILE["source"] if default_workload else None,
                # measured live in this run, on every PROFILE_STRIDE-th frame of the timed region: HIP start / stop events bound to the
                # dispatches themselves (hipExtLaunchKernel on the launch stream; no event record between the launches)
                "algorithmic_bytes_per_launch": int(alg),
                "kernel_avg_us": round(raster_avg_us, 2),
                "kernel_samples": int(n_prof.value),
                "kernel_timing": f"per-dispatch start/stop HIP events, 1 frame in {PROFILE_STRIDE}",
                "setup_kernels_avg_us": round(setup_avg_us, 2),
                # kernels of one step against the step (one frame at a time: they cannot exceed it; lanes overlap other lanes' launches)
                "kernels_fit_step": bool(raster_avg_us + setup_avg_us <= ms_per_step * 1e3 * 1.005) if not (sharded and lanes > 1) else None,
                "note": "the kernel is fp32-VALU bound, not HBM bound: algorithmic HBM traffic is ~4.5 B/pixel (DESIGN.md section 6)"
                        + ("; with several frames in flight the event-timed launches overlap other lanes' launches" if sharded and lanes > 1 else ""),
                "from_profiles": {
                    "source": PROFILE["source"],
                    "write_bytes": PROFILE["write_bytes"],
                    "fetch_bytes_x2": PROFILE["fetch_bytes_x2"],
                    "valu_wave_instructions_per_launch": PROFILE["valu_wave_instructions"],
                    "fp32_wave_instructions_per_launch": PROFILE["fp32_wave_instructions"],
                    "valu_lane_utilisation": PROFILE["valu_lane_utilisation"],
                    "traced_kernel_avg_us": PROFILE["traced_kernel_avg_us"],
                    "traced_setup_avg_us": PROFILE["traced_setup_avg_us"],
                    # the kernel source the profile was taken on against the one this run launched
                    "kernel_sources_sha": PROFILE["kernel_sources_sha"],
                    "stale": PROFILE["kernel_sources_sha"] != kernel_sources_sha(),
                } if default_workload else None,
            },
        }
        if other_mode is not None:
            out["other_light_math"] = other_mode
        if in_flight is not None:
            out["two_frames_in_flight"] = in_flight
        if sharded:
            # What bounds `value` at this N, stated in the line itself: the exchange BASELINE.json names is a gather to rank 0, whose root
            # receives every other rank's share of EVERY frame, each over the one xGMI link from that rank.  A link is busy for
            # share_bytes / rate per frame whatever the render side does; against a whole frame on one GPU that is a ceiling on the
            # speed-up.  (The rotating-root variants below put a stripe set on each link once in N frames.)
            n1 = float(np.median(per_rank["whole_frame_serial_ms"])) if per_rank else None
            out["exchange_bound"] = {
                "exchange": "gather to rank 0 over xGMI (BASELINE.json configs[3])",
                "root_link_bytes_per_frame": int(share_bytes) if world > 1 else 0,
                "link_GBps": {"rccl_low": XGMI_LINK_GBS[0], "rccl_high": XGMI_LINK_GBS[1], "peak": XGMI_LINK_GBS[2]},
                "min_ms_per_frame": {k: round(share_bytes / (r * 1e9) * 1e3, 4) for k, r in zip(("rccl_low", "rccl_high", "peak"), XGMI_LINK_GBS)} if world > 1 else None,
                "whole_frame_one_gpu_ms": round(n1, 4) if n1 else None,
                "speedup_ceiling": {k: round(n1 / (share_bytes / (r * 1e9) * 1e3), 2) for k, r in zip(("rccl_low", "rccl_high", "peak"), XGMI_LINK_GBS)} if (n1 and world > 1) else None,
                "measured_on_hardware": not REHEARSAL,
                "note": "link rates are assumptions until a multi-GPU run exists: no round has had one (SCALE_r01..r03 skipped)",
            }
            if variants:
                ok = {k: v for k, v in variants.items() if "ms_per_step" in v}
                ok["value: " + pipe.mode + f"_comms{pipe.cm}_depth{pipe.dp}_bucket{pipe.bk}"] = {"ms_per_step": round(ms_per_step, 4)}
                best = min(ok, key=lambda k: ok[k]["ms_per_step"])
                out["best_variant"] = {"name": best, "ms_per_step": ok[best]["ms_per_step"], "x_vs_value": round(ms_per_step / ok[best]["ms_per_step"], 3),
                                       "x_vs_whole_frame_one_gpu": round(n1 / ok[best]["ms_per_step"], 3) if n1 else None}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if variants is not None:
            out["exchange_variants"] = variants
        if e2e is not None:
            out.update(e2e)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(W, H, args.lights)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def end_to_end(prod, host, rxr, cfg, W, H, world):
    """What a caller of the drop-in call sees: Rasterizer::setup(..).rasterize(scene, pixels, ..) with `pixels` in host memory
    = Scene::project on the host + flatten + upload + kernels + download.  At N > 1 the same call on a multi-device context over
    all N GPUs of the node (rxr_create_multi: every GPU renders its stripes and copies them into `pixels` itself)."""
    from rusterix_amd import scenes

    if world > 1:
        ids = (C.c_int * world)(*([0] * world if REHEARSAL else range(world)))
        host.rxh_set_devices(ids, world)
    out = np.zeros(W * H * 4, np.uint8)

    def median_ms(n_min=20, t_min=0.5):
        scenes.render(cfg, out)  # warm-up (textures, buffer growth)
        scenes.render(cfg, out)
        ts, t_start = [], time.perf_counter()
        while len(ts) < n_min or time.perf_counter() - t_start < t_min:
            t0 = time.perf_counter()
            scenes.render(cfg, out)
            ts.append(time.perf_counter() - t0)
            if len(ts) >= 2000:
                break
        return float(np.median(ts)) * 1e3, len(ts)

    ms, n = median_ms()
    frame = out.reshape(H, W, 4)
    assert int(frame[..., 3].min()) == 255 and int(frame[..., :3].max()) > 0, "end-to-end frame is not a rendered frame"
    res = {"e2e_ms": round(ms, 4), "e2e_mpix_s": round(W * H / ms / 1e3, 1), "e2e_frames": n,
           "e2e_what": "median wall time of Rasterizer::rasterize into pageable host pixels: host Scene::project + flatten + upload + kernels + download"
                       + ("" if world == 1 else f", multi-device context over {world} GPUs (each GPU downloads its own stripes)")}
    # ... and into page-locked pixels from the library's allocator (rxr_alloc_pinned; nothing of the malloc heap is locked: rxr.h)
    from rusterix_amd.binding import pinned_pixels

    locked, free_locked = pinned_pixels(rxr, out.nbytes)
    if locked is not None:
        pageable = out
        try:
            out = locked
            ms_p, _ = median_ms()
            res["e2e_pinned_ms"] = round(ms_p, 4)
        finally:
            out = pageable
            del locked
            free_locked()
    return res


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


def cpu_baseline(W, H, n_lights):
    """The CPU oracle (C++ restatement of the reference algorithm, threaded over tiles like rayon; built -O3
    -ffp-contract=off, oracle/Makefile) timed on this host's cores on the SAME workload, on a bounded sample: one
    frame per thread count of a sweep, then >= 10 s of frames at the best count; plus one 1-thread frame at a quarter of
    the pixels (a 1-thread 4K frame alone would take most of the budget)."""
    from rusterix_amd import scenes
    from tests.oracle_api import load_oracle

    orc = load_oracle()
    cfg = scenes.map_scene(orc, width=W, height=H, n_lights=n_lights)
    out = np.zeros(W * H * 4, np.uint8)
    cores = os.cpu_count() or 1

    def frame(threads, c=cfg, o=out):
        r = orc.set_threads(c.setup(), threads)
        t0 = time.perf_counter()
        r.rasterize(c.scene, o, c.width, c.height, c.tile_size, c.assets)
        return time.perf_counter() - t0

    frame(cores)  # warm-up frame (not timed)
    sweep = {}
    for t in sorted({max(1, cores // 8), max(1, cores // 4), max(1, cores // 2), cores}):
        sweep[t] = min(frame(t), frame(t))
    best = min(sweep, key=sweep.get)
    t0 = time.perf_counter()
    done = 0
    while True:
        frame(best)
        done += 1
        el = time.perf_counter() - t0
        if el >= 10.0 or (done >= 2 and el >= 20.0):
            break
    dt = time.perf_counter() - t0
    # one thread, a quarter of the pixels (same scene and lights at half the width and height)
    small = scenes.map_scene(orc, width=W // 2, height=H // 2, n_lights=n_lights)
    small_out = np.zeros((W // 2) * (H // 2) * 4, np.uint8)
    t1 = frame(1, small, small_out)
    value = W * H * done / dt / 1e6
    one = (W // 2) * (H // 2) / t1 / 1e6
    return {
        "value": round(value, 3),
        "unit": "Mpixels/s",
        "cores": best,
        "host_cores": cores,
        "kind": "port",
        "ms_per_frame": round(dt / done * 1e3, 1),
        "thread_sweep_mpix_s": {str(t): round(W * H / s / 1e6, 2) for t, s in sweep.items()},
        "one_thread_mpix_s": round(one, 3),
        "scaling_1_to_best": round(value / one, 1),
        "sample": f"{done} full frames of the same workload ({W}x{H}, {n_lights} lights, tile_size 40) on {best} threads (the best of the sweep "
                  f"{sorted(sweep)}, one warm-up frame before); 1-thread figure: one frame at {W // 2}x{H // 2}; C++ restatement of the reference "
                  "algorithm (-O3 -ffp-contract=off), std::thread pool over tiles, includes Scene::project.  Scaling: "
                  f"{one:.2f} Mpixel/s on 1 thread -> {value:.1f} on {best} threads = {value / one:.1f}x (of {cores} hardware threads; more threads are "
                  "slower: the five per-tile heap allocations of rasterizer.rs:277-290, which the port keeps, serialise in the allocator).  "
                  "Context for the GPU number, never credit",
    }


if __name__ == "__main__":
    main()
