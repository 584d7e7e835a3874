"""bench.py's N > 1 path, rehearsed on ONE GPU (RXR_BENCH_REHEARSAL=1: both ranks on GPU 0, exchange over gloo with host staging --
RCCL refuses two ranks on a device).  No multi-GPU hardware is available to the builder, so this is what keeps the rank
bookkeeping, the stripe partition, the pipelining indices, the byte-identity check against the single-launch frame (asserted
inside bench.py), the multi-device end-to-end leg and the JSON line from being run for the first time by the driver."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,exchange,extra", [(2, "gather", []), (3, "allgather", ["--no-variants"]), (3, "rotate", ["--no-variants", "--bucket", "1", "--lanes", "1"]),
                                                  # several communicators, two exchanges in flight, three lanes, buckets of 3 (6 steps = 2 buckets)
                                                  (3, "rotate", ["--comms", "3", "--depth", "2", "--lanes", "3", "--bucket", "3", "--no-variants"]),
                                                  # a ragged last bucket (6 steps in buckets of 4)
                                                  (2, "gather", ["--bucket", "4", "--no-variants"])])
def test_bench_runs_its_multi_rank_path(world, exchange, extra):
    env = dict(os.environ, RXR_BENCH_REHEARSAL="1", RXR_BENCH_MIN_TIMED_S="0.05")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--no-cpu",
                         "--width", "1280", "--height", "720", "--exchange", exchange] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-4000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["rehearsal"] is True and d["steps"] == 6
    assert f"{world} GPUs" in d["config"]["sharding"] and exchange in d["config"]["sharding"]
    assert d["value"] > 0 and "e2e_ms" in d and f"{world} GPUs" in d["e2e_what"]
    if "--no-variants" not in extra:
        # the same run times the variants that lift the fixed root's link bound, and every rank's render-only / exchange-only legs
        pr_, ev = d["per_rank"], d["exchange_variants"]
        assert len(pr_["render_only_ms"]) == world and len(pr_["exchange_only_ms"]) == world and min(pr_["render_only_ms"]) > 0
        assert any(k.startswith("rotate_") for k in ev) and all("error" not in v for v in ev.values()), ev
        assert all(v["mpix_s"] > 0 for v in ev.values())
        assert d["config"]["frames_in_flight_per_gpu"] == 2 and d["config"]["frames_per_exchange"] == 4
        # the line explains itself (round-3 verdict): the root's link bound of the gather BASELINE.json names, the whole frame on one GPU in
        # the same run (one at a time and over the lanes), and the fastest exchange of the run by name
        eb, bv = d["exchange_bound"], d["best_variant"]
        assert eb["root_link_bytes_per_frame"] > 0 and eb["measured_on_hardware"] is False and eb["whole_frame_one_gpu_ms"] > 0
        assert set(eb["min_ms_per_frame"]) == set(eb["speedup_ceiling"]) == {"rccl_low", "rccl_high", "peak"}
        assert eb["speedup_ceiling"]["rccl_low"] < eb["speedup_ceiling"]["rccl_high"] < eb["speedup_ceiling"]["peak"]
        assert len(pr_["whole_frame_serial_ms"]) == world and len(pr_["whole_frame_lanes_ms"]) == world and min(pr_["whole_frame_lanes_ms"]) > 0
        assert bv["ms_per_step"] > 0 and bv["x_vs_value"] >= 1.0 and (bv["name"] in ev or bv["name"].startswith("value: "))
    assert d["value_semantics"] == "device-resident"
