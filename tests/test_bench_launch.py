"""bench.py's launch contract, without a GPU: it must refuse to measure fewer GPUs than it was asked for (a silent
`n_gpus: 1` line for `--gpus 8` is an invalid measurement) and must refuse a launcher whose world size differs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_visible_fails_before_any_measurement():
    r = run(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 64 but only" in r.stderr
    assert r.stdout.strip() == ""  # no JSON line


def test_world_size_must_equal_gpus():
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 2 but 4 rank(s) were launched" in r.stderr
    r = run(["--gpus", "1", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 1 but 2 rank(s) were launched" in r.stderr
