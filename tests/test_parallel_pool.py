"""The host worker pool (rusterix_amd/csrc/rxr_parallel.h: Scene::project over batches, the per-batch copies of rxr_upload_frame)
on its own: tests/parallel_pool_check.cpp built with ThreadSanitizer (sanitizers run on the CPU build only) and plain (the fork case,
which ThreadSanitizer does not support)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "parallel_pool_check.cpp")


def build(tmp_path, name, *flags):
    exe = str(tmp_path / name)
    pr = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", *flags, "-o", exe, SRC], capture_output=True, text=True)
    return exe, pr


def test_pool_runs_every_item_once_and_survives_a_fork(tmp_path):
    exe, pr = build(tmp_path, "pool_check")
    assert pr.returncode == 0, pr.stderr
    for threads in ("6", "2", "1"):
        run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, RXR_HOST_THREADS=threads))
        assert run.returncode == 0 and run.stdout.strip().startswith("ok"), run.stdout + run.stderr


def test_pool_is_clean_under_thread_sanitizer(tmp_path):
    exe, pr = build(tmp_path, "pool_check_tsan", "-fsanitize=thread")
    if pr.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this g++: " + pr.stderr[-200:])
    run = subprocess.run([exe, "nofork"], capture_output=True, text=True, timeout=600, env=dict(os.environ, RXR_HOST_THREADS="6"))
    assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr
