"""tests/abi_stream.c: the streaming hand-over of large frames (rxr_stream_begin / rxr_stream_begin_pinned / rxr_stream_batch3d) from a
plain C program on a GPU box -- in order, in reverse, from four threads, out of page-locked memory, and with every way of breaking
the protocol: each frame must be byte-identical to the plainly uploaded one.  Plus the host mirror's own use of it (forced onto a
mid-sized scene) against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import rusterix_amd
from rusterix_amd import scenes
from tests.test_gpu_parity import assert_exact

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streamed_frames_equal_the_plain_upload(tmp_path):
    lib = rusterix_amd.lib_paths()["rxr"]
    exe = str(tmp_path / "abi_stream")
    cc = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_stream.c"),
                         "-o", exe, "-L" + os.path.dirname(lib), "-lrxr_hip", "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout[-6000:] + run.stderr[-2000:]
    assert "DIFFERENT" not in run.stdout and run.stdout.count("identical") >= 9
    assert "pulled by the device" in run.stdout  # the page-locked cases ran


def test_contexts_of_different_threads_do_not_meet(tmp_path):
    """tests/abi_threads.c: four host threads, each with a plain and a two-member context of its own on the one GPU, render their own
    frames at the same time (plain uploads, streamed hand-overs): every frame is the frame the main thread rendered alone"""
    lib = rusterix_amd.lib_paths()["rxr"]
    exe = str(tmp_path / "abi_threads")
    cc = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_threads.c"),
                         "-o", exe, "-L" + os.path.dirname(lib), "-lrxr_hip", "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout[-6000:] + run.stderr[-2000:]
    assert run.stdout.count("40 frames, 0 failures") == 4


@pytest.mark.parametrize("mode", ["pinned", "copy", "off"])
def test_host_mirror_streams_a_projected_scene(oracle, product, monkeypatch, mode):
    """Rasterizer::rasterize of the host mirror hands every 3D batch over while the others are still being projected (forced here onto
    a 72 x 72 box grid; by default only scenes of a million elements take that path): page-locked arrays pulled by the device, ordinary
    arrays copied and shipped in groups, and the plain sequence must all give the oracle's frame"""
    if mode == "off":
        monkeypatch.setenv("RXR_STREAM_UPLOAD", "0")
    else:
        monkeypatch.setenv("RXR_STREAM_UPLOAD", "force")
        if mode == "copy":
            monkeypatch.setenv("RXR_PINNED_ARRAYS", "0")
    cfg = scenes.box_grid_scene(product, n=72, width=640, height=360)
    got = scenes.render(cfg).copy()
    again = scenes.render(cfg).copy()  # a second frame through the same (now allocated) arrays and stream state
    ref = scenes.render(scenes.box_grid_scene(oracle, n=72, width=640, height=360))
    assert_exact(got, ref, f"host-projected box grid, hand-over {mode}")
    assert_exact(again, ref, f"host-projected box grid, hand-over {mode}, second frame")
    if mode != "off":
        info = rusterix_amd.rxr_abi().rxr_debug_stream_info(product.lib.rxh_context())
        assert info == (2 if mode == "pinned" else 1), f"the frame was not streamed ({info})"
