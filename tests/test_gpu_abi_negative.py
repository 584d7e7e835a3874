"""tests/abi_negative.c: the C ABI from a plain C program on a GPU box -- one valid frame, then ~35 malformed variants of it and
calls in the wrong order.  Every one must come back as an error status with a message; the context must keep working."""
import os
import subprocess

import pytest

import rusterix_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_malformed_frames_are_answered_with_error_codes(tmp_path):
    lib = rusterix_amd.lib_paths()["rxr"]
    exe = str(tmp_path / "abi_negative")
    cc = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_negative.c"),
                         "-o", exe, "-L" + os.path.dirname(lib), "-lrxr_hip", "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok (0 failures)"), run.stdout[-6000:] + run.stderr[-2000:]
    assert run.stdout.count("rc=-") >= 30
