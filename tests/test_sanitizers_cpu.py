"""AddressSanitizer + UndefinedBehaviorSanitizer over what runs on the CPU (tools/sanitize_cpu.sh): the oracle rendering seeded
scenes of the test suite and the product's host mirror (mesh builders, clip_and_project, Scene::project through the worker pool,
the C API glue) projecting them; then the HOST code of librxr_hip.so at the shader boundary (rxr_check_shaders) fed with random
well-formed programs and ~1800 malformed word streams (tools/sanitize_shaders.py).  GPU sanitizers are not available on the pool;
the device code is covered by parity."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_mirror_are_clean_under_asan_and_ubsan():
    probe = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(probe):
        pytest.skip("no AddressSanitizer runtime for this g++")
    pr = subprocess.run([os.path.join(ROOT, "tools", "sanitize_cpu.sh"), "1"], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert pr.returncode == 0 and "sanitizers: clean" in pr.stdout, (pr.stdout + pr.stderr)[-6000:]
    assert "shader boundary under sanitizers: clean" in pr.stdout or "stage 2 skipped" in pr.stdout, (pr.stdout + pr.stderr)[-6000:]
