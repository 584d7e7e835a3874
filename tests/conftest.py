import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# compiled program sets are also kept on disk between processes (rxr_jit.hip): the tests want every run to start from nothing
# (tests/test_shader_jit_cpu.py switches the cache on, in a directory of its own)
os.environ.setdefault("RXR_JIT_CACHE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure) bound to the reference-shaped Python classes."""
    from tests.oracle_api import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def product():
    """The product: C++ host mirror + HIP back end behind the C ABI.  Fails loudly if not built."""
    import rusterix_amd

    return rusterix_amd.load()


def pytest_runtest_logstart(nodeid, location):
    """RXR_TEST_TIMESTAMPS=<file>: one line per test with its start time (to match leftovers of a run to the test that made them)."""
    path = os.environ.get("RXR_TEST_TIMESTAMPS")
    if path:
        import time

        with open(path, "a") as f:
            f.write("%s %d %s\n" % (time.strftime("%H:%M:%S"), os.getpid(), nodeid))
