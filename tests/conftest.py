import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
