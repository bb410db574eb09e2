import os
import sys

import pytest

# PyTorch bundles a HIP runtime of its own.  When a test file that uses torch is collected, torch loads first and
# libjuliet_hip.so shares that runtime; if libjuliet_hip.so initialises /opt/rocm's runtime first and torch comes later
# (a run of a subset of the files), the second runtime finds no device.  Fix the order for every selection of tests.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the CPU tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.load()
