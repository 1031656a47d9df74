import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Builds libdffw.so in-tree if it is missing (hipcc cross-compiles gfx950 without a GPU)."""
    so = os.path.join(ROOT, "dffinthewild_amd", "libdffw.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return so


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
