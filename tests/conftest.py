import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _native_pieces_built():
    """Build the native pieces (hipcc cross-compiles gfx950 without a GPU) if a fresh checkout has none yet.
    Building is not a fallback: the product still fails loudly without its library."""
    lib = os.path.join(ROOT, "oclradixsort_amd", "lib", "libadlhip.so")
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__
        __graft_entry__.build()
