import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The in-tree library is git-ignored: build it once per session if a fresh checkout lacks it (hipcc
    cross-compiles for gfx950 without a GPU).  Host-only helpers (CRC-32C, TIFF LZW) live in it too."""
    from moonsuperresolution_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP library; built on demand so a fresh checkout works (hipcc cross-compiles on CPU)."""
    from moonsuperresolution_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()
