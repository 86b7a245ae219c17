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
    """The in-tree library is git-ignored: (re)build it whenever it is missing or was built from other sources than
    the ones in the tree (content hash in csrc/.build_stamp — a stale .so after a kernel edit would silently test old
    code).  hipcc cross-compiles for gfx950 without a GPU.  Host-only helpers (CRC-32C, TIFF LZW) live in it too."""
    from moonsuperresolution_amd import _lib
    if "MSR_LIB" not in os.environ and _lib.is_stale():
        _lib.build(force=True)


def pytest_report_header(config):
    from moonsuperresolution_amd import _lib
    return f"libmoonsr_hip: {os.environ.get('MSR_LIB', _lib.LIB_PATH)}"


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP library; built on demand so a fresh checkout works (hipcc cross-compiles on CPU)."""
    from moonsuperresolution_amd import _lib
    if "MSR_LIB" not in os.environ and _lib.is_stale():
        _lib.build(force=True)
    return _lib.load()
