import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure)."""
    o = entry.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    """ctypes harness over libmibayer.so; builds it if the .so is missing."""
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    p.lib()
    return p


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    if pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible: the HIP path has no CPU fallback")
    return pkg


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "bayer2rgb_small.npz"))
