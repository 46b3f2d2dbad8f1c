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
def lab_pkg():
    """The same harness over libmibayer_lab.so (`make lab`, -DMIBAYER_LAB): the experiment kernel arms and the tuning
    environment variables that the product build does not carry."""
    p = entry.load_package(lab=True)
    if not os.path.exists(p.LIB_PATH):
        p.build()
    assert p.lib().mibayer_is_lab_build() == 1
    return p


@pytest.fixture(scope="session")
def gpu_lab_pkg(lab_pkg):
    if lab_pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible: the HIP path has no CPU fallback")
    return lab_pkg


@pytest.fixture(autouse=True)
def _fresh_plan_cache():
    """A plan one test measured (mibayer_autotune -> the process-wide plan cache) must not become another test's
    default."""
    for name in (entry.PKG_NAME, entry.PKG_NAME + "_lab"):
        mod = sys.modules.get(name)
        if mod is not None and getattr(mod, "_lib", None) is not None:
            mod._lib.mibayer_plan_cache_clear()
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "bayer2rgb_small.npz"))


@pytest.fixture(scope="session")
def golden_r2b():
    """rgb2bayer fixtures: outputs of the reference's own gst_rgb2bayer_transform (tests/golden/make_golden.py)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "rgb2bayer_small.npz"))
