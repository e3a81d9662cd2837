import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "blender-ngp_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is test infrastructure: build it on demand (plain gcc, a second or two).
    so = os.path.join(ROOT, "oracle", "_build", "libngp_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never returns, a spinning host loop) must fail, not hold the box: pytest-timeout (in the image) ends it after 5 minutes —
    the slowest test of the suite takes under 20 s."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def oracle():
    import helpers
    return helpers.load_oracle()


@pytest.fixture(scope="session")
def ngp():
    import capi
    return capi.load_ngp_hip()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product path has no CPU fallback")
    return torch.device("cuda:0")
