import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# oracle/ is test infrastructure; the product package lives in adder-codec-rs_amd/
for p in (ROOT, os.path.join(ROOT, "adder-codec-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """The suites need the in-tree shared libraries.  __graft_entry__.build() makes them; if a checkout
    has not been built yet, build what is missing once (hipcc cross-compiles gfx950 without a GPU)."""
    import shutil
    import subprocess
    pkg = os.path.join(ROOT, "adder-codec-rs_amd")
    if not os.path.exists(os.path.join(pkg, "libadder_hip.so")) and shutil.which("hipcc"):
        subprocess.check_call(["make", "-C", pkg, "-s"])
    if os.path.exists(os.path.join(pkg, "libadder_hip.so")) and \
            not os.path.exists(os.path.join(pkg, "host", "libadder_host.so")):
        subprocess.check_call(["make", "-C", os.path.join(pkg, "host"), "-s"])
    yield
