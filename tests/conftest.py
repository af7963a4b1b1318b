import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rgb-d-slam_amd", "python"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built libraries are git-ignored; they normally travel with the working tree, but a fresh checkout has none.
    # Building is not a fallback for the product (which still fails loudly without its .so): it is what build() does.
    lib = os.path.join(ROOT, "rgb-d-slam_amd", "lib")
    if not all(os.path.exists(os.path.join(lib, f)) for f in ("libcape_hip.so", "libcape_primitives.so", "test_shim.exe", "test_polygon.exe")):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "rgb-d-slam_amd", "csrc"), "all", "host"])


@pytest.fixture(scope="session")
def oracle_mod():
    import cape_oracle_py

    cape_oracle_py.build()
    return cape_oracle_py
