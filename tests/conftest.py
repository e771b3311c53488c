import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Path of the in-tree HIP library; builds it with hipcc if it is missing."""
    import blacklight_amd
    if not os.path.exists(blacklight_amd.LIB_PATH):
        from blacklight_amd import build
        build.build()
    return blacklight_amd.LIB_PATH
