import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
# The parity suite compares with the reference bit for bit: every context of the test session - and of the programs it starts -
# begins in the exact arithmetic tier (a new context otherwise starts in the tolerant one, include/blacklight_amd.h). Tests of the
# tolerant tier ask for it by name.
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")


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
