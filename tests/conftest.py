import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The in-tree libraries travel with the snapshot; if they are missing (fresh clone), build them
    once (hipcc cross-compiles gfx950 without a GPU)."""
    from particlerobotsimulations_amd import _capi
    if not (os.path.exists(_capi.HIP_SO) and os.path.exists(_capi.HOST_SO)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker only)."""
    from oracle import orclib
    orclib.build()
    return orclib


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
