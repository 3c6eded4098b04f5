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


@pytest.fixture(scope="session", autouse=True)
def _libm_gate(request, _built_libraries):
    """Once per GPU session, before any GPU test: the libm properties the phase update's 4-byte read-back rests
    on (tests/test_libm_pin.py holds the same check for the CPU suite).  If this host's powf does not have them,
    every batch created in this session uses the reference's own host loop instead (pbSetMinDistanceMode(1)):
    results stay the reference's, only the read-back grows."""
    expr = request.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import ctypes as C
    from particlerobotsimulations_amd import _capi, host
    L = host.lib()
    L.pbHostLibmCheck.argtypes = [C.c_int, C.c_uint] + [C.POINTER(C.c_ulonglong)] * 3
    L.pbHostLibmCheck.restype = C.c_int
    n, bad, inv = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    if L.pbHostLibmCheck(0, 1, C.byref(n), C.byref(bad), C.byref(inv)) != 0:
        sys.stderr.write(f"conftest: this host's powf fails the phase-update assumptions ({bad.value} square mismatches, "
                         f"{inv.value} root inversions of {n.value}); using the host loop (pbSetMinDistanceMode 1)\n")
        _capi.check(_capi.lib().pbSetMinDistanceMode(1), "pbSetMinDistanceMode")
