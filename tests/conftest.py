import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size runs (tens of seconds each); part of -m gpu, deselect with -m \"gpu and not slow\"")


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
    import json
    import platform
    from particlerobotsimulations_amd import _capi, host
    L = host.lib()
    L.pbHostLibmCheck.argtypes = [C.c_int, C.c_uint] + [C.POINTER(C.c_ulonglong)] * 3
    L.pbHostLibmCheck.restype = C.c_int
    L.pbHostLibcVersion.restype = C.c_char_p
    # The exhaustive sweep (every non-negative float, ~15 s on 8 cores) runs once per (glibc, CPU model, library
    # build): its verdict is cached next to the library; tests/test_libm_pin.py always runs it in the CPU suite.
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        cpu = platform.processor()
    key = {"glibc": L.pbHostLibcVersion().decode(), "cpu": cpu, "lib_mtime": os.path.getmtime(_capi.HOST_SO)}
    cache = os.path.join(os.path.dirname(_capi.HOST_SO), "libm_check.json")
    verdict = None
    try:
        rec = json.load(open(cache))
        if rec.get("key") == key:
            verdict = rec
    except Exception:
        pass
    if verdict is None:
        n, bad, inv = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        rc = L.pbHostLibmCheck(0, 1, C.byref(n), C.byref(bad), C.byref(inv))
        verdict = {"key": key, "ok": rc == 0, "checked": n.value, "square_mismatches": bad.value,
                   "root_inversions": inv.value}
        try:
            json.dump(verdict, open(cache, "w"))
        except OSError:
            pass
    if not verdict["ok"]:
        sys.stderr.write(f"conftest: this host's powf fails the phase-update assumptions ({verdict['square_mismatches']} "
                         f"square mismatches, {verdict['root_inversions']} root inversions of {verdict['checked']}); "
                         "using the host loop (pbSetMinDistanceMode 1, PB_MIN_DISTANCE_MODE=1 for child processes)\n")
        _capi.check(_capi.lib().pbSetMinDistanceMode(1), "pbSetMinDistanceMode")
        os.environ["PB_MIN_DISTANCE_MODE"] = "1"  # particlebot_run / particlebot_ensemble / bench.py started by tests
