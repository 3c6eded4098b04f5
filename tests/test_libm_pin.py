"""The libm properties the phase update rests on, pinned on the machine the suite runs on.

The reference takes  min_i powf(powf(lx - x_i, 2) + powf(ly - y_i, 2), 0.5f)  on the host
(/root/reference particlebot.cpp:215-228).  The engine reduces  min_i (dx*dx + dy*dy)  on the device and takes
powf(., 0.5f) of that one value on the host (csrc/pb_engine.hip phaseUpdate, mode 0).  Equal for every input iff
  (a) powf(x, 2.0f) == x*x bit for bit for every float, and
  (b) powf(., 0.5f) is non-decreasing over the non-negative floats
on the host's libm.  pbHostLibmCheck (libparticlebot_host.so) checks both over EVERY non-negative float (2^31 - 2^23
values; (a) for both signs).  If this test ever fails on a host, call pbSetMinDistanceMode(1) (the reference's own
host loop over all positions: no assumption) -- tests/conftest.py does that for the GPU suite by itself.
Also pinned: powf(x, 0.5f) is NOT sqrtf(x) for about 1.4 x 10^6 floats, i.e. the root must stay a host powf."""
import ctypes as C

import numpy as np


def _lib():
    from particlerobotsimulations_amd import host
    L = host.lib()
    L.pbHostLibmCheck.argtypes = [C.c_int, C.c_uint] + [C.POINTER(C.c_ulonglong)] * 3
    L.pbHostLibmCheck.restype = C.c_int
    L.pbHostLibcVersion.restype = C.c_char_p
    return L


def test_powf_square_is_a_multiply_and_powf_root_is_monotone_on_this_host():
    L = _lib()
    checked, bad, inv = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    rc = L.pbHostLibmCheck(0, 1, C.byref(checked), C.byref(bad), C.byref(inv))
    print("glibc", L.pbHostLibcVersion().decode(), "checked", checked.value, "square mismatches", bad.value,
          "root inversions", inv.value)
    assert checked.value == 0x7F800000 + 1           # every non-negative float up to and including +inf
    assert bad.value == 0 and inv.value == 0 and rc == 0


def test_powf_root_is_not_sqrtf():
    """Why the host root cannot be replaced by a device sqrtf: glibc's powf(x, 0.5f) differs from the correctly
    rounded square root on a sample of floats (1 361 091 of all positive floats on glibc 2.35)."""
    libm = C.CDLL("libm.so.6")
    powf = libm.powf
    powf.restype = C.c_float
    powf.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(1)
    xs = rng.integers(0x00800000, 0x7F000000, size=200_000, dtype=np.uint32).view(np.float32)
    diff = sum(1 for x in xs[:60000] if np.float32(powf(float(x), 0.5)) != np.sqrt(x))
    assert diff > 0
