"""The static-friction hold compares squared lengths against a host-computed threshold instead of taking two
square roots per bot per step (pb_device.hpp PbDevParams::holdV2 / holdF2; particlebot_impl.cuh:809-811 is
`length(vel) < 0.000001f && length(force) < 2 friction gravity`).  The decision is the same for every input iff
the threshold T(c) is the smallest float whose correctly rounded root is >= c; this checks exactly that, on the
CPU, for the constants the shipped configurations produce and for a sweep of others.  (The GPU parity suites then
check whole trajectories bit for bit against the oracle, which keeps the square roots.)"""
import numpy as np
import pytest

from particlerobotsimulations_amd import _capi


def threshold(c):
    return np.float32(_capi.lib().pbHostSqrtThreshold(float(np.float32(c))))


def neighbours(x):
    bits = np.array([x], np.float32).view(np.uint32)[0]
    lo = np.array([bits - 1 if bits > 0 else 0], np.uint32).view(np.float32)[0]
    return lo


@pytest.mark.parametrize("c", [1e-6, 2 * 0.4 * 9.8, 2 * 0.3 * 9.81, 2 * (0.4 * 10) * (9.8 * 50), 1e-30, 1e-44, 3e38,
                               1.0, 2.0, 0.5, 1.0000001, 0.99999994, 1.5e-23, 7.7e18])
def test_threshold_is_the_smallest_float_with_root_at_least_c(c):
    c = np.float32(c)
    t = threshold(c)
    assert np.sqrt(t, dtype=np.float32) >= c
    if t > 0:
        assert np.sqrt(neighbours(t), dtype=np.float32) < c


def test_sweep_agrees_with_root_compare():
    rng = np.random.default_rng(11)
    cs = np.exp(rng.uniform(np.log(1e-20), np.log(1e18), 200)).astype(np.float32)
    for c in cs:
        t = threshold(c)
        # values around the threshold and random ones: `sqrt(x) < c` and `x < T(c)` agree
        tb = int(np.array([t], np.float32).view(np.uint32)[0])
        near = np.arange(max(tb - 64, 0), min(tb + 64, 0x7F800000), dtype=np.uint32).view(np.float32)
        far = np.exp(rng.uniform(np.log(1e-38), np.log(1e38), 256)).astype(np.float32)
        x = np.concatenate([near, far, np.array([0.0, np.inf, np.nan], np.float32)])
        with np.errstate(invalid="ignore"):
            assert np.array_equal(np.sqrt(x, dtype=np.float32) < c, x < t)


def test_degenerate_constants():
    # c <= 0 or NaN: length < c never holds, and neither does dot < 0
    for c in (0.0, -1.0, float("nan"), -float("inf")):
        assert threshold(c) == 0.0
    # c = inf: every finite length is below it, inf and NaN are not
    assert threshold(float("inf")) == np.inf
