"""CPU tests (no GPU): the oracle against the reference-probe golden vectors, i.e. the evidence that
the checker itself restates the reference faithfully.  See tests/golden/ref_probe/README.md."""
import json
import os

import numpy as np

from helpers import assert_bit_equal


def _golden_pos(golden_dir, k):
    return np.fromfile(os.path.join(golden_dir, "ref_probe", f"example_like_pos_step{k}.bin"),
                       dtype=np.float32).reshape(-1, 2)


def test_placement_is_bit_exact(orc, golden_dir):
    """Particlebot::reset CONFIG_RANDOM (particlebot.cpp:612-748) driven by glibc rand()."""
    P = orc.default_params(nCells=300, nDead=0, seed=5555, light_x=-2.0, light_y=4.0, phase_std=0.0, max_time=1e9)
    s = orc.Sim(P)
    assert_bit_equal(s.get("pos"), _golden_pos(golden_dir, 0), "initial placement")
    assert np.all(s.get("rad") == np.float32(0.0775))
    assert not s.get("vel").any() and not s.get("dead").any()


def test_trajectory_is_bit_exact_through_phase_updates(orc, golden_dir):
    """Steps 1..5000 of the example-like run: radius actuation, integration, stale cell lists,
    forces, friction, four phase updates (steps 0,1200,2400,3600,4800)."""
    P = orc.default_params(nCells=300, nDead=0, seed=5555, light_x=-2.0, light_y=4.0, phase_std=0.0, max_time=1e9)
    s = orc.Sim(P)
    step = 0
    for k in (1, 2, 5, 10, 20, 50, 100, 200, 400, 1000, 5000):
        assert s.run(k - step)
        step = k
        assert_bit_equal(s.get("pos"), _golden_pos(golden_dir, k), f"step {k}")


def test_probe_centroids(orc, golden_dir):
    """SURVEY 8(c) centroids for the five example-like parameter sets (dead-bot draw, payload mode,
    circle and rectangle obstacles), quoted there to 7 decimals."""
    with open(os.path.join(golden_dir, "ref_probe", "probe_values.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        P = orc.default_params(phase_std=0.0, max_time=1e9, **c["params"])
        s = orc.Sim(P)
        assert s.run(c["steps"])
        pos = s.get("pos")
        com = pos.mean(0, dtype=np.float64)
        assert np.allclose(com, c["com"], rtol=0, atol=6e-8), (c["name"], com, c["com"])
        if "bot0" in c:
            assert np.allclose(pos[0], c["bot0"], rtol=0, atol=6e-8)
