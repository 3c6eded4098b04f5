"""CPU tests of the ensemble pipeline's host side (csrc/pb_capi.cpp, include/particlebot_ensemble.h): the producer
pool builds members in order with a bounded look-ahead, the consumer takes sub-batches in order, and what a member
looks like does not depend on the sub-batch size or on the number of producer threads (each member draws from its
own private glibc-compatible stream; placement per /root/reference particlebot.cpp:612-748)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "examples", "example_dead_cells.cfg")


def _members(m):
    # nDead > 0 with time_to_dead 0: the dead set is drawn with the placement
    return [f"seed\n{1000 + k}\nnDead\n{5 + k}" for k in range(m)]


def _dry(sub, threads, m=23, dwell=0, common=None):
    from particlerobotsimulations_amd.ensemble import PipelinedEnsemble
    p = PipelinedEnsemble(CFG, _members(m), common or {"nCells": "400"}, sub_batch=sub, host_threads=threads)
    sums, ahead = p.dry_run(dwell)
    p.close()
    return sums, ahead


def test_members_do_not_depend_on_split_or_threads():
    ref, _ = _dry(0, 1)
    assert len(set(ref.tolist())) == len(ref)          # different seeds: different blobs
    for sub, threads in ((1, 1), (4, 3), (8, 8), (5, 2), (23, 4), (100, 2)):
        sums, _ = _dry(sub, threads)
        assert np.array_equal(sums, ref), (sub, threads)


def test_members_are_the_oracles(orc):
    """The same members through the oracle's placement + dead draw (the draw is due at time 0, so the pipeline draws
    it with the placement; the oracle draws it at the top of its first step)."""
    sums, _ = _dry(3, 2, m=5)

    def fnv(*arrays):
        h = 1469598103934665603
        for a in arrays:
            for b in np.ascontiguousarray(a).view(np.uint8).reshape(-1).tolist():
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    for k in range(5):
        P = orc.load_cfg(CFG, nCells=400, seed=1000 + k, nDead=5 + k)
        placed = orc.Sim(P, reset=True)
        pos, rad = placed.get("pos"), placed.get("rad")
        placed.run(1)
        dead = placed.get("dead").astype(np.int32)
        assert dead.sum() == 5 + k
        assert int(sums[k]) == fnv(pos, rad, dead), k
        placed.close()


def test_look_ahead_is_bounded():
    """Producers never claim more than (ahead + 1) = 3 sub-batches beyond what the consumer has taken."""
    for sub, threads in ((2, 4), (4, 8)):
        _, ahead = _dry(sub, threads, m=40, dwell=5, common={"nCells": "150"})
        assert ahead <= 3 * sub, (sub, threads, ahead)
        assert ahead >= sub          # ... and they do run ahead while the consumer dwells


def test_automatic_sub_batch_is_whole_placement_rounds():
    """sub_batch -1: whole rounds of the producer pool (1 ... 8) that bring a sub-batch to ~3e6 bots; same members,
    and the look-ahead bound follows that size."""
    import ctypes as C
    from particlerobotsimulations_amd import host
    auto = host.lib().pbEnsemblePipelineAutoSubBatch
    auto.argtypes, auto.restype = [C.c_uint, C.c_int], C.c_int
    assert auto(100000, 15) == 30       # BASELINE configs[4] under a 16-CPU quota: 2 rounds (the measured optimum)
    assert auto(100000, 31) == 30       # ... with 32 CPUs: 1 round, capped at the ~3e6-bot target ...
    assert auto(100000, 127) == 30      # ... however many producers a big host has (was 127: 12.7e6 bots in flight)
    assert auto(100000, 1) == 8 and auto(100000, 3) == 24
    assert auto(1000000, 15) == 3       # big members: the bot target, not a whole round of the pool
    assert auto(10000000, 15) == 1
    assert auto(500, 15) == 120 and auto(0, 0) == 8
    ref, _ = _dry(0, 1)
    for threads in (1, 3, 5):
        sums, ahead = _dry(-1, threads, dwell=3)
        assert np.array_equal(sums, ref), threads
        assert ahead <= 3 * 8 * threads, (threads, ahead)


def test_lanes_setter_validates_its_argument():
    """pbEnsemblePipelineSetLanes: 1 ... 4 sub-batches stepped at a time (the stepping itself needs a GPU:
    tests/test_gpu_ensemble_pipeline.py); anything else is refused, and the dry-run consumer is unaffected."""
    from particlerobotsimulations_amd.ensemble import PipelinedEnsemble
    for bad in (0, 5, -1):
        with pytest.raises(ValueError):
            PipelinedEnsemble(CFG, _members(4), {"nCells": "40"}, sub_batch=2, host_threads=2, lanes=bad)
    ref, _ = _dry(0, 1)
    p = PipelinedEnsemble(CFG, _members(len(ref)), {"nCells": "400"}, sub_batch=2, host_threads=2, lanes=3)
    sums, _ = p.dry_run()
    p.close()
    assert np.array_equal(sums, ref)


def test_bad_cfg_fails_cleanly():
    from particlerobotsimulations_amd.ensemble import PipelinedEnsemble
    p = PipelinedEnsemble(os.path.join(ROOT, "examples", "no_such.cfg"), _members(3), None, sub_batch=2, host_threads=2)
    with pytest.raises(RuntimeError):
        p.dry_run()
    p.close()


def test_members_of_a_batch_must_agree_on_the_force_kernel(capfd):
    """the batch took pb_force_variant (and pb_rng) from member 0, so a per-member override was silently
    ignored for the others -- rows labelled exact could come from the tolerance kernel.  Refused now, before any
    device call (so the message is testable here)."""
    from particlerobotsimulations_amd.ensemble import LocalEnsemble
    with pytest.raises(RuntimeError):
        LocalEnsemble(CFG, ["seed\n1", "seed\n2\npb_force_variant\n3"], {"nCells": "40"})
    assert "must agree on pb_force_variant" in capfd.readouterr().err
    with pytest.raises(RuntimeError):
        LocalEnsemble(CFG, ["seed\n1\npb_rng\ncurand", "seed\n2"], {"nCells": "40"})
    assert "must agree on pb_force_variant and pb_rng" in capfd.readouterr().err
