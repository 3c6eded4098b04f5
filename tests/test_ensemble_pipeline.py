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


# ---- one placement per distinct blob ---------------------------------------------------------------
def _sweep(order, seeds=4, fractions=6):
    """A Cartesian sweep of nDead under a few seeds, seed-fastest or fraction-fastest (both orders occur: bench's
    configs[4] leg, `particlebot_ensemble --cartesian`)."""
    out = []
    for k in range(seeds * fractions):
        s, f = (k % seeds, k // seeds) if order == "seed-fastest" else (k // fractions, k % fractions)
        out.append(f"seed\n{2000 + s}\nnDead\n{3 + 7 * f}")
    return out


def _dry_sweep(members, sub, threads, share=True, common=None):
    from particlerobotsimulations_amd.ensemble import PipelinedEnsemble
    old = os.environ.get("PB_SHARE_PLACEMENTS")
    os.environ["PB_SHARE_PLACEMENTS"] = "1" if share else "0"
    try:
        p = PipelinedEnsemble(CFG, members, common or {"nCells": "400"}, sub_batch=sub, host_threads=threads)
        sums, _ = p.dry_run(0)
        counts = p.placement_counts()
        p.close()
    finally:
        if old is None:
            del os.environ["PB_SHARE_PLACEMENTS"]
        else:
            os.environ["PB_SHARE_PLACEMENTS"] = old
    return sums, counts


@pytest.mark.parametrize("order", ["seed-fastest", "fraction-fastest"])
def test_members_of_one_seed_share_one_placement_and_stay_bit_identical(order):
    """24 members = 4 seeds x 6 dead fractions: 4 placements are computed, 20 members take a copy of the placed state
    AND of the private generator's state after it (the dead draw continues that stream, particlebot.cpp:178-194) --
    and every member's placed state + dead set is what it is when each member places for itself."""
    members = _sweep(order)
    alone, c0 = _dry_sweep(members, 5, 3, share=False)
    assert c0 == (24, 0)
    assert len(set(alone.tolist())) == 24          # same blob, different dead sets: all different
    for sub, threads in ((5, 3), (0, 1), (24, 8), (1, 2), (7, 5)):
        sums, counts = _dry_sweep(members, sub, threads)
        assert np.array_equal(sums, alone), (order, sub, threads)
        # (a group placed AHEAD by an idle producer is copied by all its members, its first one included: 20 ... 23)
        assert counts[0] == 4 and 20 <= counts[1] <= 23, (order, sub, threads, counts)


def test_shared_placement_members_are_the_oracles(orc):
    """Members that TOOK a placement, against the oracle placing and drawing each of them from scratch."""
    members = _sweep("fraction-fastest", seeds=2, fractions=3)
    sums, counts = _dry_sweep(members, 2, 2)
    assert counts[0] == 2 and 4 <= counts[1] <= 5

    def fnv(*arrays):
        h = 1469598103934665603
        for a in arrays:
            for b in np.ascontiguousarray(a).view(np.uint8).reshape(-1).tolist():
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    for k in range(6):
        s, f = k // 3, k % 3
        P = orc.load_cfg(CFG, nCells=400, seed=2000 + s, nDead=3 + 7 * f)
        placed = orc.Sim(P, reset=True)
        pos, rad = placed.get("pos"), placed.get("rad")
        placed.run(1)
        dead = placed.get("dead").astype(np.int32)
        assert dead.sum() == 3 + 7 * f and int(sums[k]) == fnv(pos, rad, dead), k
        placed.close()


def test_what_is_and_is_not_shared():
    """The key is what the placement READS: members that differ in light position or nDead >= 0 share; members that
    differ in seed, size, radius or payload mode (nDead == -1: the last bot is the payload, placed left of the blob)
    do not."""
    base = "seed\n3000\nnDead\n4"
    same = [base, "seed\n3000\nnDead\n9", "seed\n3000\nnDead\n4\nlight_x\n-7", "seed\n3000\nnDead\n0\ntime_to_dead\n3"]
    _, counts = _dry_sweep(same, 0, 2)
    assert counts == (1, 3)
    differ = [base, "seed\n3001\nnDead\n4", "seed\n3000\nnDead\n4\nnCells\n401", "seed\n3000\nnDead\n-1",
              "seed\n3000\nnDead\n4\nmin_radius\n0.08"]
    sums, counts = _dry_sweep(differ, 0, 2)
    assert counts == (5, 0) and len(set(sums.tolist())) == 5
    # a unique member among shared ones is placed on its own
    mixed = [base, "seed\n3005\nnDead\n4", "seed\n3000\nnDead\n6"]
    sums, counts = _dry_sweep(mixed, 0, 3)
    assert counts[0] == 2 and counts[1] in (1, 2)
    alone, _ = _dry_sweep(mixed, 0, 3, share=False)
    assert np.array_equal(sums, alone)
