"""GPU parity of the PIPELINED ensemble (pbEnsemblePipeline*, csrc/pb_capi.cpp): host
placement of sub-batch k+1 overlapped with device stepping of sub-batch k.  The summary rows and the final states
must not depend on the sub-batch size or on the number of producer threads, and every member must equal its own
stand-alone CPU-oracle run (rows within the centroid's double-vs-float64-mean rounding, states bit for bit).
Reference behaviour per member: main.cpp:354-361 (dump; update) on Particlebot::update (particlebot.cpp:170-300)."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal
from test_gpu_baseline_configs import EX, oracle_member

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,over", [("example_obstacle.cfg", {}),
                                      ("example_dead_cells.cfg", {"nCells": "700", "nDead": "150"})])
def test_rows_and_states_do_not_depend_on_the_split(orc, cfg, over):
    from particlerobotsimulations_amd import ensemble
    seeds = [1000 + k for k in range(16)]
    common = dict({"max_time": "12.6", "dump_interval": "6"}, **over)
    members = [f"seed\n{s}" for s in seeds]
    ref_rows, ref_steps, ref_states = ensemble.run_local(EX(cfg), members, common, final_state=True)
    assert ref_rows.shape[:2] == (16, 4) and ref_steps in (1260, 1261)
    # even split, ragged split, one batch (two spellings), automatic (-1: whole placement rounds that bring a
    # sub-batch to ~3e6 bots, halved because two sub-batches are then stepped at the same time), and explicit splits
    # stepped two and three sub-batches at a time
    import ctypes as C
    from particlerobotsimulations_amd import host
    auto = host.lib().pbEnsemblePipelineAutoSubBatch
    auto.argtypes, auto.restype = [C.c_uint, C.c_int], C.c_int
    nbots = int(over.get("nCells", 500))
    for sub, threads, lanes in ((8, 3, None), (5, 1, None), (32, 4, None), (0, 2, None), (-1, 3, None), (5, 2, 2),
                                (3, 3, 3), (-1, 1, 1)):
        p = ensemble.PipelinedEnsemble(EX(cfg), members, common, sub_batch=sub, host_threads=threads,
                                       keep_final_states=True, lanes=lanes)
        steps = p.run()
        tm = p.timings
        assert steps == ref_steps
        used = max(1, auto(nbots, threads) // 2) if sub == -1 else sub
        nsub = -(-16 // used) if 0 < used < 16 else 1
        assert tm["sub_batches"] == nsub and tm["host_threads"] == threads
        assert tm["sub_batch"] == (used if 0 < used < 16 else 16)
        assert tm["lanes"] == min(lanes if lanes is not None else (2 if sub == -1 else 1), nsub)
        assert tm["wall_s"] > 0 and tm["device_s"] > 0 and tm["placement_cpu_s"] > 0
        rows, states = p.rows, p.final_states()
        p.close()
        assert np.array_equal(rows.view(np.uint32), ref_rows.view(np.uint32)), (sub, threads)
        for k in range(16):
            for key in ("pos", "vel", "rad"):
                assert_bit_equal(states[k][key], ref_states[k][key], f"{cfg} sub {sub} member {k}: {key}")
    # ... and the members are the oracle's
    for k in (0, 7, 15):
        kv = dict(seed=seeds[k], max_time=12.6, dump_interval=6.0)
        kv.update({a: int(b) for a, b in over.items()})
        orows, osim = oracle_member(orc, EX(cfg), kv, 6.0)
        assert np.array_equal(orows[:, 0].astype(np.float32), ref_rows[k, :, 0])
        assert np.abs(orows[:, 1:] - ref_rows[k, :, 1:]).max() < 2e-6
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(ref_states[k][key], osim.get(key), f"{cfg} member {k} vs oracle: {key}")
        osim.close()


def test_max_steps_bounds_every_sub_batch_alike(orc):
    from particlerobotsimulations_amd import ensemble
    members = [f"seed\n{2000 + k}" for k in range(6)]
    common = {"max_time": "1e9", "dump_interval": "6"}
    a = ensemble.PipelinedEnsemble(EX("example.cfg"), members, common, sub_batch=4, host_threads=2)
    b = ensemble.PipelinedEnsemble(EX("example.cfg"), members, common, sub_batch=0, host_threads=2)
    assert a.run(777) == 777 and b.run(777) == 777
    assert np.array_equal(a.rows.view(np.uint32), b.rows.view(np.uint32)) and a.rows.shape[1] == 3  # t = 0, 0.01, 6
    a.close(), b.close()


def test_placement_is_hidden_behind_stepping():
    """The point of the pipeline: with the host slower than it needs to be (ONE producer thread, the reference's
    O(N^1.5) placement of 20 000-bot members) the wall time is close to max(host, device) + one sub-batch's placement,
    not their sum."""
    from particlerobotsimulations_amd import ensemble
    members = [f"seed\n{3000 + k}\nnDead\n{400 * k}" for k in range(12)]
    common = {"nCells": "20000", "light_x": "-20", "light_y": "0", "max_time": "30", "dump_interval": "6"}
    cfg = EX("example_dead_cells.cfg")
    p = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=2, host_threads=1)
    steps = p.run()
    tm = p.timings
    p.close()
    assert steps in (3000, 3001)
    serial = tm["placement_cpu_s"] + tm["upload_s"] + tm["device_s"]
    one_sub = tm["placement_cpu_s"] / 6
    hidden = max(tm["placement_cpu_s"], tm["upload_s"] + tm["device_s"]) + one_sub
    print("pipeline timings", tm, "serial would be", serial)
    assert tm["wall_s"] < 1.25 * hidden + 0.3, (tm, hidden, serial)
    assert tm["wall_s"] < 0.9 * serial or min(tm["placement_cpu_s"], tm["device_s"]) < 0.15 * serial, (tm, serial)


def test_force_variant_key_reaches_the_batched_engine(orc):
    """`pb_force_variant 3` in a member's configuration: the batch of a pipelined ensemble runs the opt-in tolerance
    kernel (throughput form: 4 x 40 000 bots) -- rows within 1e-6 of the exact kernels' rows after 60 steps (they are
    NOT bit-identical: a different kernel really ran), and the default stays the exact kernels (bit-identical rows
    between two default runs)."""
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example_dead_cells.cfg")
    members = [f"seed\n{2000 + k}" for k in range(4)]
    common = {"nCells": "40000", "nDead": "0", "light_x": "-30", "light_y": "0", "max_time": "0.6", "dump_interval": "0.3",
              "pb_placement": "fastblob", "phase_std": "0"}

    def rows(extra):
        p = ensemble.PipelinedEnsemble(cfg, members, dict(common, **extra), sub_batch=0, host_threads=2,
                                       keep_final_states=True)
        p.run()
        r, st = p.rows, p.final_states()
        p.close()
        return r, st
    exact, st_exact = rows({})
    again, _ = rows({})
    fast, st_fast = rows({"pb_force_variant": "3"})
    assert np.array_equal(exact.view(np.uint32), again.view(np.uint32))
    assert exact.shape == fast.shape and np.isfinite(fast).all()
    assert np.abs(fast[:, :, 1:3].astype(np.float64) - exact[:, :, 1:3]).max() <= 1e-6 * np.abs(exact[:, :, 1:3]).max()
    differs = any(not np.array_equal(a["pos"].view(np.uint32), b["pos"].view(np.uint32)) for a, b in zip(st_exact, st_fast))
    assert differs, "pb_force_variant 3 did not change the kernel"
    for a, b in zip(st_exact, st_fast):
        d = np.linalg.norm(a["pos"].astype(np.float64) - b["pos"], axis=1)
        # (60 un-resynchronised steps of a chaotic blob: most bots still agree exactly, the tail has started to drift)
        assert np.median(d) <= 1e-7 and np.quantile(d, 0.99) <= 1e-3 and d.max() <= 60 * 2.5e-4


def test_members_that_share_a_placement_equal_their_stand_alone_oracle_runs(orc):
    """a Cartesian dead-fraction sweep (3 fractions x 2 seeds, the dead set drawn at t = 0 for one
    seed's worth of configuration and at t = 2 for the other) places each seed's blob ONCE; the members that took a copy
    of the placement -- and of the private generator's state after it -- end bit-equal to the oracle placing, drawing
    and stepping each member from scratch, rows included."""
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example_dead_cells.cfg")
    for ttd in ("0", "2"):
        common = {"nCells": "600", "max_time": "6.2", "dump_interval": "6", "time_to_dead": ttd}
        spec = [(seed, nd) for seed in (4100, 4101) for nd in (0, 90, 240)]
        members = [f"seed\n{s}\nnDead\n{nd}" for s, nd in spec]
        p = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=4, host_threads=3, keep_final_states=True)
        steps = p.run()
        tm, rows, states = p.timings, p.rows, p.final_states()
        p.close()
        assert steps in (620, 621) and tm["placements_run"] == 2 and 4 <= tm["placements_shared"] <= 5
        # the one-batch form (pbEnsembleCreate) groups its members the same way: same rows, same states
        lrows, lsteps, lstates = ensemble.run_local(cfg, members, common, final_state=True)
        assert lsteps == steps and np.array_equal(lrows.view(np.uint32), rows.view(np.uint32))
        for k in range(len(spec)):
            for key in ("pos", "vel", "rad"):
                assert_bit_equal(lstates[k][key], states[k][key], f"run_local vs pipeline, member {k}: {key}")
        for k, (s, nd) in enumerate(spec):
            orows, osim = oracle_member(orc, cfg, dict(nCells=600, seed=s, nDead=nd, max_time=6.2, dump_interval=6.0,
                                                       time_to_dead=float(ttd)), 6.0)
            assert np.array_equal(orows[:, 0].astype(np.float32), rows[k, :, 0]), (ttd, k)
            assert np.abs(orows[:, 1:] - rows[k, :, 1:]).max() < 2e-6
            for key in ("pos", "vel", "rad"):
                assert_bit_equal(states[k][key], osim.get(key), f"time_to_dead {ttd} member {k} (seed {s}, nDead {nd}): {key}")
            assert int(osim.get("dead").sum()) == nd
            osim.close()
