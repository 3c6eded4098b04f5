"""GPU parity of the BASELINE.json configurations that had no `-m gpu` oracle test of their own
(VERDICT round 1): the bench headline workload itself (configs[2]: 10^6 bots, SQUARE lattice,
bench.workload_params), config 4 (example_obstacle.cfg / example_object_transport.cfg as batched
seed ensembles through pbEnsembleRun) and config 5 at member level (example_dead_cells.cfg scaled to
10^5 bots with 0 / 20 % / 40 % dead).  Everything is compared with the CPU oracle BIT FOR BIT on
the state arrays; summary rows (double-precision on-device centroid) within 1e-6 of the oracle's
positions averaged in float64."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, simparams_from_orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = lambda name: os.path.join(ROOT, "examples", name)
STATE_KEYS = ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r")


def dump_due(t, di):
    """particlebot.cpp:309-310 in fp32: a row is written unless t - di*floor(t/di) > 0.01f."""
    f = np.float32
    t, di = f(t), f(di)
    return not (f(t - f(di * f(np.floor(f(t / di))))) > f(0.01))


def oracle_member(orc, cfg, over, di):
    """display() loop (main.cpp:354-361) of one ensemble member on the oracle: summary rows
    (t, COMx, COMy, distance of the COM to the light) at every dump time, then the final state."""
    P = orc.load_cfg(cfg, **over)
    sim = orc.Sim(P)
    rows = []
    while True:
        if dump_due(sim.time, di):
            c = sim.view("pos").astype(np.float64).mean(0)
            rows.append((sim.time, c[0], c[1], np.hypot(c[0] - P.light_x, c[1] - P.light_y)))
        if sim.update():
            break
    return np.array(rows), sim


def test_bench_headline_workload_matches_oracle(orc):
    """bench.py's `value` workload, exactly as bench.make_sim builds it (10^6 bots on the square
    lattice at pitch 0.155, 2048^2 grid, walls +-240, force variant 2 pinned), for 32 steps from
    t = 0: the step-0 phase update and sort, the un-fused first step and 31 fused launches.  Every
    state array equals the oracle's bit for bit at steps 1, 12 and 32."""
    import bench
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    n = 1_000_000
    P = orc.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0, light_y=0.0,
                           grid=2048, arena_half=240.0)
    # the oracle's parameter block is the bench's
    sp_bench, _k1 = bench.workload_params(n, seed=1)
    sp_orc, _k2 = simparams_from_orc(P)
    for name in ("nCells", "nDead", "gravity", "spring", "damping", "shear", "attraction", "friction", "min_radius",
                 "max_radius", "rise_period", "light_x", "light_y", "phase_std", "numCells", "Nx", "constraint"):
        assert getattr(sp_bench, name) == getattr(sp_orc, name), name
    for name in ("gridSize", "cellSize", "worldOrigin"):
        a, b = getattr(sp_bench, name), getattr(sp_orc, name)
        assert (a.x, a.y) == (b.x, b.y), name
    gsim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
    assert gsim.config()["force_variant"] == 2 and gsim.config()["lanes_per_bot"] == 1
    # ... and the HEADLINE form since round 6 (`value`, `roofline`: both magnitude sums, everything collideD writes);
    # gsim above is the line's `default_form`
    gboth = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1, force_sums=1)
    assert gboth.config()["attraction_sums"] == 1 and gboth.config()["dead_sum_form"] == 0
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    osim = orc.Sim(P, reset=True, hex=True)
    osim.set("pos", bench.square_lattice(n, bench.LATTICE_PITCH))
    done = 0
    for upto in (1, 12, 32):
        gsim.step(upto - done)
        osim.run(upto - done)
        done = upto
        st = gsim.get_state()
        for key in STATE_KEYS:
            assert_bit_equal(st[key], osim.get(key), f"bench workload, step {done}: {key}")
        assert gsim.time == osim.time
        gboth.step(upto - gboth.stats()["steps"])
        sb = gboth.get_state()
        for key in tuple(STATE_KEYS) + ("absForce_a",):
            if sb[key] is not None:
                assert_bit_equal(sb[key], osim.get(key), f"bench workload with both sums, step {done}: {key}")
        assert sb["absForce_a"] is not None
    # the lattice is in contact everywhere: a dense, jammed workload (not a dilute gas)
    assert (st["absForce_r"] > 0).mean() > 0.99
    s = gsim.stats()
    assert s["resorts"] == 1 and s["fused_launches"] >= 29
    gsim.close()
    gboth.close()
    osim.close()


@pytest.mark.parametrize("cfg,max_time,di", [("example_obstacle.cfg", "12.6", "6"),
                                             ("example_object_transport.cfg", "12.6", "6")])
def test_config4_seed_ensembles_match_oracle(orc, cfg, max_time, di):
    """BASELINE config 4: the obstacle course (500 bots, 3 circles) and the object transport (200
    bots + payload) as 16-seed Monte-Carlo batches through pbEnsembleRun -- random placement from
    each member's own libc stream, phase noise (phase_std 0.6) at t = 0 and t = 12, 1260 steps --
    against 16 stand-alone oracle runs: 4 summary rows each and the final state bit for bit."""
    from particlerobotsimulations_amd import ensemble
    seeds = [1000 + k for k in range(16)]
    common = {"max_time": max_time, "dump_interval": di}
    rows, steps, states = ensemble.run_local(EX(cfg), [f"seed\n{s}" for s in seeds], common, final_state=True)
    # rows at t = 0, 0.01 (the reference's gate `> 0.01f` lets the second step through), 6 and 12
    assert rows.shape[0] == 16 and rows.shape[1] == 4 and steps in (1260, 1261), (rows.shape, steps)
    for k, s in enumerate(seeds):
        orows, osim = oracle_member(orc, EX(cfg), dict(seed=s, max_time=float(max_time), dump_interval=float(di)),
                                    float(di))
        assert orows.shape == (4, 4)
        assert np.array_equal(orows[:, 0].astype(np.float32), rows[k, :, 0]), (k, orows[:, 0], rows[k, :, 0])
        assert np.abs(orows[:, 1:] - rows[k, :, 1:]).max() < 2e-6, (k, orows, rows[k])
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(states[k][key], osim.get(key), f"{cfg} seed {s}: {key}")
        osim.close()
    # sixteen different blobs
    assert len({tuple(np.round(r[-1, 1:3], 5)) for r in rows}) == 16


def test_config5_dead_fraction_members_match_oracle(orc):
    """BASELINE config 5 at member level: example_dead_cells.cfg scaled to 10^5 bots, light moved
    outside the blob to (-40, 0), dead fractions 0 / 20 % / 40 % (one seed each) as ONE batch --
    the reference's random placement of 10^5 bots, the dead-bot draw at t = 0 from each member's own
    stream, noise, 106 steps of the batched engine -- against three stand-alone oracle runs."""
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example_dead_cells.cfg")
    common = {"nCells": "100000", "light_x": "-40", "light_y": "0", "max_time": "1.05", "dump_interval": "0.5"}
    members = [(0, 4001), (20000, 4002), (40000, 4003)]
    rows, steps, states = ensemble.run_local(cfg, [f"seed\n{s}\nnDead\n{d}" for d, s in members], common,
                                             final_state=True)
    assert rows.shape[:2] == (3, 4) and steps >= 106, (rows.shape, steps)  # rows at t = 0, 0.01, 0.5, 1.0
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    for k, (nd, s) in enumerate(members):
        orows, osim = oracle_member(orc, cfg, dict(nCells=100000, nDead=nd, seed=s, light_x=-40.0, light_y=0.0,
                                                   max_time=1.05, dump_interval=0.5), 0.5)
        assert int(osim.get("dead").sum()) == nd
        assert np.array_equal(orows[:, 0].astype(np.float32), rows[k, :, 0])
        assert np.abs(orows[:, 1:] - rows[k, :, 1:]).max() < 2e-6, (k, orows, rows[k])
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(states[k][key], osim.get(key), f"dead fraction {nd}/100000: {key}")
        # dead bots never actuate: their radius is still the placement radius
        dead = osim.get("dead") != 0
        assert np.all(states[k]["rad"][dead] == np.float32(osim.P.min_radius))
        osim.close()


def test_large_blob_long_window_with_noise_and_resorts(orc):
    """10^5 bots through class Particlebot from the O(N) fastblob placement: cuRAND-compatible XORWOW phase
    noise at every phase update (every 3 s), re-sorts every 5 s with stale lists in between while the
    blob crawls, 20 % of the bots dead from the start (installed in both), 1 300 steps -- the fused
    engine's state equals the oracle's bit for bit.  (fastblob is a product-side generator, so the
    oracle is started from the product's placement; everything after it is the reference's schedule.)"""
    from particlerobotsimulations_amd import host
    cfg = EX("example_dead_cells.cfg")
    n = 100000
    g = host.HostSim(cfg, nCells=str(n), nDead="0", light_x="-40", light_y="0", max_time="1e9",
                     phase_update_interval="3", sort_interval="5", pb_placement="fastblob", pb_rng="curand")
    pos = g.get("pos")
    dead = (np.random.default_rng(5).random(n) < 0.2).astype(np.int32)
    # dead flags go in through the engine handle's state (HostSim has no setter for them): use the C-ABI
    import ctypes as C
    from particlerobotsimulations_amd import _capi
    L = host.lib()
    L.pbHostEngineHandle.restype = C.c_void_p
    L.pbHostEngineHandle.argtypes = [C.c_void_p]
    sim = C.c_void_p(L.pbHostEngineHandle(g._h))
    _capi.check(_capi.lib().pbSimSetState(sim, None, None, None, None, _capi.np_ptr(dead)))
    P = orc.load_cfg(cfg, nCells=n, nDead=0, light_x=-40.0, light_y=0.0, max_time=1e9,
                     phase_update_interval=3.0, sort_interval=5.0, rngKind=1)
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    o = orc.Sim(P, reset=False)
    o.set("pos", pos)
    o.set("rad", g.get("rad"))
    o.set("dead", dead)
    steps = 1300
    assert g.advance(steps) == steps
    o.run(steps, sort_interval=5.0)
    for key in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(g.get(key), o.get(key), f"10^5-bot blob after {steps} steps: {key}")
    moved = np.linalg.norm(g.get("pos") - pos, axis=1)
    assert moved.max() > 0.05 and np.isfinite(g.get("pos")).all()
    # dead bots never actuated; live ones did
    rad = g.get("rad")
    assert np.all(rad[dead != 0] == np.float32(P.min_radius)) and (rad[dead == 0] > np.float32(P.min_radius)).any()


def test_thirty_two_million_bots_match_oracle_and_stay_sane(orc):
    """32 x 10^6 bots (3.3 GB of state, 4096^2 grid, walls +-470): beyond every cache and past 2^24
    slots (32-bit slot / byte-offset arithmetic, radix sort with 25-bit keys).  Two steps equal the
    oracle bit for bit on every state array; 12 more keep the lattice's symmetric centroid and finite
    state; throughput per bot is that of the 10^6 arena or better."""
    import bench
    import particlerobotsimulations_amd as pb
    from particlerobotsimulations_amd import make_params
    pb.legacy.cudaInit(0, None)
    n = 32_000_000
    P = orc.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-460.0, light_y=0.0,
                           grid=4096, arena_half=470.0)
    sp, keep = simparams_from_orc(P)
    sim = pb.Sim(sp, wall_half=470.0, keepalive=keep)
    pos = bench.square_lattice(n, bench.LATTICE_PITCH)
    assert np.abs(pos).max() < 470.0 - 0.2
    sim.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                  phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    osim = orc.Sim(P, reset=True, hex=True)
    osim.set("pos", pos)
    del pos
    sim.step(2)
    osim.run(2)
    st = sim.get_state()
    for key in STATE_KEYS:
        assert_bit_equal(st[key], osim.get(key), f"32e6 bots, step 2: {key}")
    osim.close()
    del st
    done, ms = sim.step_timed(12)
    assert done == 12
    cx, cy = sim.centroid()
    assert abs(cx) < 0.05 and abs(cy) < 0.05
    us_per_million = ms * 1e3 / 12 / 32.0
    assert us_per_million < 200.0, us_per_million   # the 10^6-bot arena costs ~115 us per step; generous for a cold device
    sim.close()
