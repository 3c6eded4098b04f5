"""GPU tests of the cuRAND-compatible XORWOW phase noise (`pb_rng curand` / pbSimSetRng): the HIP
kernels (per-bot state initialised by binary decomposition of the bot index over the jump table,
Box-Muller with the cached second value) against the oracle's sequential restatement, bit for bit,
through the engine, the batched ensemble, both engines of class Particlebot and a checkpoint resume."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, simparams_from_orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = lambda name: os.path.join(ROOT, "examples", name)
STATE_KEYS = ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r")


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def rng_states(pb, sim, member=0):
    from particlerobotsimulations_amd import _capi
    arr = (_capi.pbRngState * sim.n)()
    _capi.check(_capi.lib().pbSimGetRngStatesOf(sim._h, member, C.cast(arr, C.c_void_p)), "pbSimGetRngStatesOf")
    a = np.frombuffer(arr, dtype=np.uint32).reshape(sim.n, 12)
    return a.copy()


@pytest.mark.parametrize("kind", [1, 2])
def test_device_states_equal_oracle_states(pb, orc, kind):
    """curand_init(seed, i, 0) for 70 000 bots on the device (each bot: its index's set bits select
    jump matrices) == the oracle's chain of 69 999 sequential 2^67-step jumps."""
    n = 70_000
    P = orc.default_params(nCells=n, nDead=0, seed=424242, phase_std=0.6, max_time=1e9, rngKind=kind)
    osim = orc.Sim(P, reset=False)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    from particlerobotsimulations_amd import _capi
    _capi.check(_capi.lib().pbSimSetRng(gsim._h, kind))
    assert gsim.config()["rng"] == kind
    dev = rng_states(pb, gsim)
    ptr = orc.lib().orc_sim_array(osim._h, 9)
    host = np.frombuffer((C.c_uint32 * (8 * n)).from_address(ptr), dtype=np.uint32).reshape(n, 8)
    assert np.array_equal(dev[:, 1:6], host[:, 0:5])       # v[5]
    assert np.array_equal(dev[:, 0], host[:, 5])           # d
    assert (dev[:, 6] == 0).all() and (dev[:, 7] == kind).all()
    assert len({tuple(r) for r in dev[:2000, 1:6]}) == 2000


@pytest.mark.parametrize("kind", [1, 2])
def test_engine_with_xorwow_noise_matches_oracle(pb, orc, kind):
    """Whole simulation with noise: phase updates at steps 0, 300 and 600 (phase_update_interval 3 s):
    the second one consumes the cached Box-Muller value, the third draws a fresh pair -- with a re-sort
    in between (states live in ORIGINAL bot order, slots move)."""
    P = orc.default_params(nCells=3000, nDead=0, seed=77, phase_std=0.6, max_time=1e9, rngKind=kind,
                           phase_update_interval=3.0, sort_interval=2.0)
    osim = orc.Sim(P)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    from particlerobotsimulations_amd import _capi
    _capi.check(_capi.lib().pbSimSetRng(gsim._h, kind))
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    done = 0
    for upto in (1, 250, 301, 650):
        osim.run(upto - done, sort_interval=2.0)
        assert gsim.step(upto - done, sort_interval=2.0) == upto - done
        done = upto
        st = gsim.get_state()
        for k in STATE_KEYS:
            assert_bit_equal(st[k], osim.get(k), f"kind {kind}, step {done}: {k}")
    assert gsim.phase_draws == 3 and gsim.stats()["resorts"] >= 3
    # the noise really is there and really differs from the counter generator's
    P0 = orc.default_params(nCells=3000, nDead=0, seed=77, phase_std=0.6, max_time=1e9, phase_update_interval=3.0,
                            sort_interval=2.0)
    o0 = orc.Sim(P0)
    o0.run(1, sort_interval=2.0)
    o1 = orc.Sim(P)
    o1.run(1, sort_interval=2.0)
    assert not np.array_equal(o0.get("phase"), o1.get("phase"))


def test_phase_draw_counter_rebuilds_states_for_resume(pb, orc):
    """pbSimSetPhaseDraws(k) with an XORWOW generator replays k draws per bot: states equal those of a
    simulation that really went through k phase updates (what loadCheckpoint relies on)."""
    P = orc.default_params(nCells=500, nDead=0, seed=9, phase_std=0.6, max_time=1e9, rngKind=1,
                           phase_update_interval=1.0)
    sp, keep = simparams_from_orc(P)
    from particlerobotsimulations_amd import _capi
    a = pb.Sim(sp, keepalive=keep)
    _capi.check(_capi.lib().pbSimSetRng(a._h, 1))
    osim = orc.Sim(P)
    a.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                dead=osim.get("dead"))
    a.step(250)   # phase updates at t = 0, 1, 2
    assert a.phase_draws == 3
    b = pb.Sim(sp, keepalive=keep)
    _capi.check(_capi.lib().pbSimSetRng(b._h, 1))
    b.phase_draws = 3
    assert np.array_equal(rng_states(pb, a), rng_states(pb, b))


@pytest.mark.parametrize("engine", ["fused", "legacy"])
def test_cfg_key_pb_rng_curand_end_to_end(orc, tmp_path, engine):
    """`pb_rng curand` in a configuration: loader -> class Particlebot (both engines) -> CSV byte for
    byte against the oracle with rngKind = 1, through the phase update at t = 0 (noise ON: the shipped
    examples all have phase_std 0.6)."""
    from particlerobotsimulations_amd import host
    from test_gpu_host import oracle_csv, product_csv
    cfg = EX("example_dead_cells.cfg")
    over = dict(max_time=1.3, testing=1, dump_interval=0.5)
    a, b = str(tmp_path / "orc.csv"), str(tmp_path / f"{engine}.csv")
    osim = oracle_csv(orc, cfg, a, rngKind=1, **over)
    gsim = product_csv(host, cfg, b, engine, pb_rng="curand", **{k: str(v) for k, v in over.items()})
    assert open(a, "rb").read() == open(b, "rb").read()
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(gsim.get(k), osim.get(k), f"{engine} final {k}")
    # and it is a different trajectory from the default generator's
    c = str(tmp_path / "default.csv")
    product_csv(host, cfg, c, engine, **{k: str(v) for k, v in over.items()})
    assert open(c, "rb").read() != open(b, "rb").read()


def test_ensemble_members_with_xorwow(orc):
    """A batch of seeds with `pb_rng curand` in the common overrides: every member's final state equals
    its own oracle run (per-member seed -> per-member generator states)."""
    from particlerobotsimulations_amd import ensemble
    from test_gpu_baseline_configs import oracle_member
    cfg = EX("example_object_transport.cfg")
    common = {"max_time": "12.3", "dump_interval": "6", "pb_rng": "curand"}
    seeds = [31, 32, 33, 34]
    rows, steps, states = ensemble.run_local(cfg, [f"seed\n{s}" for s in seeds], common, final_state=True)
    for k, s in enumerate(seeds):
        orows, osim = oracle_member(orc, cfg, dict(seed=s, max_time=12.3, dump_interval=6.0, rngKind=1), 6.0)
        assert np.abs(orows[:, 1:] - rows[k, :, 1:]).max() < 2e-6
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(states[k][key], osim.get(key), f"seed {s}: {key}")
        osim.close()
