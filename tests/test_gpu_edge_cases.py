"""Edge cases of the fused engine against the oracle, bit for bit: the smallest simulations, sizes
around wave / workgroup / resident-capacity boundaries, every bot dead, full obstacle tables, bots
pushed into the walls, everything in one cell, and every kernel form (per-step 1/2/4/8 lanes per
bot, resident) for each."""
import numpy as np
import pytest

from helpers import assert_bit_equal, jittered_blob, simparams_from_orc
from particlerobotsimulations_amd import _capi

pytestmark = pytest.mark.gpu

STATE_KEYS = ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r")


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def pair_with_state(pb, orc, P, pos, vel, rad, dead=None, wall_half=0.0):
    n = P.nCells
    osim = orc.Sim(P, reset=False)  # no placement: state given explicitly
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, wall_half=wall_half, keepalive=keep)
    phase = np.zeros(n, np.float32)
    dead = np.zeros(n, np.int32) if dead is None else dead
    for name, a in (("pos", pos), ("vel", vel), ("rad", rad), ("phase", phase), ("dead", dead)):
        osim.set(name, a)
    gsim.set_state(pos=pos, vel=vel, rad=rad, phase=phase, dead=dead)
    return osim, gsim


def compare(osim, gsim, what):
    st = gsim.get_state()
    for k in STATE_KEYS:
        assert_bit_equal(st[k], osim.get(k), f"{what}: {k}")


FORMS = [("auto", None, 0), ("lanes1", 1, 1), ("lanes2", 2, 1), ("lanes4", 4, 1), ("lanes8", 8, 1), ("lanes16", 16, 1), ("lanes32", 32, 1),
         ("lanes64", 64, 1), ("resident", None, 2)]


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1023, 1024, 1025])
def test_sizes_around_boundaries(pb, orc, n):
    """n = 1 (no neighbour at all) up to one past the resident kernel's capacity, in every form."""
    rng = np.random.default_rng(1000 + n)
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(3.0, -2.0))
    for name, lanes, resident in FORMS:
        if resident == 2 and n > 1024:
            continue
        P = orc.default_params(nCells=n, nDead=0, seed=5, phase_std=0.4, max_time=1e9, light_x=-4.0, light_y=1.0)
        osim, gsim = pair_with_state(pb, orc, P, pos, vel, rad)
        if lanes is not None:
            gsim.set_lanes_per_bot(lanes)
        gsim.set_resident(resident)
        step = 0
        for k in (1, 2, 60, 1205):
            osim.run(k - step, sort_interval=0.5)
            assert gsim.step(k - step, sort_interval=0.5) == k - step
            step = k
            compare(osim, gsim, f"n={n} {name} step {k}")


def test_all_dead_and_none_dead(pb, orc):
    n = 400
    rng = np.random.default_rng(7)
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(5.0, 0.0))
    for dead in (np.ones(n, np.int32), np.zeros(n, np.int32)):
        P = orc.default_params(nCells=n, nDead=0, seed=5, phase_std=0.0, max_time=1e9)
        osim, gsim = pair_with_state(pb, orc, P, pos, vel, rad, dead=dead)
        osim.run(300)
        gsim.step(300)
        compare(osim, gsim, f"dead={int(dead[0])}")
    # dead bots never actuate: radii unchanged
    assert_bit_equal(gsim.get_state()["rad"] if dead[0] else rad, rad, "radii")


def test_full_obstacle_tables(pb, orc):
    """10 circular + 10 rectangular obstacles (the tables' capacity), bots started on top of them."""
    n = 900
    rng = np.random.default_rng(11)
    pos, vel, rad = jittered_blob(n, 0.17, rng, center=(4.0, 0.0))
    ang = np.linspace(0, 2 * np.pi, 10, endpoint=False)
    P = orc.default_params(nCells=n, nDead=0, seed=5, phase_std=0.0, max_time=1e9, light_x=-5.0, light_y=0.0,
                           n_cir_obstacles=10, x_cir_obs=list(4.0 + 1.5 * np.cos(ang)),
                           y_cir_obs=list(1.5 * np.sin(ang)), r_cir_obs=[0.2 + 0.03 * k for k in range(10)],
                           nobstacles=10, x1obs=[2.0 + 0.4 * k for k in range(10)],
                           x2obs=[2.15 + 0.4 * k for k in range(10)], y1obs=[-2.6 + 0.1 * k for k in range(10)],
                           y2obs=[-2.2 + 0.1 * k for k in range(10)])
    osim, gsim = pair_with_state(pb, orc, P, pos, vel, rad)
    step = 0
    for k in (1, 10, 400):
        osim.run(k - step)
        gsim.step(k - step)
        step = k
        compare(osim, gsim, f"obstacles step {k}")


def test_bots_in_the_walls_and_one_cell(pb, orc):
    """Bots beyond the wall clamp on all four sides (the integrator pulls them back and reflects the
    velocity), and a pile of bots inside one grid cell (a 25-cell stencil with one crowded cell)."""
    rng = np.random.default_rng(13)
    n = 300
    pos = np.empty((n, 2), np.float32)
    pos[:60] = [63.99, 0.0] + rng.uniform(-0.3, 0.3, (60, 2))
    pos[60:120] = [-63.99, 5.0] + rng.uniform(-0.3, 0.3, (60, 2))
    pos[120:180] = [7.0, 63.99] + rng.uniform(-0.3, 0.3, (60, 2))
    pos[180:240] = [-9.0, -63.99] + rng.uniform(-0.3, 0.3, (60, 2))
    pos[240:] = [1.0, 1.0] + rng.uniform(-0.05, 0.05, (60, 2))  # one cell (cell size 0.235)
    vel = (rng.standard_normal((n, 2)) * 0.5).astype(np.float32)
    rad = rng.uniform(0.0775, 0.1175, n).astype(np.float32)
    P = orc.default_params(nCells=n, nDead=0, seed=5, phase_std=0.0, max_time=1e9, light_x=0.0, light_y=0.0)
    for resident in (1, 2):
        osim, gsim = pair_with_state(pb, orc, P, pos.astype(np.float32), vel, rad)
        gsim.set_resident(resident)
        step = 0
        for k in (1, 5, 200):
            osim.run(k - step, sort_interval=0.3)
            gsim.step(k - step, sort_interval=0.3)
            step = k
            compare(osim, gsim, f"walls resident={resident} step {k}")
    assert np.isfinite(gsim.get_state()["pos"]).all()


def test_zero_steps_and_repeated_calls(pb, orc):
    """step(0) is a no-op; 1-step calls equal one batched call (also checked elsewhere for 1000 bots)."""
    n = 200
    rng = np.random.default_rng(17)
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(5.0, 0.0))
    P = orc.default_params(nCells=n, nDead=0, seed=5, phase_std=0.0, max_time=1e9)
    osim, gsim = pair_with_state(pb, orc, P, pos, vel, rad)
    assert gsim.step(0) == 0
    compare(osim, gsim, "after step(0)")
    for _ in range(25):
        assert gsim.step(1) == 1
    osim.run(25)
    compare(osim, gsim, "25 single steps")


def test_huge_arena_with_aliased_cells(pb, orc):
    """Walls at +-2000 (the fast exact forms' limit is 2048) over the default 512^2 grid, whose span
    is only 120 units: cells alias, so the stencil pairs bots that are thousands of units apart --
    including pairs whose x or y coordinates differ by one ulp.  Bit for bit against the oracle."""
    rng = np.random.default_rng(31)
    n = 4000
    pos = rng.uniform(-1990.0, 1990.0, (n, 2)).astype(np.float32)
    # columns of bots sharing x (or differing by an ulp) at aliasing distance: 512 cells * 0.235
    span = np.float32(512 * 0.235)
    base = pos[:200].copy()
    pos[200:400] = base + np.float32([1.0, 0.0]) * span * np.float32(3)
    pos[400:600, 0] = np.nextafter(base[:, 0], np.float32(4000))
    pos[400:600, 1] = base[:, 1] + span * np.float32(5)
    pos = np.clip(pos, -1990, 1990).astype(np.float32)
    vel = (rng.standard_normal((n, 2)) * 0.05).astype(np.float32)
    rad = rng.uniform(0.0775, 0.1175, n).astype(np.float32)
    P = orc.default_params(nCells=n, nDead=0, seed=5, phase_std=0.0, max_time=1e9, light_x=0.0, light_y=0.0,
                           arena_half=2000.0, grid=512)
    assert abs(P.cellSizeX - 0.235) < 0.01  # the default grid geometry, only the walls moved
    for lanes in (1, 8, 16, 64):
        osim, gsim = pair_with_state(pb, orc, P, pos, vel, rad, wall_half=2000.0)
        gsim.set_lanes_per_bot(lanes)
        gsim.set_resident(1)
        step = 0
        for k in (1, 4, 60):
            osim.run(k - step, sort_interval=0.2)
            gsim.step(k - step, sort_interval=0.2)
            step = k
            compare(osim, gsim, f"huge arena lanes={lanes} step {k}")


@pytest.mark.parametrize("trial", range(14))
def test_random_parameter_sets(pb, orc, trial):
    """Deterministic 'fuzz' over the parameter space the examples never visit: spring, dashpot, shear,
    friction, gravity, attraction, radii, actuation period, constraint, constrained contraction on and
    off, all three light_shadow modes behind obstacles, phase noise, payload factors, wall damping --
    every draw stepped through a phase update with frequent re-sorts and compared bit for bit."""
    rng = np.random.default_rng(9000 + trial)
    n = int(rng.choice([60, 130, 300, 700, 1500]))
    payload = trial % 4 == 3
    rmin = float(rng.uniform(0.05, 0.09))
    kw = dict(
        nCells=n, nDead=-1 if payload else 0, seed=int(rng.integers(1, 10**6)), max_time=1e9,
        light_x=float(rng.uniform(-8, 8)), light_y=float(rng.uniform(-8, 8)),
        spring=float(rng.uniform(200, 3000)), damping=float(rng.uniform(0, 30)), shear=float(rng.uniform(0, 60)),
        friction=float(rng.uniform(0.05, 0.9)), gravity=float(rng.uniform(1, 9.81)),
        attraction=float(rng.choice([0.0, 1e-6, 4.8e-5, 1e-3])), boundaryDamping=float(rng.choice([-1.0, -0.5])),
        min_radius=rmin, max_radius=rmin * float(rng.uniform(1.2, 1.8)), rise_period=float(rng.choice([1.0, 2.0, 3.0])),
        Nx=int(rng.integers(2, 8)), constraint=float(rng.uniform(0.1, 2.0)),
        constrained_contraction=int(trial % 2), constraint_contraction=float(rng.uniform(1, 20)),
        phase_std=float(rng.choice([0.0, 0.3, 1.0])), phase_update_interval=12.0,
        light_shadow=int(trial % 3), massFactor=float(rng.uniform(1, 3)), frictionFactor=float(rng.uniform(0.5, 2)),
        attractionFactor=float(rng.uniform(0.1, 1.0)), radFactor=float(rng.uniform(1.0, 2.5)))
    if trial % 3:
        kw.update(n_cir_obstacles=2, x_cir_obs=[2.0, 6.5], y_cir_obs=[0.5, -1.0], r_cir_obs=[0.4, 0.3],
                  nobstacles=1, x1obs=[3.0], x2obs=[3.2], y1obs=[-2.0], y2obs=[-0.6])
    P = orc.default_params(**kw)
    osim = orc.Sim(P, reset=True)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    if trial % 5 == 4:
        gsim.set_lanes_per_bot(1)
    si = float(rng.choice([0.23, 1.7, 180.0]))
    step = 0
    for k in (1, 7, 400, 1210):
        osim.run(k - step, sort_interval=si)
        assert gsim.step(k - step, sort_interval=si) == k - step
        step = k
        compare(osim, gsim, f"trial {trial} n={n} step {k}")


@pytest.mark.parametrize("case", ["payload_obstacles", "wrap_walls"])
def test_sixty_four_bit_offset_sweep_on_a_small_batch(pb, orc, case, monkeypatch):
    """The throughput sweep's 64-bit-offset form (k_force<..., BIG>, used from 2^28 bots: one arena of 10^9
    bots in tests/soak_huge_arena.py) forced onto a few thousand bots: payload factors + obstacles, and the
    x-wrap / aliased cells / wall clamps, bit for bit against the oracle through re-sorts."""
    monkeypatch.setenv("PB_ALLOW_ENV_OVERRIDES", "1")
    monkeypatch.setenv("PB_DEBUG_FORCE_BIG", "1")
    rng = np.random.default_rng(77)
    if case == "payload_obstacles":
        n = 4001
        P = orc.default_params(nCells=n, nDead=-1, seed=3, phase_std=0.6, max_time=1e9, attractionFactor=0.5,
                               massFactor=2.0, radFactor=2.0, n_cir_obstacles=2, x_cir_obs=[2.0, 6.5],
                               y_cir_obs=[0.5, -1.0], r_cir_obs=[0.4, 0.3], nobstacles=1, x1obs=[3.0], x2obs=[3.2],
                               y1obs=[-2.0], y2obs=[-0.6])
        osim = orc.Sim(P, reset=True)
        pos, vel, rad = osim.get("pos"), osim.get("vel"), osim.get("rad")
    else:
        n = 3000
        P = orc.default_params(nCells=n, nDead=0, seed=4, phase_std=0.0, max_time=1e9)
        osim = orc.Sim(P, reset=True)
        pos, vel, rad = jittered_blob(n, 0.158, rng, center=(59.0, -59.5), jitter=0.12)   # at the wall and the grid wrap
        osim.set("pos", pos), osim.set("vel", vel), osim.set("rad", rad)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    gsim.set_lanes_per_bot(1)
    assert gsim.config()["offsets64"] == 1 and gsim.config()["lanes_per_bot"] == 1
    gsim.set_state(pos=pos, vel=vel, rad=rad, phase=osim.get("phase"), dead=osim.get("dead"))
    step = 0
    for k in (1, 30, 260):
        osim.run(k - step, sort_interval=1.0)
        assert gsim.step(k - step, sort_interval=1.0) == k - step
        step = k
        compare(osim, gsim, f"64-bit offsets, {case}, step {k}")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("nan_at", [(0,), (5, 70), (149,), (3, 149)])
def test_phase_update_follows_the_reference_loop_when_positions_are_nan(pb, orc, nan_at, mode):
    """A blown-up simulation: the reference's min-distance loop `min_d = (min_d < dist ? min_d : dist)`
    (particlebot.cpp:217-227) lets a NaN distance replace the running minimum and the next bot's distance
    replace the NaN, i.e. it returns the minimum over the bots AFTER the last NaN one (NaN if that is the
    last bot).  The engine's on-device reduction must give the same phases (found by tests/soak_fuzz.py), and so
    must its host-loop form (pbSimSetMinDistanceMode 1, the fallback for hosts whose libm fails
    tests/test_libm_pin.py)."""
    n = 150
    P = orc.default_params(nCells=n, nDead=0, seed=21, phase_std=0.0, max_time=1e9, light_x=-3.0, light_y=2.0)
    osim = orc.Sim(P, reset=True)
    pos = osim.get("pos")
    # make the bots BEFORE the last NaN index the closest to the light, so that ignoring NaNs would differ
    pos[:max(nan_at)] += np.float32(-2.0)
    for i in nan_at:
        pos[i] = np.nan
    osim.set("pos", pos)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    gsim.set_state(pos=pos, vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"), dead=osim.get("dead"))
    _capi.check(_capi.lib().pbSimSetMinDistanceMode(gsim._h, mode))
    osim.run(1)
    gsim.step(1)
    a, b = gsim.get_state()["phase"], osim.get("phase")
    both_nan = np.isnan(a) & np.isnan(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert_bit_equal(np.where(both_nan, 0, a).astype(np.float32), np.where(both_nan, 0, b).astype(np.float32), "phase")
    if max(nan_at) == n - 1:
        assert np.isnan(b).all()
    else:
        assert np.isfinite(b[[i for i in range(n) if i not in nan_at]]).all()
