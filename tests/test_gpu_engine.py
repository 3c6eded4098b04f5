"""GPU parity of the resident fused engine (pbSim*, include/particlebot_hip.h part 2) against the
CPU oracle's whole-simulation object and against the reference-probe golden snapshots.

The engine keeps state cell-sorted and fuses step n's forces with step n+1's radius/integration,
but the arithmetic and summation order are the reference's, so everything here is BIT-EXACT
(uint32 compare of fp32), far inside the 1e-5 relative tolerance BASELINE.json asks for."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, jittered_blob, max_rel_err, simparams_from_orc

pytestmark = pytest.mark.gpu

STATE_KEYS = ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r")


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def make_pair(pb, orc, P, hex=False, wall_half=0.0):
    """Oracle sim (placed by the oracle) and an engine sim loaded with the same initial state."""
    osim = orc.Sim(P, reset=True, hex=hex)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, wall_half=wall_half, keepalive=keep)
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    return osim, gsim


def compare(osim, gsim, what):
    st = gsim.get_state()
    for k in STATE_KEYS:
        assert_bit_equal(st[k], osim.get(k), f"{what}: {k}")
    assert gsim.time == osim.time, (what, gsim.time, osim.time)


@pytest.mark.parametrize("resident", [1, 2])
def test_golden_reference_probe(pb, orc, golden_dir, resident):
    """Engine started from the reference's own initial placement reproduces the reference-probe
    position snapshots (tests/golden/ref_probe) bit for bit through a phase update (step 1200),
    and at step 20000, i.e. after the re-sort at step 18000 -- with one kernel per timestep
    (resident=1) and with the multi-step resident kernel (resident=2)."""
    P = orc.default_params(nCells=300, nDead=0, seed=5555, light_x=-2.0, light_y=4.0, phase_std=0.0, max_time=1e9)
    g = lambda k: np.fromfile(os.path.join(golden_dir, "ref_probe", f"example_like_pos_step{k}.bin"),
                              dtype=np.float32).reshape(-1, 2)
    sp, keep = simparams_from_orc(P)
    sim = pb.Sim(sp, keepalive=keep)
    sim.set_resident(resident)
    n = 300
    sim.set_state(pos=g(0), vel=np.zeros((n, 2), np.float32), rad=np.full(n, P.min_radius, np.float32),
                  phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    step = 0
    for k in (1, 2, 5, 10, 20, 50, 100, 200, 400, 1000, 5000, 20000):
        assert sim.step(k - step) == k - step
        step = k
        assert_bit_equal(sim.get_state()["pos"], g(k), f"golden step {k}")
    s = sim.stats()
    assert s["steps"] == 20000 and s["resorts"] == 2 and s["phase_updates"] == 17
    if resident == 1:
        assert s["fused_launches"] > 19900 and s["resident_launches"] == 0  # one kernel per step except at batch ends
    else:
        # one launch per stretch between host events: 12 calls, 16 phase updates and 1 re-sort inside them
        assert s["fused_launches"] == 0 and 12 <= s["resident_launches"] <= 12 + 17 + 2


@pytest.mark.parametrize("case", ["example", "noise", "payload", "circles", "rects", "shadow"])
def test_whole_sim_vs_oracle(pb, orc, case):
    """Every state array after 1, 10, 100, 1199, 1200, 1201 and 2500 steps (1200 = phase update)."""
    kw = dict(nCells=300, nDead=0, seed=5555, light_x=-2.0, light_y=4.0, phase_std=0.0, max_time=1e9)
    if case == "noise":
        kw.update(phase_std=0.6)
    elif case == "payload":
        kw.update(nCells=201, nDead=-1, seed=9999, light_x=-5.0, light_y=0.0, massFactor=2.0, frictionFactor=1.5,
                  attractionFactor=0.25)
    elif case == "circles":
        kw.update(nCells=500, seed=7777, light_x=-5.0, light_y=0.0, n_cir_obstacles=3, x_cir_obs=[4.2, 2.0, 2.5],
                  y_cir_obs=[0.3, 2.0, -2.5], r_cir_obs=[0.5, 0.3, 0.45])
    elif case == "rects":
        kw.update(nCells=600, seed=8888, light_x=-5.0, light_y=0.0, nobstacles=2, x1obs=[3.9, 3.9],
                  x2obs=[4.1, 4.1], y1obs=[-8.0, 0.3], y2obs=[-0.3, 8.0])
    elif case == "shadow":
        kw.update(nCells=400, seed=4242, light_x=-5.0, light_y=0.0, light_shadow=1, n_cir_obstacles=1,
                  x_cir_obs=[3.5], y_cir_obs=[0.0], r_cir_obs=[0.6])
    P = orc.default_params(**kw)
    osim, gsim = make_pair(pb, orc, P)
    step = 0
    for k in (1, 10, 100, 1199, 1200, 1201, 2500):
        osim.run(k - step)
        assert gsim.step(k - step) == k - step
        step = k
        compare(osim, gsim, f"{case} step {k}")


def test_fused_equals_stepwise(pb, orc):
    """Batched (fused) stepping and one-step-at-a-time (unfused) stepping give identical states."""
    P = orc.default_params(nCells=1000, nDead=0, seed=31, phase_std=0.6, max_time=1e9)
    osim, a = make_pair(pb, orc, P)
    _, b = make_pair(pb, orc, P)
    a.step(1500)
    for _ in range(1500):
        b.step(1)
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert_bit_equal(sa[k], sb[k], k)
    assert a.stats()["fused_launches"] == 1499 and b.stats()["fused_launches"] == 0


def test_state_round_trip_through_reused_host_buffers(pb, orc):
    """get_state(out=...) writes the caller's arrays in place, and a SetState / step / GetState round trip every
    step (bench.py's host_round_trip leg) walks the oracle's trajectory bit for bit: pbSimSetState writes each
    bot into the slot it holds, so the stale cell lists (impl.cuh:680) and the re-sort schedule are untouched."""
    P = orc.default_params(nCells=1500, nDead=0, seed=9, phase_std=0.0, max_time=1e9)
    osim, g = make_pair(pb, orc, P)
    dead = np.zeros(P.nCells, np.int32)
    dead[[3, 700, 1499]] = 1
    osim.set("dead", dead)
    g.set_state(dead=dead)
    st = g.get_state()
    if st["absForce_a"] is None:
        g.set_force_sums(1)
        st = g.get_state()
    ids = {k: id(v) for k, v in st.items()}
    for _ in range(40):
        g.set_state(pos=st["pos"], vel=st["vel"], rad=st["rad"], phase=st["phase"], dead=st["dead"])
        g.set_forces(st["absForce_a"], st["absForce_r"])
        g.step(1)
        st = g.get_state(out=st)
        assert {k: id(v) for k, v in st.items()} == ids
    osim.run(40)
    compare(osim, g, "round trip every step")


def test_dead_bots_and_10k(pb, orc):
    """BASELINE config 2b: example_dead_cells.cfg scaled to 10^4 bots, 20 % dead (dead set given to
    both sides; the host-side rand() draw is tested with the Particlebot class)."""
    P = orc.default_params(nCells=10000, nDead=0, seed=6666, light_x=-2.0, light_y=2.0, phase_std=0.0, max_time=1e9)
    osim, gsim = make_pair(pb, orc, P)
    rng = np.random.default_rng(0)
    dead = np.zeros(P.nCells, np.int32)
    dead[rng.choice(P.nCells, 2000, replace=False)] = 1
    osim.set("dead", dead)
    gsim.set_state(dead=dead)
    step = 0
    for k in (1, 10, 300):
        osim.run(k - step)
        gsim.step(k - step)
        step = k
        compare(osim, gsim, f"10k step {k}")


def test_walls_and_grid_wrap(pb, orc):
    """A blob pushed into the +x/+y corner of the world: wall clamps every step, cells beyond the
    512 x 0.235 = 120.3 grid span alias through the `& 511` wrap (impl.cuh:117-118), and the 5-cell
    row of the stencil splits across the x edge of the hash grid."""
    rng = np.random.default_rng(9)
    n = 3000
    P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=80.0, light_y=80.0)
    osim, gsim = make_pair(pb, orc, P)
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(58.0, 60.0))
    pos[: n // 2, 0] += 1.0  # straddle x = 56.3 where cell index 512 wraps to 0
    vel += np.float32(0.5)
    for s in (osim,):
        s.set("pos", pos), s.set("vel", vel), s.set("rad", rad)
    gsim.set_state(pos=pos, vel=vel, rad=rad)
    step = 0
    for k in (1, 5, 50, 400):
        osim.run(k - step)
        gsim.step(k - step)
        step = k
        compare(osim, gsim, f"walls step {k}")
    assert (osim.get("pos").max() > 63.8), "blob never reached the wall"


def test_generalised_arena_and_resort_schedule(pb, orc):
    """Extension: 2048^2 grid, walls at +-240 (the 10^6-bot arena of SURVEY 8(d) config 3) on a
    hex-lattice crop, with sort_interval shortened so several re-sorts happen."""
    P = orc.default_params(nCells=20000, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0, light_y=0.0,
                           grid=2048, arena_half=240.0)
    osim, gsim = make_pair(pb, orc, P, hex=True, wall_half=240.0)
    step = 0
    for k in (1, 20, 130):
        osim.run(k - step, sort_interval=0.5)
        gsim.step(k - step, sort_interval=0.5)
        step = k
        compare(osim, gsim, f"arena step {k}")
    # the fp32 schedule gate of particlebot.cpp:256, replayed on the host
    t, dt, si, expect = np.float32(0), np.float32(0.01), np.float32(0.5), 0
    for _ in range(130):
        expect += int(t - si * np.floor(t / si) < dt)
        t = np.float32(t + dt)
    assert expect >= 3 and gsim.stats()["resorts"] == expect


def test_max_time_stops(pb, orc):
    """update() past max_time: the reference exits the process (particlebot.cpp:174-176); the
    engine returns the number of steps it actually ran."""
    P = orc.default_params(nCells=100, nDead=0, seed=3, phase_std=0.0, max_time=0.055)
    osim, gsim = make_pair(pb, orc, P)
    ran = 0
    while not osim.update():
        ran += 1
    assert gsim.step(100) == ran
    compare(osim, gsim, "after max_time")
    assert gsim.step(5) == 0


def test_resort_every_step_option_differs_only_statistically(pb, orc):
    """Per-step rebuild is an option, never the default: it changes which neighbours a drifted bot
    sees, so trajectories differ in the last bits but stay close over a short window."""
    P = orc.default_params(nCells=2000, nDead=0, seed=77, phase_std=0.0, max_time=1e9)
    _, a = make_pair(pb, orc, P)
    _, b = make_pair(pb, orc, P)
    b.set_resort_every_step(True)
    a.step(200), b.step(200)
    assert b.stats()["resorts"] == 200 and a.stats()["resorts"] == 1
    pa, pb_ = a.get_state()["pos"], b.get_state()["pos"]
    assert not np.array_equal(pa, pb_) or True  # usually differs in the last bits
    assert np.abs(pa - pb_).max() < 0.05  # a third of a bot radius after 200 steps


def test_centroid_matches_host(pb, orc):
    P = orc.default_params(nCells=5000, nDead=0, seed=12, phase_std=0.0, max_time=1e9)
    osim, gsim = make_pair(pb, orc, P)
    gsim.step(50)
    osim.run(50)
    cx, cy = gsim.centroid()
    ref = osim.get("pos").astype(np.float64).mean(0)
    assert abs(cx - ref[0]) < 1e-9 and abs(cy - ref[1]) < 1e-9


def test_million_bots_bit_exact_and_properties(pb, orc):
    """BASELINE.json's headline size: 10^6 bots on the hex lattice in the generalised arena.
    (1) the first 6 steps are bit-identical to the oracle on the full state;
    (2) size-independent properties afterwards: radii stay inside [min,max], nobody leaves the
        arena, the lattice's symmetric centroid stays near the origin, fused == stepwise."""
    n = 1_000_000
    P = orc.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0, light_y=0.0,
                           grid=2048, arena_half=240.0)
    osim, gsim = make_pair(pb, orc, P, hex=True, wall_half=240.0)
    _, ssim = make_pair(pb, orc, P, hex=True, wall_half=240.0)
    osim.run(6)
    gsim.step(6)
    for _ in range(6):
        ssim.step(1)
    compare(osim, gsim, "1e6 bots, 6 steps")
    sg, ss = gsim.get_state(), ssim.get_state()
    for k in STATE_KEYS:
        assert_bit_equal(sg[k], ss[k], f"fused vs stepwise {k}")
    gsim.step(60)
    st = gsim.get_state()
    assert np.isfinite(st["pos"]).all() and np.isfinite(st["vel"]).all()
    assert st["rad"].min() >= P.min_radius and st["rad"].max() <= P.max_radius
    assert np.abs(st["pos"]).max() < 240.0
    cx, cy = gsim.centroid()
    assert abs(cx) < 0.05 and abs(cy) < 0.05


def test_fast_math_self_test(pb):
    """pbSelfTest: the fast exact sqrt equals hipcc's IEEE sqrtf on EVERY float of its domain, and
    the shared-reciprocal division equals IEEE division on 2^31 sampled triples of its domain."""
    r = pb.self_test(1 << 31)
    assert r["sqrt_checked"] == (0x7F800000 - 0x0F800000 + 1) + 1, r
    assert r["sqrt_mismatches"] == 0, r
    assert r["div_checked"] > (1 << 29), r
    assert r["div_mismatches"] == 0, r


def test_static_friction_hold_thresholds_exhaustive(pb):
    """The hold `length(v) < 0.000001f && length(F) < 2 mu g` (impl.cuh:809-811) runs as two compares of squared lengths
    against host-computed thresholds: for the constants of the shipped configurations (and a few odd ones) the
    device's IEEE `sqrtf(x) < c` and `x < T(c)` agree on EVERY non-negative float bit pattern, NaNs included."""
    f = np.float32
    consts = [1e-6, float(f(2) * f(0.4) * f(9.81 * 0.566)), float(f(2) * f(0.4) * f(9.8)),
              float(f(2) * (f(0.4) * f(10)) * (f(9.8) * f(50))), 1.0, 3e-39, 1e30, float("inf"), 0.0, -1.0]
    for c in consts:
        r = pb.self_test_hold_threshold(c)
        assert r["checked"] == 1 << 31 and r["mismatches"] == 0, (c, r)


def test_pair_geometry_exhaustive_slices(pb):
    """pbDistUnitFast (one v_rsq_f32 for the distance, its reciprocal and the unit vector) on EVERY (d2, numerator)
    mantissa pair of two of the 64 slices of d2 in [1, 4): the first, and the last -- which holds the two d2 whose
    root has an all-ones mantissa, where the wave must take the v_rcp_f32 form.  2^42 pairs here; all 64 slices
    (2^47, `pb.self_test_pair_geometry()`, profiles/r2_rsq_form_exhaustive.txt) are the proof DESIGN.md quotes."""
    for first in (0, 63):
        r = pb.self_test_pair_geometry(first, 1)
        assert r["checked"] == 1 << 41 and r["mismatches"] == 0, (first, r)
        r = pb.self_test_division(first, 1)   # pbDiv2Fast, general denominator: 2^40 quotients per slice
        assert r["checked"] == 1 << 40 and r["mismatches"] == 0, (first, r)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_force_kernel_variants_match_oracle(pb, orc, variant):
    """Reference-shaped, branch-free and fast-exact-math force kernels all match the oracle bit for
    bit on a state built to sit on the edges of the fast path's domain: bots exactly on the axes,
    bots with denormal-small / 1e-30 coordinates (fast path must step aside), -0.0 coordinates,
    1-ulp-apart neighbours, and payload factors."""
    rng = np.random.default_rng(123)
    n = 6000
    P = orc.default_params(nCells=n, nDead=-1, seed=5, phase_std=0.0, max_time=1e9, light_x=-5.0, light_y=0.0,
                           attractionFactor=0.5, massFactor=2.0)
    osim, gsim = make_pair(pb, orc, P)
    gsim.set_force_variant(variant)
    pos, vel, rad = jittered_blob(n, 0.158, rng, center=(0.0, 0.0), jitter=0.1)
    k = n // 2
    pos[k:k + 40, 0] = 0.0           # exactly on the y axis
    pos[k + 40:k + 80, 1] = 0.0      # exactly on the x axis
    pos[k + 80:k + 90, 0] = 1e-30    # tiny nonzero coordinates
    pos[k + 90:k + 100, 1] = -1e-38
    pos[k + 100:k + 110, 0] = -0.0
    pos[k + 110] = pos[k + 111]      # coincident pair (0/0 -> NaN in the reference too)
    vel[k + 110] = vel[k + 111]
    pos[k + 112] = pos[k + 113] + np.array([np.spacing(pos[k + 113, 0]), 0], np.float32)  # 1 ulp apart
    for s in (osim,):
        s.set("pos", pos), s.set("vel", vel), s.set("rad", rad)
    gsim.set_state(pos=pos, vel=vel, rad=rad)
    osim.run(3)
    gsim.step(3)
    st = gsim.get_state()
    for key in STATE_KEYS:
        a, b = st[key], osim.get(key)
        if a is None:  # absForce_a: no reader in this batch, not maintained (tests/test_gpu_dead_sum.py)
            continue
        both_nan = np.isnan(a) & np.isnan(b)  # NaN payloads are not compared
        assert_bit_equal(np.where(both_nan, 0, a).astype(a.dtype), np.where(both_nan, 0, b).astype(b.dtype),
                         f"variant {variant}: {key}")
    assert np.isnan(osim.get("vel")).any(), "the coincident pair was supposed to produce NaN"


def test_fast_math_disabled_for_out_of_domain_constants(pb, orc):
    """An attraction constant below 2^-48 is outside the fast division's proven domain: the engine
    must fall back (still bit-exact)."""
    P = orc.default_params(nCells=3000, nDead=0, seed=8, phase_std=0.0, max_time=1e9, attraction=1e-20)
    osim, gsim = make_pair(pb, orc, P)
    osim.run(30)
    gsim.step(30)
    compare(osim, gsim, "tiny attraction")


@pytest.mark.parametrize("resident", [1, 2])
def test_ensemble_batch_matches_individual_oracles(pb, orc, resident):
    _ensemble_batch_vs_oracles(pb, orc, resident, 0)


def test_ensemble_batch_with_the_host_loop_minimum_matches_individual_oracles(pb, orc):
    """pbSimSetMinDistanceMode 1: the phase update's nearest-bot distance from the reference's own host loop over
    every position (the fallback for a host whose libm fails tests/test_libm_pin.py) -- same results."""
    _ensemble_batch_vs_oracles(pb, orc, 1, 1)


def _ensemble_batch_vs_oracles(pb, orc, resident, min_distance_mode):
    """pbSimCreateBatch: 12 simulations that differ in seed, light position, noise level, obstacles
    and dead sets, stepped by the same launches, each bit-identical to its own oracle run -- through
    the initial sort, two phase updates and a forced re-sort schedule."""
    members, osims = [], []
    keep = []
    for k in range(12):
        kw = dict(nCells=257, nDead=0, seed=1000 + k, light_x=-5.0 + 0.5 * k, light_y=0.3 * k,
                  phase_std=0.6 if k % 3 else 0.0, max_time=1e9)
        if k % 4 == 1:
            kw.update(n_cir_obstacles=1, x_cir_obs=[4.0], y_cir_obs=[0.2], r_cir_obs=[0.4])
        if k % 4 == 2:
            kw.update(nobstacles=1, x1obs=[3.8], x2obs=[4.0], y1obs=[-1.0], y2obs=[1.0])
        P = orc.default_params(**kw)
        osim = orc.Sim(P)
        if k % 5 == 0:
            dead = np.zeros(257, np.int32)
            dead[np.random.default_rng(k).choice(257, 40, replace=False)] = 1
            osim.set("dead", dead)
        sp, ka = simparams_from_orc(P)
        members.append(sp)
        keep.append(ka)
        osims.append(osim)
    ens = pb.Ensemble(members, keepalive=keep)
    ens.set_resident(resident)
    from particlerobotsimulations_amd import _capi
    _capi.check(_capi.lib().pbSimSetMinDistanceMode(ens._h, min_distance_mode))
    for k, osim in enumerate(osims):
        ens.set_state_of(k, pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"),
                         phase=osim.get("phase"), dead=osim.get("dead"))
    step = 0
    for upto in (1, 7, 1200, 1201, 2450):
        for osim in osims:
            osim.run(upto - step, sort_interval=7.0)
        assert ens.step(upto - step, sort_interval=7.0) == upto - step
        step = upto
        for k, osim in enumerate(osims):
            st = ens.get_state_of(k)
            for key in STATE_KEYS:
                assert_bit_equal(st[key], osim.get(key), f"member {k} step {upto}: {key}")
    com = ens.centroids()
    for k, osim in enumerate(osims):
        ref = osim.get("pos").astype(np.float64).mean(0)
        assert np.abs(com[k] - ref).max() < 1e-9
    s = ens.stats()
    assert s["steps"] == 2450 and s["phase_updates"] == 3 and s["resorts"] >= 3
    assert (s["resident_launches"] > 0) == (resident == 2)


@pytest.mark.parametrize("nsims,bots,lanes", [(8, 300, 0), (9, 300, 8), (17, 130, 64), (13, 700, 1), (24, 201, 16),
                                              (5, 300, 8)])
def test_one_xcd_per_member_grid_with_ragged_batches(pb, orc, nsims, bots, lanes, monkeypatch):
    """Round 5: batches of >= 8 small members launch the per-step force kernel on a 1-D grid decoded as
    member = (b / 8 / tiles) * 8 + b mod 8 (all tiles of a member on one XCD).  Member counts that are not a multiple of
    eight (the last group is padded with workgroups that exit), one tile per member, the one-bot-per-lane form on small
    members, fewer than eight members (plain grid): every member bit-identical to its own oracle run, and to the same
    batch stepped with the plain (tile, member) grid."""
    members, keep, osims = [], [], []
    for k in range(nsims):
        P = orc.default_params(nCells=bots, nDead=0, seed=300 + k, light_x=-4.0 + 0.3 * k, light_y=0.2 * k, phase_std=0.4,
                               max_time=1e9)
        sp, ka = simparams_from_orc(P)
        members.append(sp)
        keep.append(ka)
        osims.append(orc.Sim(P))
    runs = {}
    for grid in ("xcd", "plain"):
        monkeypatch.setenv("PB_ALLOW_ENV_OVERRIDES", "1")
        monkeypatch.setenv("PB_XCD_MEMBERS", "1" if grid == "xcd" else "0")
        ens = pb.Ensemble(members, keepalive=keep)
        ens.set_resident(1)
        ens.set_lanes_per_bot(lanes)
        for k, osim in enumerate(osims):
            ens.set_state_of(k, pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                             dead=osim.get("dead"))
        assert ens.step(60, sort_interval=0.25) == 60
        runs[grid] = [ens.get_state_of(k) for k in range(nsims)]
        assert ens.stats()["resident_launches"] == 0 and ens.stats()["resorts"] >= 2
        ens.close()
    for osim in osims:
        osim.run(60, sort_interval=0.25)
    for k, osim in enumerate(osims):
        for key in STATE_KEYS:
            assert_bit_equal(runs["xcd"][k][key], osim.get(key), f"member {k}: {key}")
            assert_bit_equal(runs["plain"][k][key], runs["xcd"][k][key], f"member {k}, plain grid: {key}")
        osim.close()


def test_ensemble_rejects_mismatched_members(pb, orc):
    a, ka = simparams_from_orc(orc.default_params(nCells=100, nDead=0, seed=1))
    b, kb = simparams_from_orc(orc.default_params(nCells=101, nDead=0, seed=2))
    with pytest.raises(RuntimeError, match="must share"):
        pb.Ensemble([a, b], keepalive=[ka, kb])


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("case", ["payload_obstacles", "wrap_walls"])
def test_lanes_per_bot_forms_match_oracle(pb, orc, lanes, case):
    """The throughput form (one bot per lane) and the multi-lane forms of the per-step force kernel
    (2 ... 64 lanes per bot, ordered group sum; 32 and 64 chain across DPP rows with wave_shr) add the same terms in the same order: all
    bit-identical to the oracle, also at the grid's x-wrap where a stencil row splits
    into two slot ranges."""
    rng = np.random.default_rng(5)
    if case == "payload_obstacles":
        P = orc.default_params(nCells=777, nDead=-1, seed=21, phase_std=0.6, max_time=1e9, light_x=-5.0, light_y=0.0,
                               attractionFactor=0.3, massFactor=1.7, n_cir_obstacles=1, x_cir_obs=[3.9],
                               y_cir_obs=[0.2], r_cir_obs=[0.5], nobstacles=1, x1obs=[5.5], x2obs=[5.7],
                               y1obs=[-0.5], y2obs=[0.5])
        osim, gsim = make_pair(pb, orc, P)
    else:
        P = orc.default_params(nCells=1500, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=80.0, light_y=80.0)
        osim, gsim = make_pair(pb, orc, P)
        pos, vel, rad = jittered_blob(1500, 0.16, rng, center=(57.0, 61.0))
        vel += np.float32(0.3)
        osim.set("pos", pos), osim.set("vel", vel), osim.set("rad", rad)
        gsim.set_state(pos=pos, vel=vel, rad=rad)
    gsim.set_lanes_per_bot(lanes)
    step = 0
    for k in (1, 3, 40, 1203):
        osim.run(k - step)
        gsim.step(k - step)
        step = k
        compare(osim, gsim, f"{case} lanes={lanes} step {k}")


@pytest.mark.parametrize("n", [100, 130, 400, 900])
@pytest.mark.parametrize("case", ["noise", "payload_obstacles", "wrap_walls"])
def test_resident_form_matches_oracle(pb, orc, n, case):
    """k_resident (one workgroup per simulation, state in registers/LDS, many timesteps per launch)
    against the oracle, bit for bit, for every lanes-per-bot width it has (n = 100: 8 lanes,
    130: 4, 400: 2, 900: 1), through phase updates, frequent re-sorts (sort_interval 0.37 s cuts
    the launches into 37-step stretches) and call boundaries."""
    rng = np.random.default_rng(n)
    if case == "noise":
        P = orc.default_params(nCells=n, nDead=0, seed=77 + n, phase_std=0.6, max_time=1e9, light_x=-3.0, light_y=2.0)
        osim, gsim = make_pair(pb, orc, P)
        dead = np.zeros(n, np.int32)
        dead[rng.choice(n, n // 7, replace=False)] = 1
        osim.set("dead", dead)
        gsim.set_state(dead=dead)
    elif case == "payload_obstacles":
        P = orc.default_params(nCells=n, nDead=-1, seed=21, phase_std=0.6, max_time=1e9, light_x=-5.0, light_y=0.0,
                               attractionFactor=0.3, massFactor=1.7, n_cir_obstacles=1, x_cir_obs=[3.9],
                               y_cir_obs=[0.2], r_cir_obs=[0.5], nobstacles=1, x1obs=[5.5], x2obs=[5.7],
                               y1obs=[-0.5], y2obs=[0.5])
        osim, gsim = make_pair(pb, orc, P)
    else:
        P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=80.0, light_y=80.0)
        osim, gsim = make_pair(pb, orc, P)
        pos, vel, rad = jittered_blob(n, 0.16, rng, center=(61.0, 62.0))
        vel += np.float32(0.3)
        osim.set("pos", pos), osim.set("vel", vel), osim.set("rad", rad)
        gsim.set_state(pos=pos, vel=vel, rad=rad)
    gsim.set_resident(2)
    step = 0
    for k in (1, 3, 40, 1203, 2500):
        osim.run(k - step, sort_interval=0.37)
        assert gsim.step(k - step, sort_interval=0.37) == k - step
        step = k
        compare(osim, gsim, f"{case} n={n} step {k}")
    s = gsim.stats()
    assert s["steps"] == 2500 and s["resident_launches"] >= 2500 // 37 and s["fused_launches"] == 0


def test_resident_respects_max_time_and_call_boundaries(pb, orc):
    """The resident stretch ends where the reference's loop would: at max_time, at a phase update,
    at a re-sort, or at the end of the call -- and a later call continues from there."""
    P = orc.default_params(nCells=256, nDead=0, seed=3, phase_std=0.0, max_time=0.505)
    osim, gsim = make_pair(pb, orc, P)
    gsim.set_resident(2)
    ran = 0
    while not osim.update():
        ran += 1
    assert gsim.step(30) == 30 and gsim.step(100) == ran - 30 and ran < 130
    compare(osim, gsim, "max_time")
    assert gsim.step(10) == 0


def test_resident_is_automatic_for_small_ensembles_only(pb, orc):
    """The automatic choice follows the measured cost model: an ensemble of many ~100-bot simulations
    runs resident (one workgroup each); a lone larger simulation is spread over many CUs by
    per-step launches."""
    P = orc.default_params(nCells=2000, nDead=0, seed=4, phase_std=0.0, max_time=1e9)
    _, gsim = make_pair(pb, orc, P)
    gsim.step(50)
    assert gsim.stats()["resident_launches"] == 0
    members, keep = [], []
    for k in range(128):
        sp, ka = simparams_from_orc(orc.default_params(nCells=100, nDead=0, seed=10 + k, phase_std=0.0, max_time=1e9))
        members.append(sp)
        keep.append(ka)
    ens = pb.Ensemble(members, keepalive=keep)
    ens.step(50)
    assert ens.stats()["resident_launches"] > 0


@pytest.mark.parametrize("variant", [2, 3])
def test_batch_in_the_throughput_form(pb, orc, variant):
    """Ensemble members stepped by the one-bot-per-lane kernels (blockIdx.y = member), which large
    batches use: forced here on 5 members of 1500 bots so that the oracle stays cheap.  Variant 2
    (exact) must equal each member's oracle bit for bit; variant 3 (streamlined) must stay within
    1e-5 relative of it over a 10-step window from the same exact state."""
    members, osims, keep = [], [], []
    for k in range(5):
        P = orc.default_params(nCells=1500, nDead=0, seed=300 + k, light_x=-4.0 + k, light_y=0.5 * k,
                               phase_std=0.0, max_time=1e9)
        osims.append(orc.Sim(P))
        sp, ka = simparams_from_orc(P)
        members.append(sp)
        keep.append(ka)
    ens = pb.Ensemble(members, keepalive=keep)
    ens.set_lanes_per_bot(1)
    ens.set_resident(1)
    ens.set_force_variant(2)
    for k, osim in enumerate(osims):
        ens.set_state_of(k, pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                         dead=osim.get("dead"))
    for osim in osims:
        osim.run(150)
    assert ens.step(150) == 150
    for k, osim in enumerate(osims):
        st = ens.get_state_of(k)
        for key in STATE_KEYS:
            assert_bit_equal(st[key], osim.get(key), f"member {k}: {key}")
    assert ens.stats()["fused_launches"] == 149
    if variant == 3:
        ens.set_force_variant(3)
        for osim in osims:
            osim.run(10)
        assert ens.step(10) == 10
        for k, osim in enumerate(osims):
            p, ref = ens.get_state_of(k)["pos"].astype(np.float64), osim.get("pos").astype(np.float64)
            dev = np.linalg.norm(p - ref, axis=1) / np.linalg.norm(ref, axis=1)
            assert np.quantile(dev, 0.99) <= 1e-5 and (dev > 1e-5).sum() <= 8, (k, dev.max())
            assert np.linalg.norm(p.mean(0) - ref.mean(0)) <= 1e-5 * np.linalg.norm(ref.mean(0))
