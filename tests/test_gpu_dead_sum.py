"""The throughput form of the force kernel without Sum|F_attr| (pbSimSetForceSums mode 0, the default
when no member has constrained_contraction): absForce_a is a dead value there
(particlebot_kernel_impl.cuh:167-169 is its only reader) and the contact magnitudes are parked in LDS
and rooted after the neighbour sweep.  Everything that has a reader -- positions, velocities, radii,
phases, absForce_r -- must equal the oracle bit for bit exactly as with both sums kept (mode 1), and
mode 1 must still deliver absForce_a."""
import numpy as np
import pytest

from helpers import assert_bit_equal, jittered_blob, simparams_from_orc

pytestmark = pytest.mark.gpu

LIVE_KEYS = ("pos", "vel", "rad", "phase", "absForce_r")


@pytest.fixture(scope="module")
def orc():
    from oracle import orclib
    return orclib


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def nanmasked_equal(a, b, what):
    both = np.isnan(a) & np.isnan(b)  # NaN payloads are not compared
    assert_bit_equal(np.where(both, 0, a).astype(a.dtype), np.where(both, 0, b).astype(b.dtype), what)


def build(pb, orc, P, state, mode, variant=2, big=False, wall_half=0.0):
    osim = orc.Sim(P, reset=True)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, wall_half=wall_half, keepalive=keep)
    gsim.set_lanes_per_bot(1)  # the throughput form, whatever the batch size
    gsim.set_force_variant(variant)
    gsim.set_force_sums(mode)
    full = dict(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                dead=osim.get("dead"))
    full.update(state)
    for k, v in full.items():
        osim.set(k, v)
    gsim.set_state(**full)
    return osim, gsim


def check(osim, gsim, mode, what):
    st = gsim.get_state()
    for key in LIVE_KEYS:
        nanmasked_equal(st[key], osim.get(key), f"{what}: {key}")
    if mode == 1:
        nanmasked_equal(st["absForce_a"], osim.get("absForce_a"), f"{what}: absForce_a")
    else:
        assert st["absForce_a"] is None


def blob_state(n, seed, **kw):
    rng = np.random.default_rng(seed)
    pos, vel, rad = jittered_blob(n, 0.158, rng, center=(0.3, -0.2), jitter=0.12, **kw)
    return dict(pos=pos, vel=vel, rad=rad)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("variant", [1, 2])
def test_blob_with_payload_and_edge_values(pb, orc, mode, variant):
    """Payload factors, bots on the axes (the fast exact forms step aside wave by wave), tiny and -0.0
    coordinates, a coincident pair (NaN), 1-ulp neighbours: fused and un-fused steps."""
    n = 6000
    P = orc.default_params(nCells=n, nDead=-1, seed=5, phase_std=0.0, max_time=1e9, light_x=-5.0, light_y=0.0,
                           attractionFactor=0.5, massFactor=2.0)
    s = blob_state(n, 123)
    pos, vel = s["pos"], s["vel"]
    k = n // 2
    pos[k:k + 40, 0] = 0.0
    pos[k + 40:k + 80, 1] = 0.0
    pos[k + 80:k + 90, 0] = 1e-30
    pos[k + 90:k + 100, 1] = -1e-38
    pos[k + 100:k + 110, 0] = -0.0
    pos[k + 110] = pos[k + 111]
    vel[k + 110] = vel[k + 111]
    pos[k + 112] = pos[k + 113] + np.array([np.spacing(pos[k + 113, 0]), 0], np.float32)
    osim, gsim = build(pb, orc, P, s, mode, variant)
    cfg = gsim.config()
    assert cfg["lanes_per_bot"] == 1 and cfg["dead_sum_form"] == (1 - mode) and cfg["attraction_sums"] == mode
    assert cfg["force_kind"] == variant
    step = 0
    for upto in (1, 2, 5):
        osim.run(upto - step)
        assert gsim.step(upto - step) == upto - step
        step = upto
        check(osim, gsim, mode, f"mode {mode} variant {variant} step {upto}")
    assert np.isnan(osim.get("vel")).any(), "the coincident pair was supposed to produce NaN"


@pytest.mark.parametrize("mode", [0, 1])
def test_crowded_bots_overflow_the_pending_list(pb, orc, mode):
    """More contacts per bot than a lane's pending list holds (8): clusters of 30 nearly coincident bots
    force the mid-sweep flush; the order of the Sum|F_rep| additions must not change."""
    n = 4096
    P = orc.default_params(nCells=n, nDead=0, seed=17, phase_std=0.0, max_time=1e9, light_x=3.0, light_y=1.0)
    s = blob_state(n, 7)
    rng = np.random.default_rng(70)
    for c in range(20):
        at = int(rng.integers(0, n - 40))
        s["pos"][at:at + 30] = s["pos"][at] + rng.uniform(-0.02, 0.02, size=(30, 2)).astype(np.float32)
    osim, gsim = build(pb, orc, P, s, mode)
    assert gsim.config()["dead_sum_form"] == 1 - mode
    step = 0
    for upto in (1, 3, 12):
        osim.run(upto - step)
        assert gsim.step(upto - step) == upto - step
        step = upto
        check(osim, gsim, mode, f"crowded, mode {mode}, step {upto}")
    # (by construction: 30 bots within +-0.02 of each other are all closer than 2 x 0.0775 = r + r)


@pytest.mark.parametrize("mode", [0, 1])
def test_obstacles_noise_resorts_and_long_window(pb, orc, mode):
    """Circle + rectangle obstacles (their contact terms join Sum|F_rep| AFTER the neighbour sweep, so the
    flush has to come first), phase noise, a shortened re-sort interval, 450 steps."""
    n = 3000
    P = orc.default_params(nCells=n, nDead=0, seed=44, phase_std=0.6, max_time=1e9, light_x=-4.0, light_y=2.0,
                           n_cir_obstacles=1, x_cir_obs=[-2.2], y_cir_obs=[0.4], r_cir_obs=[0.5],
                           nobstacles=1, x1obs=[1.5], x2obs=[1.7], y1obs=[-1.0], y2obs=[1.0])
    osim, gsim = build(pb, orc, P, {}, mode)
    step = 0
    for upto in (1, 40, 450):
        osim.run(upto - step, sort_interval=1.3)
        assert gsim.step(upto - step, sort_interval=1.3) == upto - step
        step = upto
        check(osim, gsim, mode, f"obstacles, mode {mode}, step {upto}")
    assert gsim.stats()["resorts"] >= 3


def test_constrained_contraction_keeps_both_sums(pb, orc):
    """With a reader (constrained_contraction = 1) mode 0 maintains absForce_a by itself."""
    n = 5000
    P = orc.default_params(nCells=n, nDead=0, seed=3, phase_std=0.0, max_time=1e9, light_x=2.0, light_y=-1.0,
                           constrained_contraction=1, constraint_contraction=6.0)
    osim, gsim = build(pb, orc, P, blob_state(n, 9), 0)
    cfg = gsim.config()
    assert cfg["attraction_sums"] == 1 and cfg["dead_sum_form"] == 0
    osim.run(300)
    assert gsim.step(300) == 300
    check(osim, gsim, 1, "constrained contraction")


@pytest.mark.parametrize("ndead", [0, -1])
@pytest.mark.parametrize("attraction", [0.0, 3e-12, 2e-9, 5e-7, 1.0e-6, 4e-4])
def test_both_sums_with_weak_attraction_constants(pb, orc, attraction, ndead):
    """Round 5: the both-sums throughput form roots |F_attr| from one v_rsq_f32 + one Newton step without a per-trip
    domain check, which is exact for 0 and for [2^-96, FLT_MAX); pbAttractionMagnitudeSafe admits a batch to that path
    only when every attraction constant a pair can see is 0 or >= 2^-20 (9.5e-7).  Constants below that (and above
    2^-40, so that the rest of the fast forms stay on), exactly at the edge, at the reference's default and 0: every
    array incl. absForce_a bit-identical to the oracle, with the payload's factors squaring the constant down."""
    n = 5000
    P = orc.default_params(nCells=n, nDead=ndead, seed=21, phase_std=0.0, max_time=1e9, light_x=-3.0, light_y=0.5,
                           attraction=attraction, attractionFactor=0.25, massFactor=2.0)
    osim, gsim = build(pb, orc, P, blob_state(n, 31), 1)
    cfg = gsim.config()
    assert cfg["lanes_per_bot"] == 1 and cfg["attraction_sums"] == 1 and cfg["dead_sum_form"] == 0
    step = 0
    for upto in (1, 2, 30):
        osim.run(upto - step)
        assert gsim.step(upto - step) == upto - step
        step = upto
        check(osim, gsim, 1, f"attraction {attraction}, step {upto}")
    assert np.isfinite(osim.get("absForce_a")).all()   # (never all zero: the 2.5 N band does not depend on the constant)


def test_64bit_offset_form(pb, orc, monkeypatch):
    """The 64-bit-offset throughput sweep (batches of 2^28 bots and more) has the dead-sum form too;
    the debug knob runs it on a small batch."""
    monkeypatch.setenv("PB_ALLOW_ENV_OVERRIDES", "1")
    monkeypatch.setenv("PB_DEBUG_FORCE_BIG", "1")
    n = 5000
    P = orc.default_params(nCells=n, nDead=0, seed=8, phase_std=0.0, max_time=1e9, light_x=-3.0, light_y=0.5)
    osim, gsim = build(pb, orc, P, blob_state(n, 10), 0)
    cfg = gsim.config()
    assert cfg["offsets64"] == 1 and cfg["dead_sum_form"] == 1
    osim.run(25)
    assert gsim.step(25) == 25
    check(osim, gsim, 0, "64-bit offsets")


def test_switching_modes_in_mid_run(pb, orc):
    """Mode 1 switched on in mid-run: absForce_a is valid from the next step on; switching it off again
    changes nothing that has a reader."""
    n = 4000
    P = orc.default_params(nCells=n, nDead=0, seed=12, phase_std=0.0, max_time=1e9, light_x=-3.0, light_y=0.5)
    osim, gsim = build(pb, orc, P, blob_state(n, 11), 0)
    osim.run(20)
    gsim.step(20)
    check(osim, gsim, 0, "before")
    gsim.set_force_sums(1)
    osim.run(1)
    gsim.step(1)
    check(osim, gsim, 1, "one step after switching on")
    gsim.set_force_sums(0)
    osim.run(30)
    gsim.step(30)
    check(osim, gsim, 0, "after switching off")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("form", ["l2", "l4", "l8", "l16", "l32", "l64", "resident128", "resident256", "resident512", "resident1024"])
def test_multi_lane_and_resident_forms(pb, orc, form, mode):
    """Every lanes-per-bot form of the per-step kernel and all four widths of the resident kernel, with and
    without Sum|F_attr|: payload factors, an obstacle, noise, two re-sorts."""
    n = {"resident128": 120, "resident256": 250, "resident512": 500, "resident1024": 1001}.get(form, 901)
    P = orc.default_params(nCells=n, nDead=-1, seed=61, phase_std=0.6, max_time=1e9, light_x=-3.0, light_y=1.0,
                           attractionFactor=0.5, massFactor=2.0, n_cir_obstacles=1, x_cir_obs=[-1.9], y_cir_obs=[0.3],
                           r_cir_obs=[0.4])
    osim = orc.Sim(P, reset=True)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    if form.startswith("resident"):
        gsim.set_resident(2)
    else:
        gsim.set_resident(1)
        gsim.set_lanes_per_bot(int(form[1:]))
    gsim.set_force_sums(mode)
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    cfg = gsim.config()
    assert cfg["dead_sum_form"] == 1 - mode and cfg["resident"] == int(form.startswith("resident"))
    step = 0
    for upto in (1, 37, 260):
        osim.run(upto - step, sort_interval=1.1)
        assert gsim.step(upto - step, sort_interval=1.1) == upto - step
        step = upto
        check(osim, gsim, mode, f"{form}, mode {mode}, step {upto}")
    if form.startswith("resident"):
        assert gsim.stats()["resident_launches"] > 0
