"""Stream form 1 (k_force_patch: force variant 3 with one LDS patch per workgroup, csrc/pb_stream.hip) against
stream form 0 (k_force_stream): the same arithmetic in the same per-lane order, so the same bits -- on the bench
lattice, on a crawling random blob with stale cell lists, at the grid's x-wrap (tiles that fall back to global
memory), with a payload and obstacles, and with tiles cut at grid-row ends."""
import numpy as np
import pytest

from helpers import assert_bit_equal, jittered_blob, simparams_from_orc

pytestmark = pytest.mark.gpu
KEYS = ("pos", "vel", "rad", "absForce_r")


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def pair(pb, make):
    sims = []
    for form in (0, 1):
        s = make()
        s.set_lanes_per_bot(1)
        s.set_resident(1)
        s.set_force_variant(3)
        s.set_stream_form(form)
        sims.append(s)
    return sims


def same(a, b, what):
    sa, sb = a.get_state(), b.get_state()
    for k in KEYS:
        assert_bit_equal(sa[k], sb[k], f"{what}: {k}")
    assert np.isfinite(sa["pos"]).all()


def test_bench_lattice(pb):
    import bench
    n = 1_000_000
    a, b = pair(pb, lambda: bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1))
    for steps in (1, 40, 300):
        a.step(steps)
        b.step(steps)
        same(a, b, f"10^6-bot lattice after +{steps} steps")
    assert b.config()["force_variant"] == 3


@pytest.mark.parametrize("case", ["blob", "payload_obstacles", "x_wrap", "fast_crawl"])
def test_blobs(pb, orc, case):
    rng = np.random.default_rng(11)
    init = None
    if case == "blob":
        P = orc.default_params(nCells=20000, nDead=0, seed=4321, light_x=-3.0, light_y=2.0, phase_std=0.0, max_time=1e9)
    elif case == "payload_obstacles":
        P = orc.default_params(nCells=6001, nDead=-1, seed=99, phase_std=0.0, max_time=1e9, light_x=-5.0, light_y=0.0,
                               attractionFactor=0.3, massFactor=1.7, n_cir_obstacles=1, x_cir_obs=[3.9],
                               y_cir_obs=[0.2], r_cir_obs=[0.5], nobstacles=1, x1obs=[5.5], x2obs=[5.7],
                               y1obs=[-0.5], y2obs=[0.5])
    elif case == "x_wrap":
        n = 6000
        P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=80.0, light_y=80.0)
        pos, vel, rad = jittered_blob(n, 0.16, rng, center=(57.0, 61.0))
        init = dict(pos=pos, vel=vel + np.float32(0.3), rad=rad)
    else:  # every bot drifts several cells between re-sorts: current cells far from the filed ones
        n = 12000
        P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=-30.0, light_y=5.0)
        pos, vel, rad = jittered_blob(n, 0.17, rng, center=(3.0, -2.0))
        init = dict(pos=pos, vel=vel + np.float32([1.5, 0.7]), rad=rad)
    osim = orc.Sim(P, reset=init is None)
    if init is None:
        init = dict(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"))
    n = P.nCells
    sp, keep = simparams_from_orc(P)

    def make():
        s = pb.Sim(sp, keepalive=keep)
        s.set_state(phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32), **init)
        return s
    a, b = pair(pb, make)
    for steps in (1, 60, 400):
        a.step(steps)
        b.step(steps)
        same(a, b, f"{case} after +{steps} steps")
