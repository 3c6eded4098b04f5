"""Every row of the force-kernel forms table (csrc/pb_force.hip: the ONE table the dispatch, pbSimGetConfig
and these tests share) against the CPU oracle, bit for bit: a payload + obstacles + noise run and a blob at the
wall / grid wrap, through re-sorts and phase updates.  A form that can be launched is a form that is listed here
(reference: collideD, particlebot_kernel_impl.cuh:657-831)."""
import numpy as np
import pytest

from helpers import assert_bit_equal, jittered_blob, simparams_from_orc

pytestmark = pytest.mark.gpu
KEYS = ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r")
N_FORMS = 17


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def _case(orc, case):
    rng = np.random.default_rng(5)
    if case == "payload_obstacles_noise":
        n = 1501
        P = orc.default_params(nCells=n, nDead=-1, seed=21, phase_std=0.6, max_time=1e9, attractionFactor=0.5,
                               massFactor=2.0, radFactor=2.0, n_cir_obstacles=2, x_cir_obs=[2.0, 6.5],
                               y_cir_obs=[0.5, -1.0], r_cir_obs=[0.4, 0.3], nobstacles=1, x1obs=[3.0], x2obs=[3.2],
                               y1obs=[-2.0], y2obs=[-0.6], phase_update_interval=0.5)
        osim = orc.Sim(P, reset=True)
    else:
        n = 2200
        P = orc.default_params(nCells=n, nDead=0, seed=22, phase_std=0.0, max_time=1e9, phase_update_interval=0.5)
        osim = orc.Sim(P, reset=True)
        pos, vel, rad = jittered_blob(n, 0.158, rng, center=(59.0, -59.5), jitter=0.12)
        osim.set("pos", pos), osim.set("vel", vel), osim.set("rad", rad)
    return P, osim


@pytest.mark.parametrize("case", ["payload_obstacles_noise", "wrap_walls"])
@pytest.mark.parametrize("row", range(N_FORMS))
def test_every_form_of_the_table_matches_the_oracle(pb, orc, row, case):
    forms = pb.force_forms()
    assert len(forms) == N_FORMS, forms
    form = forms[row]
    P, osim = _case(orc, case)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    gsim.select_force_form(row)
    cfg = gsim.config()
    assert cfg["resident"] == 0 and cfg["lanes_per_bot"] == form["lanes_per_bot"], (cfg, form)
    assert cfg["offsets64"] == form["offsets64"] and cfg["attraction_sums"] == form["attraction_sums"], (cfg, form)
    assert (cfg["force_kind"] != 0) == bool(form["flat"]), (cfg, form)
    step = 0
    for k in (1, 2, 60, 130):   # un-fused first step, fused steps, phase updates every 50 steps, a re-sort at 100
        osim.run(k - step, sort_interval=1.0)
        assert gsim.step(k - step, sort_interval=1.0) == k - step
        step = k
        st = gsim.get_state()
        for key in KEYS:
            assert_bit_equal(st[key], osim.get(key), f"form {form} {case} step {k} {key}")
    s = gsim.stats()
    assert s["resident_launches"] == 0 and s["fused_launches"] > 100
    gsim.close()


def test_dead_sum_rows_are_refused_when_a_member_reads_the_sums(pb, orc):
    P = orc.default_params(nCells=500, nDead=0, seed=3, phase_std=0.0, max_time=1e9, constrained_contraction=1)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    for row, form in enumerate(pb.force_forms()):
        if form["attraction_sums"]:
            gsim.select_force_form(row)
        else:
            with pytest.raises(RuntimeError):
                gsim.select_force_form(row)
    gsim.select_force_form(-1)
    assert gsim.config()["attraction_sums"] == 1
    gsim.close()
