"""GPU parity of the reference `extern "C"` device boundary (include/particlebot_hip.h part 1):
each HIP kernel against the CPU oracle on the same seeded inputs, BIT-EXACT (fp32 results compared
as uint32).  Reference kernels: particlebot_kernel_impl.cuh; wrappers: particlebot_cuda.cu."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_bit_equal, jittered_blob, simparams_from_orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def variants(orc):
    base = dict(nCells=2000, nDead=0, seed=11, phase_std=0.0, max_time=1e9)
    return {
        "plain": orc.default_params(**base),
        "payload": orc.default_params(**{**base, "nDead": -1, "attractionFactor": 0.5, "massFactor": 3.0,
                                         "frictionFactor": 2.0}),
        "circles": orc.default_params(**base, n_cir_obstacles=3, x_cir_obs=[1.0, 2.0, 2.5],
                                      y_cir_obs=[0.5, 2.0, -2.5], r_cir_obs=[0.5, 0.3, 0.45]),
        "rects": orc.default_params(**base, nobstacles=2, x1obs=[-1.2, 0.5], x2obs=[-1.0, 1.5],
                                    y1obs=[-8.0, 1.0], y2obs=[-1.0, 1.6]),
        "contract": orc.default_params(**{**base, "constrained_contraction": 1}),
    }


def upload_params(pb, P, wall_half=64.0):
    sp, keep = simparams_from_orc(P)
    pb.legacy.set_parameters(sp, wall_half)
    return sp, keep


@pytest.mark.parametrize("wall", [64.0, 3.0])
def test_integrate(pb, orc, wall):
    """integrateSystem (impl.cuh:53-103) including all four wall clamps."""
    rng = np.random.default_rng(1)
    P = orc.default_params(nCells=5000, nDead=0, seed=1)
    P.wallHalf = wall
    n = P.nCells
    pos = rng.uniform(-wall * 1.02, wall * 1.02, (n, 2)).astype(np.float32)
    vel = (rng.standard_normal((n, 2)) * 5).astype(np.float32)
    rad = rng.uniform(0.0775, 0.1175, n).astype(np.float32)
    keep = upload_params(pb, P, wall)
    dpos, dvel, drad = map(pb.DeviceArray.from_host, (pos, vel, rad))
    pb.legacy.integrate(dpos, dvel, drad, 0.01, n)
    opos, ovel = pos.copy(), vel.copy()
    orc.lib().orc_integrateSystem(C.byref(P), opos.reshape(-1), ovel.reshape(-1), rad, 0.01, n)
    assert_bit_equal(dpos.download(), opos, "pos")
    assert_bit_equal(dvel.download(), ovel, "vel")
    assert (np.abs(opos) >= wall - 0.12).any(), "test input never hit a wall"
    del keep


def test_hash_sort_reorder(pb, orc):
    """calcHash (impl.cuh:446-465) with negative / wrapped cells, the stable sort
    (particlebot_cuda.cu:377-382) and reorderDataAndFindCellStart (impl.cuh:469-538)."""
    rng = np.random.default_rng(2)
    P = orc.default_params(nCells=20000, nDead=0, seed=2)
    n, G = P.nCells, P.numCells
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(5.0, 0.0))
    # a few bots far away: beyond the 120.3-wide grid span (wrap aliasing) and left of the origin
    pos[:50] = rng.uniform(56.0, 64.0, (50, 2)).astype(np.float32)
    pos[50:100] = rng.uniform(-70.0, -60.0, (50, 2)).astype(np.float32)
    keep = upload_params(pb, P)
    dpos, dvel, drad = map(pb.DeviceArray.from_host, (pos, vel, rad))
    dhash, dindex = pb.DeviceArray(n, np.uint32), pb.DeviceArray(n, np.uint32)
    pb.legacy.calc_hash(dhash, dindex, dpos, n)
    ohash, oindex = np.empty(n, np.uint32), np.empty(n, np.uint32)
    orc.lib().orc_calcHash(C.byref(P), ohash, oindex, pos.reshape(-1), n)
    assert_bit_equal(dhash.download(), ohash, "hash")
    assert_bit_equal(dindex.download(), oindex, "index")

    pb.legacy.sort(dhash, dindex, n)
    orc.lib().orc_sortParticlebots(ohash, oindex, n)
    assert_bit_equal(dhash.download(), ohash, "sorted hash")
    assert_bit_equal(dindex.download(), oindex, "sorted index (stability)")
    assert np.all(np.diff(ohash.astype(np.int64)) >= 0)

    dcs, dce = pb.DeviceArray(G, np.uint32, fill=7), pb.DeviceArray(G, np.uint32, fill=7)
    dsp, dsv, dsr = pb.DeviceArray((n, 2)), pb.DeviceArray((n, 2)), pb.DeviceArray(n)
    pb.legacy.reorder(dcs, dce, dsp, dsv, dsr, dhash, dindex, dpos, dvel, drad, n, G)
    ocs, oce = np.full(G, 7, np.uint32), np.full(G, 7, np.uint32)
    osp, osv, osr = np.empty((n, 2), np.float32), np.empty((n, 2), np.float32), np.empty(n, np.float32)
    orc.lib().orc_reorderDataAndFindCellStart(C.byref(P), ocs, oce, osp.reshape(-1), osv.reshape(-1), osr, ohash,
                                              oindex, pos.reshape(-1), vel.reshape(-1), rad, n, G)
    assert_bit_equal(dcs.download(), ocs, "cellStart")
    assert_bit_equal(dce.download(), oce, "cellEnd (untouched cells keep their old value)")
    assert_bit_equal(dsp.download(), osp, "sortedPos")
    assert_bit_equal(dsv.download(), osv, "sortedVel")
    assert_bit_equal(dsr.download(), osr, "sortedRad")
    del keep


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2047, 2048, 2049, 100003])
def test_sort_sizes_and_stability(pb, orc, n):
    """Ragged sizes around the wave (64) and tile (2048) boundaries; many duplicate keys."""
    rng = np.random.default_rng(n)
    P = orc.default_params(nCells=max(n, 1), nDead=0, seed=3)
    keep = upload_params(pb, P)
    keys = rng.integers(0, min(P.numCells, max(2, n // 3)), n, dtype=np.uint32)
    vals = rng.permutation(n).astype(np.uint32)
    dk, dv = pb.DeviceArray.from_host(keys), pb.DeviceArray.from_host(vals)
    pb.legacy.sort(dk, dv, n)
    order = np.argsort(keys, kind="stable")
    assert_bit_equal(dk.download(), keys[order], "keys")
    assert_bit_equal(dv.download(), vals[order], "values")
    del keep


@pytest.mark.parametrize("name", ["plain", "contract"])
def test_update_rad(pb, orc, name):
    """updateRad_light_wave (impl.cuh:124-181): expanding under load / blocked, free and constrained
    contraction, hold window, dead bots, phase > 1e7, negative-time wrap."""
    rng = np.random.default_rng(4)
    P = variants(orc)[name]
    n = P.nCells
    keep = upload_params(pb, P)
    rad = rng.uniform(0.0775, 0.1175, n).astype(np.float32)
    phase = rng.uniform(-30.0, 5.0, n).astype(np.float32)
    phase[:20] = 9999999999.0
    dead = (rng.random(n) < 0.1).astype(np.int32)
    absA = rng.uniform(0.0, 3.0, n).astype(np.float32)
    absR = rng.uniform(0.0, 12.0, n).astype(np.float32)
    absR[rng.random(n) < 0.3] = 0.0
    pos = np.zeros((n, 2), np.float32)
    for time in (0.0, 0.37, 1.99, 2.0, 3.5, 4.0, 7.3, 12.0, 1234.56, -0.5, -900.0):
        drad, dphase, ddead = map(pb.DeviceArray.from_host, (rad, phase, dead))
        dA, dR, dpos = map(pb.DeviceArray.from_host, (absA, absR, pos))
        pb.legacy.update_rad(dpos, dA, dR, drad, dphase, time, 0.01, ddead, n)
        orad = rad.copy()
        orc.lib().orc_updateRad_light_wave(C.byref(P), absA, absR, orad, phase, time, 0.01, dead, n)
        assert_bit_equal(drad.download(), orad, f"rad at time {time}")
    del keep


@pytest.mark.parametrize("shadow", [0, 1, 2])
def test_update_phase(pb, orc, shadow):
    """updatePhase (impl.cuh:264-290) + shadow tests (:184-262) against circles and rectangles."""
    rng = np.random.default_rng(5)
    P = orc.default_params(nCells=4000, nDead=0, seed=5, light_shadow=shadow, light_x=-5.0, light_y=0.3,
                           n_cir_obstacles=2, x_cir_obs=[-1.0, 1.0], y_cir_obs=[0.5, -1.0], r_cir_obs=[0.5, 0.4],
                           nobstacles=2, x1obs=[-2.2, 0.0], x2obs=[-2.0, 0.4], y1obs=[-3.0, 1.0], y2obs=[-0.5, 1.5])
    n = P.nCells
    keep = upload_params(pb, P)
    pos = rng.uniform(-4.0, 6.0, (n, 2)).astype(np.float32)
    phase0 = rng.uniform(-3.0, 0.0, n).astype(np.float32)
    mn, mx = np.zeros(1, np.float32), np.zeros(1, np.float32)
    orc.lib().orc_minmax_light_distance(C.byref(P), pos.reshape(-1), n, mn, mx)
    spacing = np.float32(2.0) * np.float32(P.min_radius)
    dpos, dphase = pb.DeviceArray.from_host(pos), pb.DeviceArray.from_host(phase0)
    pb.legacy.update_phase(dpos, dphase, float(spacing), float(mx[0]), float(mn[0]), n)
    ophase = phase0.copy()
    orc.lib().orc_updatePhase(C.byref(P), pos.reshape(-1), ophase, float(spacing), float(mx[0]), float(mn[0]), n)
    got = dphase.download()
    assert_bit_equal(got, ophase, "phase")
    if shadow:
        assert (ophase == (np.float32(-4.0 * 2.0) if shadow == 1 else np.float32(9999999999.0))).any(), \
            "no bot was shadowed"
    del keep


def test_noise_matches_oracle_rng(pb, orc):
    """add_normal_noise with the build's own counter RNG (cuRAND parity is unpinned):
    GPU and oracle draws are bit-identical and look standard-normal."""
    n = 200000
    P = orc.default_params(nCells=n, nDead=0, seed=987654321)
    keep = upload_params(pb, P)
    from particlerobotsimulations_amd import _capi
    state = pb.DeviceArray(12 * n, np.uint32)  # sizeof(pbRngState) = 48 bytes, as sizeof(curandState)
    assert _capi.lib().pbSetRngKind(0) == 0
    pb.legacy.rng_setup(state, n)
    val = np.zeros(n, np.float32)
    dval = pb.DeviceArray.from_host(val)
    oval = val.copy()
    for draw in range(3):
        pb.legacy.add_noise(state, dval, 0.6, n)
        orc.lib().orc_add_normal_noise(P.seed, draw, oval, 0.6, n)
        assert_bit_equal(dval.download(), oval, f"after draw {draw}")
    # the same boundary with the cuRAND-compatible XORWOW generator (curand_init(seed, i, 0) per bot,
    # curand_normal per call: the pair's second value is cached in the 48-byte state)
    assert _capi.lib().pbSetRngKind(1) == 0 and _capi.lib().pbGetRngKind() == 1
    pb.legacy.rng_setup(state, n)
    dval = pb.DeviceArray.from_host(val)
    for draw in range(3):
        pb.legacy.add_noise(state, dval, 0.6, n)
    z = np.zeros((3, n), np.float32)
    orc.lib().orc_xorwow_normals(1, P.seed, n, 3, z)
    want = val.copy()
    for draw in range(3):
        want = want + (np.float32(0.6) * z[draw]).astype(np.float32)
    assert_bit_equal(dval.download(), want, "xorwow noise through the legacy boundary")
    assert _capi.lib().pbSetRngKind(7) != 0 and _capi.lib().pbSetRngKind(0) == 0
    one = np.zeros(n, np.float32)
    orc.lib().orc_add_normal_noise(P.seed, 7, one, 1.0, n)
    assert abs(one.mean()) < 0.01 and abs(one.std() - 1.0) < 0.01
    assert abs(np.mean(one ** 3)) < 0.03 and abs(np.mean(one ** 4) - 3.0) < 0.1
    del keep


@pytest.mark.parametrize("name", ["plain", "payload", "circles", "rects"])
def test_collide(pb, orc, name):
    """collide (impl.cuh:657-831): contact spring/dashpot/shear, the three attraction regimes,
    circle + rectangle obstacles, payload factors, static/kinetic friction, scatter to original
    index.  Inputs go through the real hash -> sort -> reorder chain first."""
    rng = np.random.default_rng(6)
    P = variants(orc)[name]
    n, G = P.nCells, P.numCells
    keep = upload_params(pb, P)
    pos, vel, rad = jittered_blob(n, 0.158, rng, center=(0.5, 0.6), jitter=0.12)
    absA0 = np.zeros(n, np.float32)
    absR0 = rng.uniform(0, 1, n).astype(np.float32)  # multiplied by 0 inside (impl.cuh:688)
    ohash, oindex = np.empty(n, np.uint32), np.empty(n, np.uint32)
    orc.lib().orc_calcHash(C.byref(P), ohash, oindex, pos.reshape(-1), n)
    orc.lib().orc_sortParticlebots(ohash, oindex, n)
    ocs, oce = np.zeros(G, np.uint32), np.zeros(G, np.uint32)
    osp, osv, osr = np.empty((n, 2), np.float32), np.empty((n, 2), np.float32), np.empty(n, np.float32)
    orc.lib().orc_reorderDataAndFindCellStart(C.byref(P), ocs, oce, osp.reshape(-1), osv.reshape(-1), osr, ohash,
                                              oindex, pos.reshape(-1), vel.reshape(-1), rad, n, G)
    onv, oA, oR = vel.copy(), absA0.copy(), absR0.copy()
    orc.lib().orc_collide(C.byref(P), onv.reshape(-1), oA, oR, osp.reshape(-1), osv.reshape(-1), osr, oindex, ocs,
                          oce, n, 0.01)

    dev = {k: pb.DeviceArray.from_host(v) for k, v in dict(sp=osp, sv=osv, sr=osr, idx=oindex, cs=ocs, ce=oce,
                                                            nv=vel, A=absA0, R=absR0).items()}
    pb.legacy.collide(dev["nv"], dev["A"], dev["R"], dev["sp"], dev["sv"], dev["sr"], dev["idx"], dev["cs"],
                      dev["ce"], n, G, 0.01)
    assert_bit_equal(dev["nv"].download(), onv, "newVel")
    assert_bit_equal(dev["A"].download(), oA, "absForce_a")
    assert_bit_equal(dev["R"].download(), oR, "absForce_r")
    assert (oR > 0).mean() > 0.3 and (oA > 0).mean() > 0.9, "input did not exercise both force regimes"
    del keep
