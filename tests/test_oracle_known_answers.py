"""CPU known-answer tests of the oracle's per-branch arithmetic: each
branch of collideSpheres (impl.cuh:541-594), the actuation law (impl.cuh:124-181), the wall clamp
(impl.cuh:53-103), the hash wrap (impl.cuh:106-120) and the stable sort, against values worked out
independently here in numpy float32, operation by operation."""
import ctypes as C

import numpy as np

f = np.float32


def pair(orc, P, pa, pb, va, vb, ra, rb, A):
    force = np.zeros(2, f)
    fa, fr = np.zeros(1, f), np.zeros(1, f)
    orc.lib().orc_collideSpheres(C.byref(P), np.array(pa, f), np.array(pb, f), np.array(va, f), np.array(vb, f),
                                 float(ra), float(rb), float(A), force, fa, fr)
    return force, fa[0], fr[0]


def test_contact_branch(orc):
    P = orc.default_params(nCells=2, nDead=0, seed=1)
    pa, pb = (f(0.0), f(0.0)), (f(0.12), f(0.05))
    va, vb = (f(0.01), f(-0.02)), (f(-0.03), f(0.04))
    ra = rb = f(0.0775)
    force, fa, fr = pair(orc, P, pa, pb, va, vb, ra, rb, P.attraction)
    rx, ry = f(pb[0] - pa[0]), f(pb[1] - pa[1])
    dist = np.sqrt(f(f(rx * rx) + f(ry * ry)))
    reach = f(ra + rb)
    assert dist < reach
    nx, ny = f(rx / dist), f(ry / dist)
    rvx, rvy = f(vb[0] - va[0]), f(vb[1] - va[1])
    vn = f(f(rvx * nx) + f(rvy * ny))
    tvx, tvy = f(rvx - f(vn * nx)), f(rvy - f(vn * ny))
    ks = f(f(-P.spring) * f(reach - dist))
    tx = f(f(f(f(0) + f(ks * nx)) + f(f(P.damping) * rvx)) + f(f(P.shear) * tvx))
    ty = f(f(f(f(0) + f(ks * ny)) + f(f(P.damping) * rvy)) + f(f(P.shear) * tvy))
    assert force[0] == tx and force[1] == ty
    assert fr == np.sqrt(f(f(tx * tx) + f(ty * ty))) and fa == 0
    assert tx < 0  # the spring pushes A away from B


def test_attraction_regimes(orc):
    P = orc.default_params(nCells=2, nDead=0, seed=1)
    A = f(P.attraction)
    r = f(0.0775)
    reach = f(r + r)
    for gap, regime in ((f(0.0004), 1), (f(0.0014), 2), (f(0.05), 3), (f(0.0), 1)):
        d = f(reach + gap)
        force, fa, fr = pair(orc, P, (0, 0), (d, 0), (0, 0), (0, 0), r, r, A)
        dist = np.sqrt(f(f(d * d) + f(0)))
        g = f(dist - reach)
        n = f(d / dist)
        if regime == 1:
            want = f(f(0) + f(f(2.5) * n))
        elif regime == 2:
            i1, i2 = f(0.0009), f(0.0019)
            c = f(f(2.5) + f(f(f(f(A / f(i2 * i2)) - f(2.5)) / f(i2 - i1)) * f(g - i1)))
            want = f(f(0) + f(c * n))
        else:
            want = f(f(0) + f(f(A * n) / f(g * g)))
        assert force[0] == want and force[1] == 0, (regime, force, want)
        assert fa == want and fr == 0
    # continuity of the law at the band edges (the reference's constants make it continuous)
    lo = pair(orc, P, (0, 0), (f(reach + f(0.00189999)), 0), (0, 0), (0, 0), r, r, A)[0][0]
    hi = pair(orc, P, (0, 0), (f(reach + f(0.00190001)), 0), (0, 0), (0, 0), r, r, A)[0][0]
    assert abs(lo - hi) / hi < 1e-2


def test_exactly_touching_is_not_contact(orc):
    """dist == collideDist takes the attraction branch (`dist < collideDist` is strict)."""
    P = orc.default_params(nCells=2, nDead=0, seed=1)
    r = f(0.0775)
    force, fa, fr = pair(orc, P, (0, 0), (f(r + r), 0), (0, 0), (1, 1), r, r, P.attraction)
    assert force[0] == f(2.5) and fr == 0 and fa == f(2.5)


def test_actuation_law(orc):
    P = orc.default_params(nCells=8, nDead=0, seed=1)
    n = 8
    rad0 = np.full(n, 0.0775, f)
    phase = np.array([0, 0, 0, -1.0, 9999999999.0, 0, -3.5, 0], f)
    dead = np.array([0, 0, 0, 0, 0, 1, 0, 0], np.int32)
    absA = np.zeros(n, f)
    absR = np.array([0, 3.0, 100.0, 0, 0, 0, 0, 0], f)
    rad = rad0.copy()
    t, dt = f(1.0), f(0.01)
    orc.lib().orc_updateRad_light_wave(C.byref(P), absA, absR, rad, phase, float(t), float(dt), dead, n)
    # bot 0: free expansion toward min + (max-min)/rise * t1, limited by the torque law
    target = f(f(0.0775) + f(f(f(0.1175) - f(0.0775)) / f(2.0)) * f(1.0))
    dr1 = f(target - f(0.0775))
    torque = min(f(f(f(f(dr1 * f(0.5)) * f(0.0775)) / f(0.1)) / f(0.1175)) / dt, f(0.5))
    dr = f(f(f(f(f(0.1) * f(0.1175)) / f(0.5)) * f(f(torque / f(0.0775)) - f(0))) * dt)
    assert rad[0] == f(f(0.0775) + dr)
    assert rad0[1] < rad[1] < rad[0]          # loaded: slower
    assert rad[2] == rad0[2]                  # contact force above torque/r: blocked
    assert rad[3] == rad0[3]                  # time1 = 0: target == current radius
    assert rad[4] == rad0[4] and rad[5] == rad0[5]  # phase > 1e7, dead
    assert rad[6] == rad0[6]                  # time1 < 0 wraps by 100 periods into the hold window
    # hold window: time1 >= 2*rise_period leaves the radius alone
    rad = np.full(n, 0.09, f)
    orc.lib().orc_updateRad_light_wave(C.byref(P), absA, absR, rad, np.zeros(n, f), 5.0, float(dt), np.zeros(n, np.int32), n)
    assert np.all(rad == f(0.09))
    # contraction phase (rise < time1 < 2*rise), unconstrained: jumps straight to the target
    rad = np.full(n, 0.11, f)
    orc.lib().orc_updateRad_light_wave(C.byref(P), absA, absR, rad, np.zeros(n, f), 3.0, float(dt), np.zeros(n, np.int32), n)
    assert np.all(rad == f(f(0.1175) + f(f(f(f(0.0775) - f(0.1175)) / f(2.0)) * f(3.0 - 2.0))))


def test_wall_clamp_and_hash_wrap_and_sort(orc):
    P = orc.default_params(nCells=4, nDead=0, seed=1)
    pos = np.array([[63.99, 0.0], [-63.99, 0.0], [0.0, 63.99], [0.0, -63.99]], f)
    vel = np.array([[5, 0], [-5, 0], [0, 5], [0, -5]], f)
    rad = np.full(4, 0.1, f)
    orc.lib().orc_integrateSystem(C.byref(P), pos.reshape(-1), vel.reshape(-1), rad, 0.01, 4)
    assert pos[0, 0] == f(f(64.0) - f(0.1)) and vel[0, 0] == -5 and pos[1, 0] == f(f(-64.0) + f(0.1))
    assert pos[2, 1] == f(f(64.0) - f(0.1)) and vel[3, 1] == 5
    # hash: cell = floor((p + 64) / 0.235) & 511; x = -64.1 -> cell -1 -> 511; x = 56.5 -> 512 -> 0
    pts = np.array([[-64.1, -64.0], [56.5, -64.0], [0.0, 0.0], [-63.9, -63.7]], f)
    h, idx = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
    orc.lib().orc_calcHash(C.byref(P), h, idx, pts.reshape(-1), 4)
    cs = f(0.235)
    assert h[0] == 511 and h[1] == int(np.floor(f(f(56.5) + f(64)) / cs)) - 512 == 0
    assert h[2] == 272 * 512 + 272 and h[3] == 1 * 512 + 0 and list(idx) == [0, 1, 2, 3]
    keys = np.array([5, 1, 5, 0, 1, 5], np.uint32)
    vals = np.arange(6, dtype=np.uint32)
    orc.lib().orc_sortParticlebots(keys, vals, 6)
    assert list(keys) == [0, 1, 1, 5, 5, 5] and list(vals) == [3, 1, 4, 0, 2, 5]


def test_zero_Nx_falls_back_to_the_placement_size(orc):
    """particlebot.cpp:772-773: `if(!params.Nx) params.Nx = particlebotConfigSize.x` -- ceil(sqrt(n))
    after CONFIG_RANDOM (:624), 2 * rings after the hexagonal placement (:479).  Unreachable from a
    .cfg (the key `Nx` is too short to be read), reachable through SimParams."""
    P = orc.default_params(nCells=50, nDead=0, seed=3, phase_std=0.0, max_time=1e9, Nx=0)
    s = orc.Sim(P)                      # reset() inside
    P2 = orc.default_params(nCells=50, nDead=0, seed=3, phase_std=0.0, max_time=1e9, Nx=8)
    s2 = orc.Sim(P2)
    s.run(30)
    s2.run(30)
    assert np.array_equal(s.get("rad"), s2.get("rad"))      # ceil(sqrt(50)) = 8: same actuation period
    P3 = orc.default_params(nCells=50, nDead=0, seed=3, phase_std=0.0, max_time=1e9, Nx=5)
    s3 = orc.Sim(P3)
    s3.run(30)
    assert not np.array_equal(s.get("rad"), s3.get("rad"))
