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


# ---- obstacles, friction, kick (impl.cuh:701-831), shadow + phase (impl.cuh:184-290), constrained
# ---- contraction (impl.cuh:166-172): one isolated bot, every value worked out here in numpy float32

def collide_one(orc, P, pos, vel, rad, dt=0.01):
    """orc_collide on ONE bot alone in its cell: (new velocity, absForce_a, absForce_r)."""
    pos, vel, radv = np.array([pos], f), np.array([vel], f), np.array([rad], f)
    h, idx = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    orc.lib().orc_calcHash(C.byref(P), h, idx, pos.reshape(-1), 1)
    start = np.full(P.numCells, 0xFFFFFFFF, np.uint32)
    end = np.zeros(P.numCells, np.uint32)
    start[h[0]], end[h[0]] = 0, 1
    newVel, fa, fr = np.zeros(2, f), np.zeros(1, f), np.zeros(1, f)
    orc.lib().orc_collide(C.byref(P), newVel, fa, fr, pos.reshape(-1), vel.reshape(-1), radv, idx, start, end, 1,
                          float(dt))
    return newVel, fa[0], fr[0]


def length(x, y):
    return np.sqrt(f(f(x * x) + f(y * y)))


def friction_and_kick(P, fx, fy, vx, vy, dt):
    """impl.cuh:799-825 for an ordinary bot."""
    fric, grav = f(P.friction), f(P.gravity)
    if length(vx, vy) < f(0.000001) and length(fx, fy) < f(f(f(2.0) * fric) * grav):
        fx = fy = f(0)
    vx, vy = f(vx + f(fx * dt)), f(vy + f(fy * dt))
    k = f(f(fric * grav) * dt)
    sp = length(vx, vy)
    if sp < k:
        return f(0), f(0)
    return f(vx - f(k * f(vx / sp))), f(vy - f(k * f(vy / sp)))


def test_circle_obstacle_contact(orc):
    P = orc.default_params(nCells=1, nDead=0, seed=1, n_cir_obstacles=1, x_cir_obs=[1.0], y_cir_obs=[0.5],
                           r_cir_obs=[0.4])
    px, py, vx, vy, rad, dt = f(0.62), f(0.3), f(0.2), f(-0.1), f(0.1), f(0.01)
    newVel, fa, fr = collide_one(orc, P, (px, py), (vx, vy), rad, dt)
    ox, oy, orad = f(1.0), f(0.5), f(0.4)
    ex, ey = f(px - ox), f(py - oy)
    d2 = f(f(ex * ex) + f(ey * ey))
    reach = f(rad + orad)
    assert d2 < f(reach * reach)
    dx, dy = f(f(-px) + ox), f(f(-py) + oy)
    ln = length(dx, dy)
    dx, dy = f(dx / ln), f(dy / ln)
    rvx, rvy = f(-vx), f(-vy)
    vn = f(f(rvx * dx) + f(rvy * dy))
    tvx, tvy = f(rvx - f(vn * dx)), f(rvy - f(vn * dy))
    k = f(f(f(2.0) * f(P.spring)) * f(reach - np.sqrt(d2)))   # powf(dist_2, 0.5f) -> sqrtf
    tx = f(f(f(f(0) + f(k * f(-dx))) + f(f(P.damping) * rvx)) + f(f(P.shear) * tvx))
    ty = f(f(f(f(0) + f(k * f(-dy))) + f(f(P.damping) * rvy)) + f(f(P.shear) * tvy))
    assert fr == length(tx, ty) and fa == 0
    ev = friction_and_kick(P, f(f(0) + tx), f(f(0) + ty), vx, vy, dt)
    assert newVel[0] == ev[0] and newVel[1] == ev[1]
    assert tx < 0 and ty < 0  # pushed away from the obstacle (which sits up and to the right)


def test_rectangle_obstacle_face_and_corner(orc):
    kw = dict(nCells=1, nDead=0, seed=1, nobstacles=1, x1obs=[1.0], x2obs=[1.4], y1obs=[-0.5], y2obs=[0.5])
    P = orc.default_params(**kw)
    dt = f(0.01)
    # left face: x1 - rad < x < x2 - rad with y inside the wall's span: dir (1, 0), overlap x - x1 + rad
    px, py, vx, vy, rad = f(0.95), f(0.1), f(0.3), f(0.05), f(0.1)
    newVel, fa, fr = collide_one(orc, P, (px, py), (vx, vy), rad, dt)
    dx, dy = f(1.0), f(0.0)
    overlap = f(f(px - f(1.0)) + rad)
    rvx, rvy = f(-vx), f(-vy)
    vn = f(f(rvx * dx) + f(rvy * dy))
    tvx, tvy = f(rvx - f(vn * dx)), f(rvy - f(vn * dy))
    k = f(f(f(-2.0) * f(P.spring)) * overlap)
    tx = f(f(f(f(0) + f(k * dx)) + f(f(P.damping) * rvx)) + f(f(P.shear) * tvx))
    ty = f(f(f(f(0) + f(k * dy)) + f(f(P.damping) * rvy)) + f(f(P.shear) * tvy))
    assert fr == length(tx, ty) and fa == 0
    ev = friction_and_kick(P, f(f(0) + tx), f(f(0) + ty), vx, vy, dt)
    assert newVel[0] == ev[0] and newVel[1] == ev[1] and tx < 0
    # corner (x1, y2): outside both spans, within one radius of the corner
    px, py, vx, vy = f(0.95), f(0.56), f(0.0), f(-0.2)
    newVel, fa, fr = collide_one(orc, P, (px, py), (vx, vy), rad, dt)
    cx, cy = f(px - f(1.0)), f(py - f(0.5))
    c2 = f(f(cx * cx) + f(cy * cy))
    assert c2 < f(rad * rad)
    ln = length(cx, cy)
    dx, dy = f(f(-cx) / ln), f(f(-cy) / ln)
    overlap = f(rad - np.sqrt(c2))
    rvx, rvy = f(-vx), f(-vy)
    vn = f(f(rvx * dx) + f(rvy * dy))
    tvx, tvy = f(rvx - f(vn * dx)), f(rvy - f(vn * dy))
    k = f(f(f(-2.0) * f(P.spring)) * overlap)
    tx = f(f(f(f(0) + f(k * dx)) + f(f(P.damping) * rvx)) + f(f(P.shear) * tvx))
    ty = f(f(f(f(0) + f(k * dy)) + f(f(P.damping) * rvy)) + f(f(P.shear) * tvy))
    assert fr == length(tx, ty)
    ev = friction_and_kick(P, f(f(0) + tx), f(f(0) + ty), vx, vy, dt)
    assert newVel[0] == ev[0] and newVel[1] == ev[1]
    assert tx < 0 and ty > 0  # away from the corner: left and up


def test_friction_hold_and_kinetic_friction(orc):
    P = orc.default_params(nCells=1, nDead=0, seed=1)
    dt = f(0.01)
    # at rest with no force: held (velocity stays exactly zero)
    newVel, fa, fr = collide_one(orc, P, (0.3, 0.3), (0.0, 0.0), 0.1, dt)
    assert newVel[0] == 0 and newVel[1] == 0 and fa == 0 and fr == 0
    # moving freely: kinetic friction takes friction*gravity*dt off the speed, along the motion
    vx, vy = f(0.3), f(-0.4)
    newVel, _, _ = collide_one(orc, P, (0.3, 0.3), (vx, vy), 0.1, dt)
    ev = friction_and_kick(P, f(0), f(0), vx, vy, dt)
    assert newVel[0] == ev[0] and newVel[1] == ev[1]
    k = f(f(f(P.friction) * f(P.gravity)) * dt)
    assert abs(float(length(newVel[0], newVel[1])) - (0.5 - float(k))) < 1e-6
    # slower than one friction decrement: stops dead
    newVel, _, _ = collide_one(orc, P, (0.3, 0.3), (f(0.5) * k, f(0.0)), 0.1, dt)
    assert newVel[0] == 0 and newVel[1] == 0


def test_update_phase_and_shadow_modes(orc):
    # the light at (-5, 0); bot 0 in plain view, bot 1 behind a circular obstacle, bot 2 behind a wall
    base = dict(nCells=3, nDead=0, seed=1, light_x=-5.0, light_y=0.0, n_cir_obstacles=1, x_cir_obs=[-2.0],
                y_cir_obs=[1.0], r_cir_obs=[0.3], nobstacles=1, x1obs=[-2.0], x2obs=[-1.8], y1obs=[-1.5], y2obs=[-0.5])
    pos = np.array([[1.0, 0.0], [1.0, 2.0], [1.0, -2.0]], f)
    spacing, min_d = f(0.155), f(5.9)
    lx, ly = f(-5.0), f(0.0)
    want_visible = []
    for p in pos:
        d = length(f(p[0] - lx), f(p[1] - ly))
        want_visible.append(f(f(f(min_d - d) / spacing) * f(2.0)))          # rise_period 2.0
    for mode in (0, 1, 2):
        P = orc.default_params(light_shadow=mode, **base)
        phase = np.full(3, 123.0, f)
        orc.lib().orc_updatePhase(C.byref(P), pos.reshape(-1).copy(), phase, float(spacing), 0.0, float(min_d), 3)
        assert phase[0] == want_visible[0]
        if mode == 0:
            assert phase[1] == want_visible[1] and phase[2] == want_visible[2]
        elif mode == 1:
            shadow = f(f(-(P.Nx - 1)) * f(2.0))                              # -(Nx-1)*rise_period
            assert phase[1] == shadow and phase[2] == shadow
        else:
            assert phase[1] == f(9999999999.0) and phase[2] == f(9999999999.0)
    # geometry check of the scenario itself (double precision): the sight lines really are blocked
    for p, (c, r) in ((pos[1], ((-2.0, 1.0), 0.3)),):
        a, d = np.array([-5.0, 0.0]), np.array(p, float) - np.array([-5.0, 0.0])
        t = np.dot(np.array(c) - a, d) / np.dot(d, d)
        assert 0 < t < 1 and np.linalg.norm(a + t * d - np.array(c)) < r
    t = (-2.0 + 5.0) / (1.0 + 5.0)   # the ray to bot 2 crosses x = -2 at y = -2 t
    assert -1.5 < -2.0 * t < -0.5


def test_constrained_contraction_branch(orc):
    """impl.cuh:166-172: with constrained_contraction the radius may only shrink as fast as the attraction
    load allows, and never faster than max_radius * dt (not binding here: the target is 0.001 below the radius)."""
    P = orc.default_params(nCells=3, nDead=0, seed=1, constrained_contraction=1, constraint_contraction=10.0)
    n, dt = 3, f(0.01)
    rad0 = np.full(n, 0.0985, f)
    absA = np.array([0.0, 0.05, 50.0], f)       # no load, light load, heavy load
    absR = np.zeros(n, f)
    rad = rad0.copy()
    # time1 = 3.0: contraction phase, target = max + (min-max)/rise * (time1 - rise)
    orc.lib().orc_updateRad_light_wave(C.byref(P), absA, absR, rad, np.zeros(n, f), 3.0, float(dt),
                                       np.zeros(n, np.int32), n)
    target = f(f(0.1175) + f(f(f(f(0.0775) - f(0.1175)) / f(2.0)) * f(f(3.0) - f(2.0))))
    dr1 = f(target - f(0.0985))
    assert dr1 < 0
    cc = f(10.0)
    for i in range(n):
        dr = f(0)
        load = f(absA[i] * f(0.0985))
        if f(f(-cc) * dr1) > load:
            dr = f(f(f(cc * dr1) + load) / cc)
        dr = max(dr, f(f(-f(0.1175)) * dt))
        want = f(f(0.0985) + dr)
        want = min(max(want, f(0.0775)), f(0.1175))
        assert rad[i] == want, (i, rad[i], want)
    assert rad[0] < rad[1] < rad[2] == rad0[2]   # the heavier the attraction load, the less it shrinks


def test_dump_csv_bytes(orc, tmp_path):
    """dumpParticlebot (particlebot.cpp:303-367): header at time 0 ("Seed, %u", the column names with the
    reference's comma/space pattern), then per row "%f," time, positions "%f, %f,", velocities, radii,
    centroid and distance "%f, %f, %f," -- expected text assembled here from those format strings."""
    n = 3
    P = orc.default_params(nCells=n, nDead=0, seed=4242, phase_std=0.0, max_time=1e9, light_x=-2.0, light_y=1.5)
    s = orc.Sim(P, reset=True)
    pos = np.array([[0.25, -0.5], [1.0, 2.0], [-3.125, 0.0625]], f)
    vel = np.array([[0.0, 0.001], [-0.25, 0.5], [0.0, 0.0]], f)
    rad = np.array([0.0775, 0.1, 0.1175], f)
    s.set("pos", pos), s.set("vel", vel), s.set("rad", rad)
    path = tmp_path / "dump.csv"
    assert s.dump(str(path), dump_interval=60.0, testing=1, mode="w")
    cx = f(f(f(f(0) + pos[0, 0]) + pos[1, 0]) + pos[2, 0]) / f(3)
    cy = f(f(f(f(0) + pos[0, 1]) + pos[1, 1]) + pos[2, 1]) / f(3)
    dx, dy = f(cx - f(-2.0)), f(cy - f(1.5))
    dist = f(np.sqrt(np.float64(f(f(dx * dx) + f(dy * dy)))))   # powf(powf(.,2)+powf(.,2), 0.5)
    want = "Seed, 4242\nTime,"
    want += "".join(f"Particlebot_{i}_xpos, Particlebot_{i}_ypos," for i in range(n))
    want += "".join(f"Particlebot_{i}_xvel, Particlebot_{i}_yvel," for i in range(n))
    want += "".join(f"Particlebot_{i}_rad," for i in range(n))
    want += "Centroid X, Centroid Y, Distance\n"
    want += "%f," % 0.0
    want += "".join("%f, %f," % (float(p[0]), float(p[1])) for p in pos)
    want += "".join("%f, %f," % (float(v[0]), float(v[1])) for v in vel)
    want += "".join("%f," % float(r) for r in rad)
    want += "%f, %f, %f,\n" % (float(cx), float(cy), float(dist))
    assert path.read_text() == want
    # a dump call away from a multiple of dump_interval writes nothing (particlebot.cpp:309-310)
    s.run(2)
    assert not s.dump(str(path), dump_interval=60.0, testing=1, mode="a")
    assert path.read_text() == want
