"""Manual soak (not collected by pytest): a long randomized differential run of the fused engine against
the oracle -- random physics constants, sizes (60 ... 40 000 bots), obstacles, payload mode, all three
light_shadow modes, the three phase-noise generators, every lanes-per-bot form and the resident kernel,
random re-sort intervals, dead sets, and mid-run ranged state edits -- bit for bit after every stretch.
  python tests/soak_fuzz.py [trials=200] [seed0=0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
KEYS = ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r")


def one(pb, orc, trial, seed0):
    from helpers import assert_bit_equal, simparams_from_orc
    from particlerobotsimulations_amd import _capi
    rng = np.random.default_rng(777_000 + seed0 * 100_003 + trial)
    n = int(rng.choice([60, 130, 300, 700, 1024, 1025, 1500, 5000, 12000, 40000]))
    payload = rng.random() < 0.25
    rmin = float(rng.uniform(0.05, 0.09))
    kind = int(rng.integers(0, 3))
    kw = dict(
        nCells=n, nDead=-1 if payload else 0, seed=int(rng.integers(1, 10 ** 6)), max_time=1e9,
        light_x=float(rng.uniform(-8, 8)), light_y=float(rng.uniform(-8, 8)),
        spring=float(rng.uniform(200, 3000)), damping=float(rng.uniform(0, 30)), shear=float(rng.uniform(0, 60)),
        friction=float(rng.uniform(0.05, 0.9)), gravity=float(rng.uniform(1, 9.81)),
        attraction=float(rng.choice([0.0, 1e-6, 4.8e-5, 1e-3, 1e-14])), boundaryDamping=float(rng.choice([-1.0, -0.5])),
        min_radius=rmin, max_radius=rmin * float(rng.uniform(1.2, 1.8)), rise_period=float(rng.choice([1.0, 2.0, 3.0])),
        Nx=int(rng.integers(2, 8)), constraint=float(rng.uniform(0.1, 2.0)),
        constrained_contraction=int(rng.integers(0, 2)), constraint_contraction=float(rng.uniform(1, 20)),
        phase_std=float(rng.choice([0.0, 0.3, 1.0])), phase_update_interval=float(rng.choice([3.0, 12.0])),
        light_shadow=int(rng.integers(0, 3)), massFactor=float(rng.uniform(1, 3)),
        frictionFactor=float(rng.uniform(0.5, 2)), attractionFactor=float(rng.uniform(0.1, 1.0)),
        radFactor=float(rng.uniform(1.0, 2.5)), rngKind=kind)
    if rng.random() < 0.6:
        kw.update(n_cir_obstacles=2, x_cir_obs=[2.0, 6.5], y_cir_obs=[0.5, -1.0], r_cir_obs=[0.4, 0.3],
                  nobstacles=1, x1obs=[3.0], x2obs=[3.2], y1obs=[-2.0], y2obs=[-0.6])
    P = orc.default_params(**kw)
    osim = orc.Sim(P, reset=True)
    if rng.random() < 0.4 and not payload:   # a dead set from the start
        dead = (rng.random(n) < rng.uniform(0.05, 0.4)).astype(np.int32)
        osim.set("dead", dead)
    sp, keep = simparams_from_orc(P)
    form = rng.choice(["auto", "l1", "l1big", "l2", "l4", "l8", "l16", "l32", "l64", "resident", "variant0", "variant1"])
    os.environ["PB_ALLOW_ENV_OVERRIDES"] = "1"
    os.environ["PB_DEBUG_FORCE_BIG"] = "1" if form == "l1big" else "0"   # the 64-bit-offset sweep on a small batch
    gsim = pb.Sim(sp, keepalive=keep)
    if kind:
        _capi.check(_capi.lib().pbSimSetRng(gsim._h, kind))
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    if form == "l1big":
        gsim.set_lanes_per_bot(1)
        assert gsim.config()["offsets64"] == 1
    elif form.startswith("l"):
        gsim.set_lanes_per_bot(int(form[1:]))
    elif form == "resident":
        gsim.set_resident(2)
    elif form.startswith("variant"):
        gsim.set_force_variant(int(form[-1]))
    if rng.random() < 0.3:     # both magnitude sums kept although nothing may read absForce_a
        gsim.set_force_sums(1)
    si = float(rng.choice([0.23, 1.7, 180.0]))
    step = 0
    marks = sorted({1, int(rng.integers(2, 40)), int(rng.integers(290, 420)), int(rng.integers(600, 1300))})
    for k in marks:
        osim.run(k - step, sort_interval=si)
        assert gsim.step(k - step, sort_interval=si) == k - step
        step = k
        st = gsim.get_state()
        for key in KEYS:
            a, b = st[key], osim.get(key)
            if a is None:  # absForce_a without a reader: not maintained
                continue
            nan = np.isnan(a) & np.isnan(b)
            assert_bit_equal(np.where(nan, 0, a).astype(a.dtype), np.where(nan, 0, b).astype(b.dtype),
                             f"trial {trial} (n={n}, form={form}, rng={kind}, payload={payload}) step {k}: {key}")
        if k == marks[1] and rng.random() < 0.5:   # a ranged edit in mid-run, on both sides
            lo = int(rng.integers(0, n - 1))
            cnt = int(rng.integers(1, min(50, n - lo) + 1))
            newv = (rng.standard_normal((cnt, 2)) * 0.05).astype(np.float32)
            v = osim.get("vel")
            v[lo:lo + cnt] = newv
            osim.set("vel", v)
            _capi.check(_capi.lib().pbSimSetStateRangeOf(gsim._h, 0, lo, cnt, None, _capi.np_ptr(newv), None, None, None))
    gsim.close()
    osim.close()
    return n, form


def main():
    from oracle import orclib as orc
    import particlerobotsimulations_amd as pb
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    pb.legacy.cudaInit(0, None)
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    t0 = time.perf_counter()
    seen = {}
    for t in range(trials):
        n, form = one(pb, orc, t, seed0)
        seen[form] = seen.get(form, 0) + 1
        if (t + 1) % 20 == 0:
            print(f"{t + 1} trials bit-identical ({time.perf_counter() - t0:.0f} s)", flush=True)
    print(f"OK fuzz: {trials} randomized simulations bit-identical to the oracle at every checkpoint; forms {seen}")


if __name__ == "__main__":
    main()
