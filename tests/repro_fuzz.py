"""Repro helper for tests/soak_fuzz.py: re-run one trial with fine-grained checkpoints and several kernel forms.
  python tests/repro_fuzz.py <trial> <seed0> [every=25]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import soak_fuzz
    from oracle import orclib as orc
    import particlerobotsimulations_amd as pb
    from helpers import simparams_from_orc
    from particlerobotsimulations_amd import _capi
    trial, seed0 = int(sys.argv[1]), int(sys.argv[2])
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    pb.legacy.cudaInit(0, None)
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    # replay the trial's random draws exactly as soak_fuzz.one does, but keep the objects
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
    print({k: v for k, v in kw.items() if not isinstance(v, list)})
    P = orc.default_params(**kw)
    osim = orc.Sim(P, reset=True)
    if rng.random() < 0.4 and not payload:
        dead = (rng.random(n) < rng.uniform(0.05, 0.4)).astype(np.int32)
        osim.set("dead", dead)
    sp, keep = simparams_from_orc(P)
    form = rng.choice(["auto", "l1", "l1big", "l2", "l4", "l8", "l16", "l32", "l64", "resident", "variant0", "variant1"])
    si = float(rng.choice([0.23, 1.7, 180.0]))
    print("form", form, "sort_interval", si, "n", n)
    sims = {}
    for f in (str(form), "l1", "variant0"):
        g = pb.Sim(sp, keepalive=keep)
        if kind:
            _capi.check(_capi.lib().pbSimSetRng(g._h, kind))
        g.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                    dead=osim.get("dead"))
        if f.startswith("l") and f != "l1big":
            g.set_lanes_per_bot(int(f[1:]))
        elif f.startswith("variant"):
            g.set_force_variant(int(f[-1]))
        sims[f] = g
    step = 0
    start_fine = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
    prev = None
    while step < 1300:
        k = 1 if step >= start_fine else every
        osim.run(k, sort_interval=si)
        step += k
        ref = {kk: osim.get(kk) for kk in soak_fuzz.KEYS}
        nan_bots = int(np.isnan(ref["pos"]).any(axis=1).sum())
        line = [f"step {step}: max|v| {np.nanmax(np.abs(ref['vel'])):.3g} NaN bots {nan_bots}"]
        worst = None
        for f, g in sims.items():
            assert g.step(k, sort_interval=si) == k
            st = g.get_state()
            bad = {}
            for kk in soak_fuzz.KEYS:
                a, b = st[kk], ref[kk]
                if a is None:
                    bad[kk] = 0
                    continue
                both = np.isnan(a) & np.isnan(b)
                bad[kk] = int(((a.view(np.uint32) != b.view(np.uint32)) & ~both).sum())
            line.append(f"{f}: {bad if any(bad.values()) else 'ok'}")
            if any(bad.values()) and worst is None:
                worst = (f, st)
        print("  ".join(line), flush=True)
        if worst is not None:
            f, st = worst
            for kk in ("vel", "absForce_a", "pos"):
                a, b = st[kk], ref[kk]
                if a is None:
                    continue
                both = np.isnan(a) & np.isnan(b)
                d = (a.view(np.uint32) != b.view(np.uint32)) & ~both
                idx = np.flatnonzero(d.reshape(len(a), -1).any(axis=1))
                print(f"[{f}] {kk}: {len(idx)} bots differ (NaN-masked); first {idx[:6]}")
                for i in idx[:4]:
                    print(f"   bot {i}: oracle {b[i]} gpu {a[i]}  pos(oracle) {ref['pos'][i]} rad {ref['rad'][i]}"
                          + (f"  prev pos {prev['pos'][i]} prev vel {prev['vel'][i]}" if prev else ""))
            i = idx[0]
            if prev is not None:
                p0 = prev["pos"]
                dd = np.linalg.norm(p0 - p0[i], axis=1)
                near = np.argsort(np.where(np.isnan(dd), -1.0, dd))[:12]
                print("neighbourhood of bot", i, "at the previous step (NaN distances first):")
                for j in near:
                    print(f"   bot {j}: dist {dd[j]:.6g} pos {p0[j]} vel {prev['vel'][j]} rad {prev['rad'][j]}")
            break
        prev = ref


if __name__ == "__main__":
    main()
