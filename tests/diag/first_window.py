"""The first 10-step window of force variant 3 from the exactly-touching placement (the degenerate state) and the six
later windows of tests/test_gpu_streamlined.py: fraction of bots within 1e-5, flips, max |dp| -- the numbers the
test's bounds are set from."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orclib as orc
from helpers import simparams_from_orc
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
for seed in (4321, 1, 2, 3):
    P = orc.default_params(nCells=3000, nDead=0, seed=seed, light_x=-3.0, light_y=2.0, phase_std=0.0, max_time=1e9)
    o = orc.Sim(P, reset=True)
    sp, keep = simparams_from_orc(P)
    g = pb.Sim(sp, keepalive=keep)
    g.set_state(pos=o.get("pos"), vel=o.get("vel"), rad=o.get("rad"), phase=o.get("phase"), dead=o.get("dead"))
    g.set_lanes_per_bot(1); g.set_resident(1); g.set_force_variant(3)
    o.run(10); g.step(10)
    p, r = g.get_state()["pos"].astype(np.float64), o.get("pos").astype(np.float64)
    dev = np.linalg.norm(p - r, axis=1) / np.linalg.norm(r, axis=1)
    print(f"seed {seed}: first window: within 1e-5: {(dev <= 1e-5).mean():.4f}; beyond: {(dev > 1e-5).sum()}; max |dp| {np.linalg.norm(p - r, axis=1).max():.3g}; "
          f"COM rel {np.linalg.norm(p.mean(0) - r.mean(0)) / np.linalg.norm(r.mean(0)):.2g}")
