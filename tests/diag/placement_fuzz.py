"""Random placements through the host library against the oracle's literal loop (CONFIG_RANDOM, particlebot.cpp:612-748):
sizes, seeds, radii (incl. discs wider than grid cells), payload runs.  usage: python tests/diag/placement_fuzz.py [cases=300]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orclib as orc  # noqa: E402
from particlerobotsimulations_amd import host  # noqa: E402

host.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(20261003)
bad = 0
for c in range(cases):
    rmin = float(np.round(10 ** rng.uniform(-1.7, 0.0), 4))
    rmax = float(np.round(rmin * rng.choice([0.8, 1.0, 1.2, 1.5, 2.0, 3.0]), 4))
    n = int(rng.choice([2, 3, 4, 7, 30, 200, 900, 2500, 6000]))
    # keep the blob inside the reference's 512-cell grid (beyond it the reference indexes out of bounds)
    cell = 2 * rmax
    if 2.2 * rmin * np.sqrt(n) + 6 > 0.45 * 512 * cell or 2.2 * rmin * np.sqrt(n) + 6 > 60:
        continue
    payload = rng.random() < 0.25 and n > 3
    kw = dict(nCells=n, seed=int(rng.integers(0, 2 ** 31 - 1)), min_radius=rmin, max_radius=rmax,
              nDead=-1 if payload else 0)
    if payload:
        kw["radFactor"] = float(rng.choice([1.0, 2.0, 5.0]))
    cfg = os.path.join(ROOT, "examples", "example_object_transport.cfg" if payload else "example.cfg")
    h = host.HostSim(cfg, engine="host", **{k: str(v) for k, v in kw.items()})
    o = orc.Sim(orc.load_cfg(cfg, **kw), reset=True)
    a, b = h.get("pos"), o.get("pos")
    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
        bad += 1
        first = int(np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(-1))[0] // 2)
        print("DIFFER", kw, "first at bot", first, flush=True)
    o.close()
print(f"{cases} cases drawn, {bad} differ")
