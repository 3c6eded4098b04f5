"""dev: per-step device time of a small-blob ensemble for lanes-per-bot 1 vs 8 (HIP events)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle import orclib as orc
from helpers import simparams_from_orc
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for members in (1, 8, 32, 128, 512):
    ps, keep, states = [], [], []
    base = orc.Sim(orc.default_params(nCells=n, nDead=0, seed=7777, light_x=-5.0, light_y=0.0, max_time=1e9))
    for k in range(members):
        P = orc.default_params(nCells=n, nDead=0, seed=1000 + k, light_x=-5.0, light_y=0.0, max_time=1e9)
        sp, ka = simparams_from_orc(P); ps.append(sp); keep.append(ka)
    out = []
    for lanes in (1, 4, 8):
        ens = pb.Ensemble(ps, keepalive=keep)
        for k in range(members):
            ens.set_state_of(k, pos=base.get("pos"), vel=base.get("vel"), rad=base.get("rad"), phase=base.get("phase"), dead=base.get("dead"))
        ens.set_lanes_per_bot(lanes)
        ens.step(200)
        t0 = time.perf_counter(); done, ms = ens.step_timed(3000); wall = time.perf_counter() - t0
        out.append(f"lanes {lanes}: {ms/3000*1e3:6.1f} us/step device, {wall/3000*1e6:6.1f} us/step wall")
        ens.close()
    print(f"{members:4d} x {n} bots: " + " | ".join(out), flush=True)
