import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle import orclib as orc
from helpers import jittered_blob, simparams_from_orc
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
case = sys.argv[1] if len(sys.argv) > 1 else "x_wrap"
rng = np.random.default_rng(11)
if case == "x_wrap":
    n = 6000
    P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=80.0, light_y=80.0)
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(57.0, 61.0)); vel = vel + np.float32(0.3)
else:
    n = 12000
    P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=-30.0, light_y=5.0)
    pos, vel, rad = jittered_blob(n, 0.17, rng, center=(3.0, -2.0)); vel = vel + np.float32([1.5, 0.7])
sp, keep = simparams_from_orc(P)
sims = []
for form in (0, 1):
    s = pb.Sim(sp, keepalive=keep)
    s.set_state(pos=pos, vel=vel, rad=rad, phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    s.set_lanes_per_bot(1); s.set_resident(1); s.set_force_variant(3); s.set_stream_form(form)
    sims.append(s)
total = 0
for k in range(70):
    for s in sims: s.step(1)
    a, b = sims[0].get_state(), sims[1].get_state()
    bad = np.flatnonzero((a["vel"].view(np.uint32) != b["vel"].view(np.uint32)).any(1))
    if bad.size:
        cs = 0.235
        gx = np.floor((a["pos"][:, 0] + 64) / cs).astype(int); gy = np.floor((a["pos"][:, 1] + 64) / cs).astype(int)
        print("step", k + 1, "differing bots", bad.size, "stats", sims[1].stream_stats())
        for i in bad[:10]:
            print("  bot", i, "gx,gy", gx[i], gy[i], "vel0", a["vel"][i], "vel1", b["vel"][i], "fr", a["absForce_r"][i], b["absForce_r"][i])
        print("  gx range of bad", gx[bad].min(), gx[bad].max(), "gy", gy[bad].min(), gy[bad].max(), "| all gx", gx.min(), gx.max(), "gy", gy.min(), gy.max())
        break
else:
    print("no difference in 70 steps", sims[1].stream_stats())
