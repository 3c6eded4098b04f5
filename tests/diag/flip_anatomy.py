"""Which decisions separate the streamlined kernel from the oracle?  (diagnostic; run on the GPU box)
For one bracket case and epoch: from the synchronised state, ONE step of oracle and of force variant 3; every bot
whose new velocity differs by more than 1e-5 is a decision flip; classify it by the oracle's view of the bot."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orclib as orc
import fma_bracket as fb
import particlerobotsimulations_amd as pb

case = sys.argv[1] if len(sys.argv) > 1 else "cfg5_member_1e5_dead20"
epochs = [int(x) for x in sys.argv[2:]] or [50, 400, 1195]
pb.legacy.cudaInit(0, None)
orc.lib().orc_set_num_threads(orc.usable_cpus())
P, teacher, cands, _e, _w = fb.make_teacher(orc, case, lambda P: [fb.HipCandidate(pb, P)])
c = cands[0]
step = 0
hold = 2.0 * P.friction * P.gravity
for E in epochs:
    teacher.run(E - step); c.walk(E - step); c.sync(teacher)
    for k in range(10):
        v0 = teacher.get("vel"); p0 = teacher.get("pos")
        teacher.run(1); c.step(1)
        st = c.g.get_state()
        v1, g1 = teacher.get("vel"), st["vel"]
        dv = np.linalg.norm(g1.astype(np.float64) - v1, axis=1)
        bad = np.flatnonzero(dv > 1e-5)
        fr_o, fr_g = teacher.get("absForce_r"), st["absForce_r"]
        print(f"epoch {E} step {k}: {bad.size} bots with |dv| > 1e-5; max |dv| {dv.max():.3g}; "
              f"max |d absForce_r| {np.abs(fr_g - fr_o).max():.3g}")
        for i in bad[:12]:
            was_rest = np.linalg.norm(v0[i]) < 1e-6
            print(f"   bot {i}: |v0| {np.linalg.norm(v0[i]):.3g} rest={was_rest} oracle v1 {v1[i]} gpu v1 {g1[i]} "
                  f"absForce_r oracle {fr_o[i]:.7g} gpu {fr_g[i]:.7g} |p| {np.linalg.norm(p0[i]):.3g} dead {teacher.view('dead')[i]}")
        # the first differing step is the interesting one; afterwards differences propagate
        if bad.size:
            break
    # resync
    for _ in range(0): pass
    c.resync(teacher) if False else None
    # bring both to the same step count: the GPU sim has run k+1 steps of variant 3; resync from the teacher
    c.g.set_state(pos=teacher.view("pos"), vel=teacher.view("vel"), rad=teacher.view("rad"), phase=teacher.view("phase"))
    c.g.set_forces(teacher.view("absForce_a"), teacher.view("absForce_r"))
    step = E + k + 1
