"""Who flips?  For one bracket case and epoch: 10 steps of oracle and force variant 3 from the synchronised state;
for every bot that ends more than 1e-5 relative away: the first step at which its velocity differs by > 1e-3 and both
sides' view of it there."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orclib as orc
import fma_bracket as fb
import particlerobotsimulations_amd as pb
case = sys.argv[1] if len(sys.argv) > 1 else "cfg5_member_1e5_dead20"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 400
pb.legacy.cudaInit(0, None)
orc.lib().orc_set_num_threads(orc.usable_cpus())
P, teacher, cands, _e, _w = fb.make_teacher(orc, case, lambda P: [fb.HipCandidate(pb, P)])
c = cands[0]
teacher.run(E); c.walk(E); c.sync(teacher)
hist = []
for k in range(10):
    v0 = teacher.get("vel")
    teacher.run(1); c.step(1)
    st = c.g.get_state()
    hist.append((v0, teacher.get("vel"), st["vel"].copy(), teacher.get("absForce_r"), st["absForce_r"].copy(), teacher.get("pos"), st["pos"].copy()))
dev = fb.rel_dev(hist[-1][6], hist[-1][5])
bad = np.flatnonzero(dev > 1e-5)
print(case, "epoch", E, "flipped bots:", bad.size, "dead among them:", int(teacher.view("dead")[bad].sum()))
hold = 2.0 * P.friction * P.gravity
for i in bad[:40]:
    for k, (v0, vo, vg, fo, fg, po, pg) in enumerate(hist):
        if np.linalg.norm(vg[i].astype(np.float64) - vo[i]) > 1e-3:
            print(f" bot {i} dead {teacher.view('dead')[i]} |p| {np.linalg.norm(po[i]):.1f} first at step {k}: |v0| {np.linalg.norm(v0[i]):.3g} "
                  f"oracle v {vo[i]} gpu v {vg[i]} fr o/g {fo[i]:.6g}/{fg[i]:.6g} final dev {dev[i]:.2g}")
            break
    else:
        print(f" bot {i}: no step with |dv| > 1e-3; final dev {dev[i]:.2g}, |p| {np.linalg.norm(hist[-1][5][i]):.2f}, max |dv| "
              f"{max(np.linalg.norm(h[2][i].astype(np.float64) - h[1][i]) for h in hist):.2g}")
