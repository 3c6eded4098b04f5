#!/usr/bin/env python3
"""us per batched timestep of an ensemble of `members` copies of an example configuration, per-step form
against resident form: the table DESIGN.md 6b quotes and the cost model in pb_engine.hip (residentWanted)
is fitted to.
  python tools/resident_sweep.py [--steps 2000]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--members", default="1,8,32,64,128,256")
    args = ap.parse_args()
    from oracle import orclib as orc   # placement only (a tool, not the product path)
    import particlerobotsimulations_amd as pb
    from helpers import simparams_from_orc
    pb.legacy.cudaInit(0, None)
    cfgs = ["example_dead_cells.cfg", "example_object_transport.cfg", "example.cfg", "example_obstacle.cfg",
            "example_gap.cfg"]
    members = [int(x) for x in args.members.split(",")]
    print("cfg (bots)                      members: " + "  ".join(f"{m:>17d}" for m in members))
    for cfg in cfgs:
        P = orc.load_cfg(os.path.join(ROOT, "examples", cfg))
        P.max_time = 1e9
        P.nDead = 0 if P.nDead > 0 else P.nDead
        o = orc.Sim(P, reset=True)
        st = {k: o.get(k) for k in ("pos", "vel", "rad", "phase", "dead")}
        row = []
        for m in members:
            cell = []
            for mode in (1, 2):   # 1 = per-step launches, 2 = resident
                plist, keeps = [], []
                for k in range(m):
                    sp, keep = simparams_from_orc(P)
                    plist.append(sp)
                    keeps.append(keep)
                ens = pb.Ensemble(plist, keepalive=keeps)
                for k in range(m):
                    ens.set_state_of(k, **st)
                ens.set_resident(mode)
                ens.step(300)
                best = min(ens.step_timed(args.steps)[1] * 1e3 / args.steps for _ in range(2))
                auto_sim = ens
                auto_sim.set_resident(0)
                chosen = auto_sim.config()["resident"]
                ens.close()
                cell.append(best)
            row.append(f"{cell[0]:6.2f}/{cell[1]:6.2f} {'R' if chosen else 'S'}{'*' if (cell[1] < cell[0]) == bool(chosen) else '!'}")
        print(f"{cfg:28s} ({P.nCells:4d})   " + "  ".join(f"{c:>17s}" for c in row), flush=True)
    print("per-step/resident us per batched step; R/S = what the automatic choice picks; * = it is the faster one")


if __name__ == "__main__":
    main()
