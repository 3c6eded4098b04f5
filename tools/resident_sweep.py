#!/usr/bin/env python3
"""us per batched timestep of an ensemble of `members` copies of an example configuration, per-step form
against resident form: the table DESIGN.md section 6 quotes and the cost model in pb_engine.hip (residentWanted)
is fitted to.
  python tools/resident_sweep.py [--steps 2000]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


LISTS = {"x1obs": "nobstacles", "x2obs": "nobstacles", "y1obs": "nobstacles", "y2obs": "nobstacles",
         "x_cir_obs": "n_cir_obstacles", "y_cir_obs": "n_cir_obstacles", "r_cir_obs": "n_cir_obstacles"}
SKIP = {"gridSizeX", "gridSizeY", "worldOriginX", "worldOriginY", "cellSizeX", "cellSizeY", "timestep", "sort_interval",
        "dump_interval", "camera_x", "camera_y", "light_radius", "display_interval", "video_interval", "csv_filename",
        "video_filename", "wallHalf", "rngKind"}


def simparams_of(fc):
    """SimParams (+ keepalive) from a resolved configuration of the host library."""
    from particlerobotsimulations_amd import make_params
    d = {}
    for name, _ in fc._fields_:
        if name in SKIP:
            continue
        if name in LISTS:
            d[name] = [getattr(fc, name)[i] for i in range(getattr(fc, LISTS[name]))]
        else:
            d[name] = getattr(fc, name)
    d["gridSize"] = (fc.gridSizeX, fc.gridSizeY)
    d["worldOrigin"] = (fc.worldOriginX, fc.worldOriginY)
    d["cellSize"] = (fc.cellSizeX, fc.cellSizeY)
    return make_params(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--members", default="1,8,32,64,128,256")
    ap.add_argument("--libdir", default=None, help="load the libraries from this directory (experimental builds)")
    args = ap.parse_args()
    if args.libdir:
        from particlerobotsimulations_amd import _capi
        _capi.LIB_DIR = os.path.abspath(args.libdir)
        _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so")
        _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
    import particlerobotsimulations_amd as pb
    from particlerobotsimulations_amd import host
    pb.legacy.cudaInit(0, None)
    cfgs = ["example_dead_cells.cfg", "example_object_transport.cfg", "example.cfg", "example_obstacle.cfg",
            "example_gap.cfg"]
    members = [int(x) for x in args.members.split(",")]
    print("cfg (bots)                      members: " + "  ".join(f"{m:>17d}" for m in members))
    for cfg in cfgs:
        path = os.path.join(ROOT, "examples", cfg)
        over = {"max_time": "1e9"}
        if host.load_config(path).nDead > 0:
            over["nDead"] = "0"
        hs = host.HostSim(path, engine="host", **over)     # placement by class Particlebot, no GPU work
        st = {k: hs.get(k).copy() for k in ("pos", "vel", "rad", "phase", "dead")}
        hs.close()
        nbots = int(host.load_config(path, **over).nCells)
        row = []
        for m in members:
            cell = []
            for mode in (1, 2):   # 1 = per-step launches, 2 = resident
                plist, keeps = [], []
                for k in range(m):
                    sp, keep = simparams_of(host.load_config(path, **over))
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
        print(f"{cfg:28s} ({nbots:4d})   " + "  ".join(f"{c:>17s}" for c in row), flush=True)
    print("per-step/resident us per batched step; R/S = what the automatic choice picks; * = it is the faster one")


if __name__ == "__main__":
    main()
