#!/usr/bin/env python3
"""us/step of one simulation on the bench lattice for every lanes-per-bot form of the per-step force kernel,
over batch sizes: the table the automatic dispatch thresholds in pb_engine.hip (forcePlan) are fitted to.
  python tools/lanes_sweep.py [--steps 400] [--sizes 300,1000,...] [--forms 1,2,4,8,16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--sizes", default="300,1000,4000,8192,12000,20000,30000,49152,60000,80000,100000,131072,"
                                       "150000,200000,300000")
    ap.add_argument("--forms", default="1,2,4,8,16")
    ap.add_argument("--libdir", default=None, help="load the libraries from this directory (experimental builds)")
    args = ap.parse_args()
    if args.libdir:
        from particlerobotsimulations_amd import _capi
        _capi.LIB_DIR = os.path.abspath(args.libdir)
        _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so")
        _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    forms = [int(x) for x in args.forms.split(",")]
    print("bots      " + "".join(f"   L={L:<5d}" for L in forms) + "   auto (L)")
    for n in [int(x) for x in args.sizes.split(",")]:
        row = []
        for L in forms + [0]:
            sim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
            sim.set_resident(1)
            sim.set_lanes_per_bot(L)
            sim.step(200)
            best = min(sim.step_timed(args.steps)[1] * 1e3 / args.steps for _ in range(3))
            auto = sim.config()["lanes_per_bot"]
            sim.close()
            row.append(best)
        best_L = forms[min(range(len(forms)), key=lambda i: row[i])]
        print(f"{n:9d} " + "".join(f"{t:9.2f}{'*' if forms[i] == best_L else ' '}" for i, t in enumerate(row[:-1])) +
              f"{row[-1]:9.2f} ({auto})", flush=True)


if __name__ == "__main__":
    main()
