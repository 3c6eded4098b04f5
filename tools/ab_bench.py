#!/usr/bin/env python3
"""Interleaved A/B timing of force-kernel variants in ONE process on ONE device (the methodology of
cdna_hip_programming.md rule 24): every variant gets its own simulation started from the same
state; rounds of K steps alternate between them; device time comes from HIP events on each
simulation's stream.  Also checks that all variants end bit-identical.

  python tools/ab_bench.py --variants 0,1 --bots 1000000 --rounds 6 --steps 200 --skip 100
  python tools/ab_bench.py --variants 2s1,2 ...     (2s1: variant 2 with both magnitude sums kept)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import benchkit as bench  # noqa: E402  (the workload builders)
from bench_legs import BlobPlacement  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--bots", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--skip", type=int, default=100, help="untimed steps before the first round")
    ap.add_argument("--pitch", type=float, default=bench.LATTICE_PITCH)
    ap.add_argument("--lattice", default="square")
    ap.add_argument("--libdir", default=None, help="load the libraries from this directory (experimental builds)")
    args = ap.parse_args()
    if args.libdir:
        from particlerobotsimulations_amd import _capi
        _capi.LIB_DIR = os.path.abspath(args.libdir)
        _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so")
        _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    n = args.bots
    variants = args.variants.split(",")
    if args.lattice == "blob":   # the reference's kind of initial state: a random blob (pb_placement fastblob)
        pos, _ = BlobPlacement(n).get()
    else:
        pos = (bench.square_lattice(n, args.pitch) if args.lattice == "square"
               else bench.hex_lattice(n, np.float32(args.pitch)))
    sims = {}
    for v in variants:
        sp, keep = bench.workload_params(n, seed=1)
        s = pb.Sim(sp, wall_half=240.0, keepalive=keep)
        # "2" = force variant 2; "2s1" = variant 2 with both magnitude sums kept (pbSimSetForceSums 1); "3w0" / "3w1" =
        # the streamlined kernel with its neighbour walk pinned row by row / flattened (pbSimSetStreamWalk; "3": automatic)
        walk = None
        if "w" in v:
            v, walk = v.split("w")[0], int(v.split("w")[1])
        s.set_force_variant(int(v.split("s")[0]))
        if "s" in v:
            s.set_force_sums(int(v.split("s")[1]))
        if walk is not None:
            s.set_stream_walk(walk)
        s.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                    phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
        s.step(args.skip)
        sims[v if walk is None else f"{v}w{walk}"] = s
    times = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            done, ms = sims[v].step_timed(args.steps)
            times[v].append(ms / args.steps * 1e3)
    ref = sims[variants[0]].get_state()
    for v in variants:
        st = sims[v].get_state()
        same = all(np.array_equal(st[k].view(np.uint32), ref[k].view(np.uint32))
                   for k in ("pos", "vel", "rad", "absForce_a", "absForce_r") if st[k] is not None and ref[k] is not None)
        dev = np.abs(st["pos"].astype(np.float64) - ref["pos"]).max()
        t = np.array(times[v])
        print(f"variant {v}: us/step per round {np.round(t, 1).tolist()}  median {np.median(t):.1f}  min {t.min():.1f}  "
              f"=> {n / np.median(t) * 1e6:.3e} particle-steps/s   {sims[v].config()}   bit-identical to variant {variants[0]}: {same}  max|dpos| {dev:.3g}")


if __name__ == "__main__":
    main()
