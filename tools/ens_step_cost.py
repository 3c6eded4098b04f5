#!/usr/bin/env python3
"""us per batched timestep of the BASELINE configs[3] ensembles (32 members each), one batch alone and both batches
driven concurrently from two host threads, after `skip` steps of relaxation.  --libdir: experimental builds."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--libdir", default=None)
ap.add_argument("--members", type=int, default=32)
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--skip", type=int, default=600)
args = ap.parse_args()
if args.libdir:
    from particlerobotsimulations_amd import _capi
    _capi.LIB_DIR = os.path.abspath(args.libdir)
    _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so")
    _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
import particlerobotsimulations_amd as pb  # noqa: E402
from particlerobotsimulations_amd import ensemble  # noqa: E402
pb.legacy.cudaInit(0, None)
EX = lambda n: os.path.join(ROOT, "examples", n)
common = {"max_time": "1e9", "dump_interval": "6"}
mk = lambda cfg: ensemble.LocalEnsemble(EX(cfg), [f"seed\n{1000 + k}" for k in range(args.members)], common)


def timed(ens, steps):
    for e in ens:
        e.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=e.run_steps, args=(steps,)) for e in ens[1:]]
    for t in th:
        t.start()
    ens[0].run_steps(steps)
    for t in th:
        t.join()
    for e in ens:
        e.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


if os.environ.get("PB_ENS_SPLIT"):
    # the same 32 + 32 members as FOUR batches of 16 on four streams (four host threads)
    k = int(os.environ["PB_ENS_SPLIT"])
    mk2 = lambda cfg, lo, hi: ensemble.LocalEnsemble(EX(cfg), [f"seed\n{1000 + j}" for j in range(lo, hi)], common)
    per = args.members // k
    parts = [mk2(c, i * per, (i + 1) * per) for c in ("example_obstacle.cfg", "example_object_transport.cfg") for i in range(k)]
    for e in parts:
        e.run_steps(args.skip)
    for rep in range(2):
        print(f"{2 * k} batches of {per} members on {2 * k} streams: {timed(parts, args.steps):.2f} us per step of all {2 * args.members} members")
    sys.exit(0)
a, b = mk("example_obstacle.cfg"), mk("example_object_transport.cfg")
for e in (a, b):
    e.run_steps(args.skip)
for rep in range(2):
    print(f"obstacle alone {timed([a], args.steps):.2f} us/step | transport alone {timed([b], args.steps):.2f} | "
          f"both, two host threads {timed([a, b], args.steps):.2f}")
