#!/usr/bin/env python3
"""A/B of the pipelined multi-step launch (pbSimSetPipelined 2) against one launch per step on the bench arena:
bit-for-bit comparison of every state array after the same steps, then timing of both.
    python tools/pipelined_ab.py [n_bots] [steps] [--time-only]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import particlerobotsimulations_amd as pb

args = [a for a in sys.argv[1:] if not a.startswith("--")]
time_only = "--time-only" in sys.argv
n = int(args[0]) if len(args) > 0 else 1_000_000
steps = int(args[1]) if len(args) > 1 else 400
DT = float(os.environ.get("PB_AB_DT", "0.01"))   # 0: frozen positions (timing experiments with the waits knocked out)
a = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
b = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
a.set_pipelined(1)
b.set_pipelined(2)


def differ():
    sa, sb = a.get_state(), b.get_state()
    return [f for f in sa if sa[f] is not None and
            not np.array_equal(np.asarray(sa[f]).view(np.uint32), np.asarray(sb[f]).view(np.uint32))]


if not time_only:
    for k in (1, 5, 37, 300) + tuple([500] * int(os.environ.get('PB_AB_LONG', '0'))):
        a.step(k)
        b.step(k)
        b.synchronize()
        bad = differ()
        print(f"after +{k} steps: {'IDENTICAL' if not bad else 'DIFFERENT in ' + ','.join(bad)}; "
              f"pipelined launches so far {b.stats()['pipelined_launches']}", flush=True)
        if bad:
            sys.exit(1)
for rep in range(2):
    for name, s in (("per-step", a), ("pipelined", b)):
        s.step(200, dt=DT)
        done, ms = s.step_timed(steps, dt=DT)
        print(f"{name:10s} {ms * 1e3 / done:8.2f} us/step over {done} steps", flush=True)
if not time_only:
    bad = differ()
    print("final:", "IDENTICAL" if not bad else "DIFFERENT " + ",".join(bad))
else:
    sa, sb = a.get_state(), b.get_state()
    print("pipelined state: non-finite positions", int((~np.isfinite(sb["pos"])).any(axis=1).sum()), "of", n,
          "| bots whose position differs from the per-step run's", int((sa["pos"] != sb["pos"]).any(axis=1).sum()),
          "| max |dpos|", float(np.nanmax(np.abs(sa["pos"] - sb["pos"]))))
