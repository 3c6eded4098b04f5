"""Chunked steps re-priced: one timestep as C tile-range launches of the exact kernel on C
streams, chunk c of step n+1 waiting for chunks c-1, c, c+1 of step n (PB_DEBUG_CHUNKS=C[:streams], an experiment
switch behind PB_ALLOW_ENV_OVERRIDES=1; csrc/pb_force.hip launchForceT).  Each configuration in its own process
(the switch is read when the simulation is created), GPU_MAX_HW_QUEUES=8, 10^6 and 8 x 10^6 bots on the bench
lattice, 4 x 1000 timed steps after 300; the final positions are hashed: every configuration must give the same bits.
    python tools/experiments/chunked_steps_r4.py > profiles/r4_chunked_steps.txt"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, bench
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
n, steps = int(sys.argv[1]), int(sys.argv[2])
s = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
s.step(300)
us = []
for _ in range(4):
    d, ms = s.step_timed(steps)
    us.append(ms * 1e3 / d)
h = hashlib.sha1(s.get_state()["pos"].tobytes()).hexdigest()[:12]
print(" ".join(f"{u:.1f}" for u in us), "best %%.1f" %% min(us), h)
""" % ROOT


def run(n, steps, chunks, queues="8"):
    env = dict(os.environ, PB_ALLOW_ENV_OVERRIDES="1", GPU_MAX_HW_QUEUES=queues)
    env.pop("PB_DEBUG_CHUNKS", None)
    if chunks != "1":
        env["PB_DEBUG_CHUNKS"] = chunks
    out = subprocess.run([sys.executable, "-c", CHILD, str(n), str(steps)], env=env, capture_output=True, text=True,
                         timeout=900)
    return (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]


def main():
    print("chunks[:streams]  us/step (4 x timed regions)  best  sha1(pos)[:12]   -- GPU_MAX_HW_QUEUES=8 unless said")
    for n, steps in ((1_000_000, 1000), (8_000_000, 150)):
        print(f"== {n} bots")
        for chunks in ("1", "2", "4", "8", "8:4", "16:8", "1"):
            print(f"chunks {chunks:6s}: {run(n, steps, chunks)}", flush=True)
    print("== 1000000 bots, GPU_MAX_HW_QUEUES=4 (HIP's default)")
    for chunks in ("1", "8"):
        print(f"chunks {chunks:6s}: {run(1_000_000, 1000, chunks, queues='4')}", flush=True)


if __name__ == "__main__":
    main()
