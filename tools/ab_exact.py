"""A/B of the exact headline kernel between library builds (lib dirs as arguments), each in its own process:
us per step at 10^6 and 8 x 10^6 bots (4 timed regions after 300 steps) and a hash of the final positions."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, hashlib
sys.path.insert(0, %r)
from particlerobotsimulations_amd import _capi
_capi.LIB_DIR = os.path.abspath(sys.argv[1]); _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so"); _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
import bench
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
n = int(sys.argv[2])
s = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1); s.step(300)
us = []
for _ in range(4):
    d, ms = s.step_timed(1000 if n <= 1000000 else 120); us.append(ms * 1e3 / d)
print(" ".join("%%.2f" %% u for u in us), "best %%.2f" %% min(us), hashlib.sha1(s.get_state()["pos"].tobytes()).hexdigest()[:10])
""" % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        for n in (1000000, 8000000):
            out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "particlerobotsimulations_amd", lib), str(n)],
                                 capture_output=True, text=True, timeout=600)
            print(f"{lib:8s} n={n:8d}: {(out.stdout.strip().splitlines() or [out.stderr[-200:]])[-1]}", flush=True)
