"""force variant 3 at 10^6 bots for a library build (lib dir as argv[1]): us per step, twice"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particlerobotsimulations_amd import _capi
if len(sys.argv) > 1:
    _capi.LIB_DIR = os.path.abspath(sys.argv[1]); _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so"); _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
import hashlib, bench
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
for rep in range(2):
    s = bench.make_sim(pb, 1000000, bench.LATTICE_PITCH, seed=1); s.set_force_variant(3); s.step(300)
    d, ms = s.step_timed(2000)
    print("streamlined us/step %.2f" % (ms * 1e3 / d), hashlib.sha1(s.get_state()["pos"].tobytes()).hexdigest()[:10], flush=True)
    s.close()
