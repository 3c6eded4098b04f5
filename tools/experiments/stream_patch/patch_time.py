import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
if len(sys.argv) > 1:
    from particlerobotsimulations_amd import _capi
    _capi.LIB_DIR = os.path.abspath(sys.argv[1]); _capi.HIP_SO = os.path.join(_capi.LIB_DIR, 'libparticlebot_hip.so'); _capi.HOST_SO = os.path.join(_capi.LIB_DIR, 'libparticlebot_host.so')
import particlerobotsimulations_amd as pb
pb.legacy.cudaInit(0, None)
n=1000000
for form in (0,1,1):
    s=bench.make_sim(pb,n,bench.LATTICE_PITCH,seed=1); s.set_force_variant(3); s.set_stream_form(form)
    s.step(300)
    d,ms=s.step_timed(2000)
    print("form",form,"us/step %.2f"%(ms*1e3/d), "tiles,fallbacks", s.stream_stats(), flush=True)
    s.close()
