"""tests/fma_bracket.py's statistics for force variant 3 of an EXPERIMENTAL library build (lib dir as argv[1]):
flips per case next to the FMA build's, and the streamlined kernel's time at 10^6 bots."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from particlerobotsimulations_amd import _capi
if len(sys.argv) > 1:
    _capi.LIB_DIR = os.path.abspath(sys.argv[1]); _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so"); _capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
import particlerobotsimulations_amd as pb
from oracle import orclib as orc
import fma_bracket as fb, bench
pb.legacy.cudaInit(0, None)
orc.lib().orc_set_num_threads(orc.usable_cpus()); orc.variant_lib("fma").orc_set_num_threads(orc.usable_cpus())
tot = {}
for case in sys.argv[2:] or ["cfg2b_dead_cells_10k", "cfg5_member_1e5_dead20"]:
    r = fb.measure_case(orc, case, lambda P: [fb.HipCandidate(pb, P), fb.OracleCandidate(orc, P, "fma")])
    for name, row in fb.summarise(r).items():
        print(case, name, "flips", row["flips_total"], "of", row["bot_windows"], "p99 %.2g" % row["p99_max"], "max %.2g" % row["max_max"], flush=True)
s = bench.make_sim(pb, 1000000, bench.LATTICE_PITCH, seed=1); s.set_force_variant(3); s.step(300)
d, ms = s.step_timed(2000); print("streamlined us/step %.2f" % (ms * 1e3 / d))
