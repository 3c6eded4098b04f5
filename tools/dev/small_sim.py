"""dev: one small simulation from an example .cfg, N steps, per-step vs resident timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particlerobotsimulations_amd import host
cfg = sys.argv[1] if len(sys.argv) > 1 else "example.cfg"
for res in ("1", "2"):
    os.environ["PB_RESIDENT"] = res
    s = host.HostSim(os.path.join(ROOT, "examples", cfg), max_time="1e9")
    s.advance(500)
    nb = len(s.get('rad'))
    t0 = time.perf_counter(); s.advance(20000); dt = time.perf_counter() - t0
    print(f"{cfg} n={nb} resident={res}: {dt/20000*1e6:.2f} us/step wall")
