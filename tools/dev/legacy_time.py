"""dev: per-step time of the class through the legacy (call-for-call) seam vs the fused engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particlerobotsimulations_amd import host
n = sys.argv[1] if len(sys.argv) > 1 else "100000"
for eng in ("fused", "legacy"):
    s = host.HostSim(os.path.join(ROOT, "examples", "million_bots.cfg"), engine=eng, nCells=n, max_time="1e9")
    s.advance(50)
    t0 = time.perf_counter(); s.advance(300); s.get("pos"); dt = time.perf_counter() - t0
    print(f"n={n} engine={eng}: {dt/300*1e6:.1f} us/step")
    s.close()
