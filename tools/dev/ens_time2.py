"""dev: ensemble timing of the per-step batch path with forced lanes-per-bot forms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particlerobotsimulations_amd import ensemble
cfg = os.path.join(ROOT, "examples", sys.argv[1])
members = int(sys.argv[2]); tmax = sys.argv[3]
os.environ["PB_RESIDENT"] = "1"
out = []
for lanes in ("0", "1", "4", "8"):
    os.environ["PB_LANES_PER_BOT"] = lanes
    over = [ensemble.member_overrides(k, 1000) for k in range(members)]
    t0 = time.perf_counter()
    rows, steps = ensemble.run_local(cfg, over, common={"max_time": tmax, "dump_interval": "10"})
    dt = time.perf_counter() - t0
    out.append(f"L{lanes}={dt/steps*1e6:.1f}")
print(sys.argv[1], "members", members, " ".join(out[1:]), "us/batched step")
