"""dev: ensemble timing, per-step batch vs resident form (PB_RESIDENT), product path only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particlerobotsimulations_amd import ensemble
cfg = os.path.join(ROOT, "examples", sys.argv[1])
members = int(sys.argv[2]); tmax = sys.argv[3]
for res in ("1", "0", "1", "2"):  # first pass = warm-up
    os.environ["PB_RESIDENT"] = res
    over = [ensemble.member_overrides(k, 1000) for k in range(members)]
    t0 = time.perf_counter()
    rows, steps = ensemble.run_local(cfg, over, common={"max_time": tmax, "dump_interval": "10"})
    dt = time.perf_counter() - t0
    print(f"{sys.argv[1]} members={members} steps={steps} resident={res}: {dt:.2f} s, {dt/steps*1e6:.2f} us/batched step, com0={rows[0,-1,1]:.6f}")
