"""us per step per 10^6 bots of the exact kernels (force variant 2) and the tolerance kernel (3) on the bench lattice
at arena sizes inside and beyond the Infinity Cache (256 MB): 10^6 bots = 64 MB of state per step."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import particlerobotsimulations_amd as pb  # noqa: E402
pb.legacy.cudaInit(0, None)
for n in (1_000_000, 2_000_000, 4_000_000, 8_000_000):  # (each size costs ~n x 25 us of host-side lattice building: 13 GPU-minutes in all)
    out = []
    for v in (2, 3):
        s = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
        s.set_force_variant(v)
        s.step(60)
        steps = max(40, int(400e6 / n))
        d, ms = s.step_timed(steps)
        out.append(ms * 1e3 / d / (n / 1e6))
        s.close()
    print(f"{n:>9} bots: exact {out[0]:.2f} us per 1e6, tolerance {out[1]:.2f} us per 1e6 (ratio {out[1] / out[0]:.3f})", flush=True)
