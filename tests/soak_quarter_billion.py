"""Manual soak (not collected by pytest): one arena of 2.5 x 10^8 bots -- 26 GB of state, 16384^2 grid
(1 GiB cell table), walls +-1300 -- the largest batch the 32-bit byte-offset sweeps admit (2^28 bots):
two steps bit for bit against the oracle, then a timed stretch.
  python tests/soak_quarter_billion.py [bots=250000000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    from oracle import orclib
    import particlerobotsimulations_amd as pb
    from helpers import assert_bit_equal, simparams_from_orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
    pb.legacy.cudaInit(0, None)
    P = orclib.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-1290.0, light_y=0.0,
                              grid=16384, arena_half=1300.0)
    sp, keep = simparams_from_orc(P)
    t0 = time.perf_counter()
    sim = pb.Sim(sp, wall_half=1300.0, keepalive=keep)
    pos = bench.square_lattice(n, bench.LATTICE_PITCH)
    assert np.abs(pos).max() < 1299.0
    sim.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                  phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    print(f"created and loaded in {time.perf_counter() - t0:.1f} s", flush=True)
    orclib.lib().orc_set_num_threads(orclib.usable_cpus())
    osim = orclib.Sim(P, reset=True, hex=True)
    osim.set("pos", pos)
    del pos
    t0 = time.perf_counter()
    osim.run(2)
    print(f"oracle: 2 steps in {time.perf_counter() - t0:.1f} s on {orclib.lib().orc_num_threads()} threads", flush=True)
    sim.step(2)
    st = sim.get_state()
    for key in ("pos", "vel", "rad", "absForce_a", "absForce_r"):
        assert_bit_equal(st[key], osim.get(key), f"step 2: {key}")
    osim.close()
    del st
    done, ms = sim.step_timed(20)
    cx, cy = sim.centroid()
    print(f"OK {n} bots: 2 steps bit-identical to the oracle; then {done} steps at {ms / done:.2f} ms/step = "
          f"{ms * 1e3 / done / (n / 1e6):.1f} us per 10^6 bots, {n * done / (ms * 1e-3):.3e} particle-steps/s; "
          f"centroid ({cx:.2e}, {cy:.2e}); {sim.stats()}")
    sim.close()


if __name__ == "__main__":
    main()
