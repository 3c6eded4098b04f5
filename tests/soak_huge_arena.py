"""Manual soak (not collected by pytest): ONE arena of 2.5 x 10^8 ... 10^9 bots on one GPU (132 bytes of device
memory per bot: 26 ... 132 GB), walls and grid sized for the lattice.  Below 2^28 bots the throughput sweep
uses 32-bit byte offsets, from 2^28 on the 64-bit form (k_force<..., BIG>).  Two steps bit for bit against
the oracle, then a timed stretch.
  python tests/soak_huge_arena.py [bots=250000000]      (measured: 2.5e8, 3e8 and 1e9 bots, results/README.md)"""
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
    # arena and grid sized for the lattice: walls 10 % beyond it, grid = next power of two spanning the arena
    half = float(np.ceil(np.ceil(np.sqrt(n)) * bench.LATTICE_PITCH * 0.5 * 1.1 / 50.0) * 50.0)
    grid = 512
    while grid * 0.235 < 2 * half:
        grid *= 2
    print(f"{n} bots: walls +-{half}, grid {grid}^2", flush=True)
    P = orclib.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-(half - 10.0),
                              light_y=0.0, grid=grid, arena_half=half)
    sp, keep = simparams_from_orc(P)
    t0 = time.perf_counter()
    sim = pb.Sim(sp, wall_half=half, keepalive=keep)
    print("config:", sim.config(), flush=True)
    pos = bench.square_lattice(n, bench.LATTICE_PITCH)
    assert np.abs(pos).max() < half - 1.0
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
