"""Manual soak (not collected by pytest): one very large arena on the GPU engine and on the oracle.

  python tests/soak_large_arena.py [bots=32000000] [steps=6]
"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    side = int(np.ceil(np.sqrt(n)))
    half = float(np.ceil(side * bench.LATTICE_PITCH / 2 + 8))
    grid = 1 << int(np.ceil(np.log2(2 * half / 0.235)))
    P = orclib.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-half + 10, light_y=0.0,
                              grid=grid, arena_half=half)
    print(f"{n} bots, lattice {side}^2, walls +-{half}, grid {grid}^2", flush=True)
    pb.legacy.cudaInit(0, None)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, wall_half=half, keepalive=keep)
    pos = bench.square_lattice(n, bench.LATTICE_PITCH)
    rng = np.random.default_rng(1)
    vel = (rng.standard_normal((n, 2)) * 0.01).astype(np.float32)   # not at rest: exercises every force path
    rad = rng.uniform(0.0775, 0.1175, n).astype(np.float32)
    zeros = np.zeros(n, np.float32)
    dead = np.zeros(n, np.int32)
    gsim.set_state(pos=pos, vel=vel, rad=rad, phase=zeros, dead=dead)
    orclib.lib().orc_set_num_threads(orclib.usable_cpus())
    osim = orclib.Sim(P, reset=False)
    for name, a in (("pos", pos), ("vel", vel), ("rad", rad), ("phase", zeros), ("dead", dead)):
        osim.set(name, a)
    t0 = time.perf_counter()
    done, ms = gsim.step_timed(steps)
    print(f"GPU: {steps} steps in {ms:.1f} ms device time = {n * steps / ms / 1e6:.2f}e9 particle-steps/s "
          f"(includes the initial sort)", flush=True)
    osim.run(steps)
    st = gsim.get_state()
    for key in ("pos", "vel", "rad", "absForce_a", "absForce_r"):
        assert_bit_equal(st[key], osim.get(key), key)
    done, ms = gsim.step_timed(20)
    print(f"OK {n} bots x {steps} steps bit-identical to the oracle ({time.perf_counter() - t0:.0f} s); steady state "
          f"{ms / 20 * 1e3:.0f} us/step = {n * 20 / ms / 1e6:.2f}e9 particle-steps/s")


if __name__ == "__main__":
    main()
