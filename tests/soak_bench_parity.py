"""Manual soak (not collected by pytest): the bench.py headline workload itself -- 10^6 bots, square
lattice, 100 + 2400 steps -- on the GPU engine and on the CPU oracle (OpenMP), bit for bit.

  python tests/soak_bench_parity.py [bots=1000000] [steps=2500] [force_sums=1]

force_sums 1 (default since round 6): the HEADLINE form, both magnitude sums (everything collideD writes; absForce_a is
compared too); 0: the library's default form (absForce_a not maintained).
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
    from helpers import assert_bit_equal
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    pb.legacy.cudaInit(0, None)
    sums = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    gsim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1, force_sums=sums)
    print(f"kernel: {gsim.force_kernel_name().split('(')[0]}", flush=True)
    P = orclib.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0, light_y=0.0,
                              grid=2048, arena_half=240.0)
    orclib.lib().orc_set_num_threads(orclib.usable_cpus())
    osim = orclib.Sim(P, reset=True, hex=True)
    osim.set("pos", bench.square_lattice(n, bench.LATTICE_PITCH))
    done, t0 = 0, time.perf_counter()
    while done < steps:
        k = min(500, steps - done)
        gsim.step(k)
        osim.run(k)
        done += k
        st = gsim.get_state()
        for key in ("pos", "vel", "rad", "absForce_a", "absForce_r"):
            assert_bit_equal(st[key], osim.get(key), f"step {done}: {key}")
        speed = float(np.abs(st["vel"]).max())
        print(f"step {done}: bit-identical; max |v| {speed:.3g}; {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"OK bench workload: {n} bots x {steps} steps, GPU engine == oracle on every state array")


if __name__ == "__main__":
    main()
