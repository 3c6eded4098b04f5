#!/usr/bin/env python3
"""us/step of the throughput form (one bot per lane, exact kernel) against the number of bots on the
bench lattice: how a wave's lifetime and the launch's ramp/tail depend on waves per SIMD.
  python tools/occupancy_sweep.py [--steps 300]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--sizes", default="16384,65536,131072,262144,393216,524288,655360,786432,1048576,1310720,"
                                       "1572864,2097152,3145728,4194304")
    ap.add_argument("--lds", default=None,
                    help="comma list of dynamic-LDS byte counts: caps workgroups per CU (160 KiB / bytes) of the "
                         "unchanged kernel, to price an LDS-hungry redesign before writing it")
    args = ap.parse_args()
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    if args.lds:
        os.environ["PB_ALLOW_ENV_OVERRIDES"] = "1"
        n = 1_000_000
        for lds in [int(x) for x in args.lds.split(",")]:
            os.environ["PB_DEBUG_LDS_BYTES"] = str(lds)
            sim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
            sim.step(100)
            best = min(sim.step_timed(args.steps)[1] * 1e3 / args.steps for _ in range(3))
            sim.close()
            per_cu = min(8, (160 * 1024) // lds) if lds else 8
            print(f"dynamic LDS {lds:6d} B -> {per_cu} workgroups/CU = {per_cu} waves/SIMD: {best:8.2f} us/step at 10^6 bots",
                  flush=True)
        return
    for n in [int(x) for x in args.sizes.split(",")]:
        sim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
        sim.set_lanes_per_bot(1)
        sim.step(100)
        best = 1e9
        for _ in range(3):
            done, ms = sim.step_timed(args.steps)
            best = min(best, ms * 1e3 / done)
        sim.close()
        waves = n / 64 / 1024
        print(f"{n:9d} bots  {waves:6.2f} waves/SIMD  {best:8.2f} us/step  {best / waves:7.2f} us per wave-per-SIMD  "
              f"{n / best * 1e6:.3e} particle-steps/s", flush=True)


if __name__ == "__main__":
    main()
