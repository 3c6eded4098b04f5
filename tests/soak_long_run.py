"""Manual soak test (not collected by pytest): a whole example run -- hundreds of thousands of steps,
every re-sort and phase update of the real schedule -- on the GPU engine and on the CPU oracle,
compared bit for bit at checkpoints.

  python tests/soak_long_run.py examples/example.cfg 720000 [checkpoint_every=60000] [rng=0|1|2]
(rng: 0 the counter generator, 1 the cuRAND-compatible XORWOW, 2 the rocRAND-seeded XORWOW)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import orclib as orc
    import particlerobotsimulations_amd as pb
    from helpers import assert_bit_equal, simparams_from_orc
    cfg, steps = sys.argv[1], int(sys.argv[2])
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
    rng = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    P = orc.load_cfg(cfg, rngKind=rng)
    P.max_time = 1e9
    pb.legacy.cudaInit(0, None)
    if P.nDead > 0:
        return class_level(cfg, steps, every, rng)  # the dead-bot draw is the class's job: compare through it
    osim = orc.Sim(P, reset=True)
    sp, keep = simparams_from_orc(P)
    gsim = pb.Sim(sp, keepalive=keep)
    if rng:
        from particlerobotsimulations_amd import _capi
        _capi.check(_capi.lib().pbSimSetRng(gsim._h, rng))
    gsim.set_state(pos=osim.get("pos"), vel=osim.get("vel"), rad=osim.get("rad"), phase=osim.get("phase"),
                   dead=osim.get("dead"))
    done = 0
    tg = to = 0.0
    while done < steps:
        k = min(every, steps - done)
        t0 = time.perf_counter()
        gsim.step(k)
        gsim.synchronize()
        tg += time.perf_counter() - t0
        t0 = time.perf_counter()
        osim.run(k)
        to += time.perf_counter() - t0
        done += k
        st = gsim.get_state()
        for key in ("pos", "vel", "rad", "phase", "absForce_a", "absForce_r"):
            assert_bit_equal(st[key], osim.get(key), f"step {done}: {key}")
        com = st["pos"].astype(np.float64).mean(0)
        print(f"step {done}: bit-identical; COM ({com[0]:.6f}, {com[1]:.6f}); GPU {tg:.1f} s, oracle {to:.1f} s",
              flush=True)
    s = gsim.stats()
    print(f"OK {os.path.basename(cfg)}: {P.nCells} bots x {steps} steps bit-identical; {s['resorts']} re-sorts, "
          f"{s['phase_updates']} phase updates, rng kind {rng}; GPU {tg:.1f} s, oracle (1 thread) {to:.1f} s")


def class_level(cfg, steps, every, rng=0):
    """class Particlebot (placement + dead-bot draw from its private generator + fused engine) against
    the oracle's whole-simulation object"""
    from oracle import orclib as orc
    from particlerobotsimulations_amd import host
    from helpers import assert_bit_equal
    h = host.HostSim(cfg, engine="fused", max_time="1e9", pb_rng=["pbrng", "curand", "rocrand"][rng])
    P = orc.load_cfg(cfg, rngKind=rng)   # (product first: creating it calls srand())
    P.max_time = 1e9
    osim = orc.Sim(P, reset=True)
    done = 0
    while done < steps:
        k = min(every, steps - done)
        h.advance(k)
        osim.run(k)
        done += k
        for key in ("pos", "vel", "rad", "dead"):
            assert_bit_equal(h.get(key), osim.get(key), f"step {done}: {key}")
        print(f"step {done}: bit-identical (class level), dead bots {int(osim.get('dead').sum())}", flush=True)
    print(f"OK {os.path.basename(cfg)}: {P.nCells} bots x {steps} steps bit-identical through class Particlebot "
          f"(dead-bot draw included)")


if __name__ == "__main__":
    main()
