#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the particle-robot update loop on MI355X.

Workload (BASELINE.json configs[2]): 10^6 oscillating bots on a hexagonal
lattice (spacing 2*min_radius) in the generalised arena (2048^2 grid, walls +-240), one light at
(-230, 0), phase_std 0, dt 0.01, sort_interval 180.  A "step" is one timestep of the whole arena:
radius actuation + integration + neighbour forces + friction for every bot (one fused kernel).
State is resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--bots B]

N > 1 (launched by torch.distributed.run, one rank per GPU): the path does not shard a single
arena (neighbour forces couple every cell each step), so every rank runs its own independent arena
-- an ensemble member with its own seed offset -- with no collective in the
timed region; rank 0 gathers the per-arena centroid summaries over RCCL afterwards.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALG_BYTES_PER_PARTICLE_STEP = 64.0  # read 36 + write 28
# Headline workload: SQUARE lattice at pitch 2*min_radius (every bot touches 4 neighbours).
# SURVEY.md 8(d) proposed a HEXAGONAL lattice at that pitch.  Measured with the oracle: any hexagonal
# packing is numerically unstable under the reference's own parameters -- six contacts per bot put
# the explicit tangential damping at 6*shear*dt = 2.4 > 2 -- so it "boils" (speeds of several units/s,
# contact forces ~1000 N), the touching one first implodes and then expands into a dilute gas with
# no neighbours left (a step then costs 10x less), and at 10^6 bots it ends in NaN.  A square lattice
# (4 contacts, 1.6 < 2) is calm and jammed like the reference's random blobs, stays dense for the
# whole run (~57 candidate pairs per bot) and has a steady per-step cost.  The survey-literal hex
# lattice is still measured and reported under "survey_literal_lattice".
LATTICE_PITCH = 0.155
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def square_lattice(n, pitch):
    """side x side bots (row-major, bot i at column i % side, row i // side), centred on the origin."""
    import numpy as np
    f = np.float32
    side = int(np.ceil(np.sqrt(n)))
    i = np.arange(n, dtype=np.int64)
    half = f((side - 1) * 0.5)
    pos = np.empty((n, 2), dtype=f)
    pos[:, 0] = ((i % side).astype(f) - half) * f(pitch)
    pos[:, 1] = ((i // side).astype(f) - half) * f(pitch)
    return pos


def hex_lattice(n, spacing):
    """The initHexGrid recipe (particlebot.cpp:438-481) in float32, vectorised per ring."""
    import numpy as np
    f = np.float32
    h = f(np.sqrt(f(3.0)) * f(0.5))  # powf(3,0.5f)*0.5f
    ux = np.array([1.0, 0.5, -0.5, -1.0, -0.5, 0.5, 1.0], dtype=f)
    uy = np.array([0.0, h, h, 0.0, -h, -h, 0.0], dtype=f)
    pos = np.zeros((n, 2), dtype=f)
    i, ring, sp = 1, 1, f(spacing)
    while i < n:
        j = np.arange(ring, dtype=np.int32)
        for k in range(6):
            if i >= n:
                break
            a = (ux[k] * (ring - j).astype(f)).astype(f) * sp
            b = (ux[k + 1] * sp).astype(f) * j.astype(f)
            x = (a + b).astype(f)
            a = (uy[k] * (ring - j).astype(f)).astype(f) * sp
            b = (uy[k + 1] * sp).astype(f) * j.astype(f)
            y = (a + b).astype(f)
            m = min(ring, n - i)
            pos[i:i + m, 0] = x[:m]
            pos[i:i + m, 1] = y[:m]
            i += m
        ring += 1
    return pos


def workload_params(n_bots, seed):
    """SimParams of the synthetic phototaxis arena (main.cpp defaults + overrides)."""
    import numpy as np
    from particlerobotsimulations_amd import make_params
    f = np.float32
    max_radius = f(0.1175)
    cell = float(max_radius * f(2))
    grid = 2048
    d = dict(
        gridSize=(grid, grid), numCells=grid * grid, worldOrigin=(-240.0, -240.0), cellSize=(cell, cell),
        nCells=n_bots, nDead=0, gravity=float(f(9.81 * float(f(0.566)))), spring=1000.0, damping=10.0,
        shear=40.0, attraction=float(f(3.0) * f(0.000015884)), boundaryDamping=-1.0, friction=float(f(0.4)),
        massFactor=1.0, frictionFactor=1.0, radFactor=2.0, attractionFactor=0.0, constraint=0.5,
        constraint_contraction=10.0, centroid_steps=24000, centroid_int=10.0, centroid_radius=0.05,
        light_x=-230.0, light_y=0.0, phase_update_interval=12.0, control=0, config=4,
        min_radius=float(f(0.0775)), max_radius=float(max_radius), rise_period=2.0, freq=float(f(0.5) / f(25)),
        nobstacles=0, n_cir_obstacles=0, Nx=5, phase_std=0.0, seed=seed, light_shadow=0, testing=0,
        constrained_contraction=0, display_shadow=0, time_to_dead=0.0, max_time=1e9)
    return make_params(d)


def cpu_baseline(n_bots, pitch=LATTICE_PITCH, budget_s=12.0):
    """The oracle (our CPU port: the reference has no CPU path) timed on this host's cores on the
    SAME workload, for a bounded number of steps."""
    import numpy as np
    from oracle import orclib
    P = orclib.default_params(nCells=n_bots, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0,
                              light_y=0.0, grid=2048, arena_half=240.0)
    cores = orclib.usable_cpus()
    orclib.lib().orc_set_num_threads(cores)
    sim = orclib.Sim(P, reset=True, hex=True)
    sim.set("pos", square_lattice(n_bots, pitch))
    sim.run(1)  # first step: includes the initial sort
    t0 = time.perf_counter()
    steps = 0
    while True:
        sim.run(1)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 5000:
            break
    cores_used = orclib.lib().orc_num_threads()
    # the same arena on ONE thread, for a per-core figure: a few steps are enough
    orclib.lib().orc_set_num_threads(1)
    t1 = time.perf_counter()
    steps1 = 0
    while True:
        sim.run(1)
        steps1 += 1
        el1 = time.perf_counter() - t1
        if el1 > min(3.0, budget_s / 4) or steps1 >= 200:
            break
    orclib.lib().orc_set_num_threads(cores_used)
    sim.close()
    return {"value": n_bots * steps / el, "unit": "particle-steps/s", "cores": cores_used,
            "value_1_thread": n_bots * steps1 / el1,
            "kind": "port",
            "sample": f"{steps} steps of the same {n_bots}-bot arena after 1 warm-up step, OpenMP over bots "
                      f"({el:.1f} s); reported, not optimised"}


def make_sim(pb, n, pitch, seed, lattice="square"):
    import numpy as np
    sp, keep = workload_params(n, seed=seed)
    sim = pb.Sim(sp, wall_half=240.0, keepalive=keep)
    pos = square_lattice(n, pitch) if lattice == "square" else hex_lattice(n, np.float32(pitch))
    sim.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                  phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    return sim


def survey_literal(pb, n, steps, warmup):
    """The hexagonal lattice exactly as SURVEY.md 8(d) words it, reported beside the headline."""
    sim = make_sim(pb, n, LATTICE_PITCH, seed=1, lattice="hex")
    sim.step(warmup)
    first = min(300, steps)
    d1, ms1 = sim.step_timed(first)
    d2, ms2 = (0, 0.0) if steps <= first else sim.step_timed(steps - first)
    cx, cy = sim.centroid()
    sim.close()
    return {"lattice": "hexagonal", "pitch": LATTICE_PITCH, "steps": steps, "warmup": warmup,
            "finite_at_end": bool(cx == cx and cy == cy),
            "value": n * (d1 + d2) / ((ms1 + ms2) * 1e-3), "unit": "particle-steps/s (device time)",
            "us_per_step_first_300": ms1 * 1e3 / max(d1, 1),
            "us_per_step_rest": (ms2 * 1e3 / d2) if d2 else None,
            "note": "numerically unstable packing: dense only while it implodes, then a dilute gas / NaN "
                    "(see the LATTICE_PITCH comment in bench.py)"}


def streamlined_leg(pb, n, pitch, steps, warmup):
    """The opt-in streamlined force arithmetic (force variant 3; NOT bit-identical, DESIGN.md
    "Streamlined") on the same workload: its deviation from the exact kernel over one 10-step window
    from the same state, then its throughput over `steps` steps (device time, HIP events)."""
    import numpy as np
    exact, fast = make_sim(pb, n, pitch, seed=1), make_sim(pb, n, pitch, seed=1)
    for s in (exact, fast):
        s.set_force_variant(2)
        s.step(warmup)
    fast.set_force_variant(3)
    exact.step(10)
    fast.step(10)
    a, b = exact.get_state()["pos"].astype(np.float64), fast.get_state()["pos"].astype(np.float64)
    exact.close()
    d = np.linalg.norm(b - a, axis=1)
    rel = d / np.maximum(np.linalg.norm(a, axis=1), 1.0)
    com = float(np.linalg.norm(a.mean(0) - b.mean(0)))
    done, ms = fast.step_timed(steps)
    cx, cy = fast.centroid()
    fast.close()
    achieved = ALG_BYTES_PER_PARTICLE_STEP * n * done / (ms * 1e-3) / 1e9
    return {"value": n * done / (ms * 1e-3), "unit": "particle-steps/s (device time)", "steps": done,
            "ms_per_step": ms / max(done, 1), "finite_at_end": bool(cx == cx and cy == cy),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "kernel": "k_force_stream<FUSE>"},
            "parity": {"against": f"the exact kernel (bit-identical to the oracle) from the same state after "
                                  f"{warmup} steps", "window_steps": 10, "max_abs_dpos": float(d.max()),
                       "median_abs_dpos": float(np.median(d)), "bots_beyond_1e-5_relative": int((rel > 1e-5).sum()),
                       "com_abs_dev": com},
            "note": "opt-in: pbSimSetForceVariant(sim, 3).  v_rsq/v_rcp/FMA arithmetic, |F_attr| taken from its "
                    "coefficient, contact terms added after attraction terms.  Not bit-identical; held to 1e-5 "
                    "relative over 10-step windows by tests/test_gpu_streamlined.py.  `value` above is the exact "
                    "kernel."}


def profiled_traffic():
    """HBM bytes per k_force launch from the committed rocprofv3 PMC passes of this same command
    (profiles/latest_traffic.json, written by tools/profile.sh); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as fh:
            return json.load(fh)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--bots", type=int, default=1_000_000)
    ap.add_argument("--pitch", type=float, default=LATTICE_PITCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-survey-literal", action="store_true")
    ap.add_argument("--no-streamlined", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the cpu_baseline sample")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: exercises the N>1 code path")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1 or args.force_dist:
        # torch first: its bundled HIP runtime must be the one instance in the process
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import particlerobotsimulations_amd as pb

    if dist is None:
        pb.legacy.cudaInit(0, None)  # otherwise torch.cuda.set_device above already chose this rank's GPU

    n = args.bots
    sim = make_sim(pb, n, args.pitch, seed=1 + rank)

    def barrier():
        sim.synchronize()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    sim.step(args.warmup)
    barrier()
    s0 = sim.stats()
    t0 = time.perf_counter()
    done, dev_ms = sim.step_timed(args.steps)
    barrier()
    wall = time.perf_counter() - t0
    s1 = sim.stats()
    assert done == args.steps, (done, args.steps)

    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # the only data exchange of an ensemble: per-arena summaries (time, COMx, COMy), gathered
        cx, cy = sim.centroid()
        mine = torch.tensor([sim.time, cx, cy], dtype=torch.float64, device="cuda")
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        summaries = [[float(x) for x in v.tolist()] for v in allv]
    else:
        cx, cy = sim.centroid()
        summaries = [[sim.time, cx, cy]]
    assert cx == cx and cy == cy, "simulation state went NaN: the benchmark workload is invalid"

    if rank == 0:
        launches = (s1["fused_launches"] - s0["fused_launches"]) + (s1["plain_launches"] - s0["plain_launches"])
        value = world * n * args.steps / wall
        # dominant kernel = k_force (one launch per step); duration from the HIP events recorded on
        # the simulation's own stream around the timed region
        avg_launch_s = (dev_ms * 1e-3) / max(launches, 1)
        achieved = ALG_BYTES_PER_PARTICLE_STEP * n / avg_launch_s / 1e9
        tr = profiled_traffic() if n == 1_000_000 else None
        out = {
            "metric": "particle-steps/sec at 10^6 bots; achieved HBM GB/s vs peak; 1/2/4/8-GPU ensemble",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic phototaxis arena (BASELINE configs[2]): square lattice of oscillating "
                                   f"bots at pitch {args.pitch} (jammed and dense for the whole run), one light "
                                   "at (-230,0), 2048^2 grid, walls +-240, phase_std 0",
                       "bots_per_gpu": n, "dt": 0.01, "sort_interval": 180.0,
                       "parallelism": "single arena" if world == 1 else f"{world} independent arenas, one per GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                         "traffic_source": (f"profiles/latest_traffic.json ({tr['profile']}): {tr['method']}"
                                            if tr else None),
                         "valu": ({k: tr[k] for k in ("valu_roofline_frac", "valu_time_us", "launch_us_unprofiled",
                                                      "ns_simple", "ns_trans", "trans_per_wave",
                                                      "valu_issue_share", "valu_insts_per_wave",
                                                      "valu_lane_utilisation") if k in tr} if tr else None),
                         "kernel": "k_force<FUSE> (forces of step n + radius/integration of step n+1)",
                         "launches": launches, "avg_launch_us": avg_launch_s * 1e6,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_PARTICLE_STEP * n,
                         "note": "the kernel is VALU-bound, not HBM-bound: ~50 neighbour pairs per bot, each "
                                 "with 4 IEEE divisions and 2 IEEE square roots (DESIGN.md section 5).  `valu` "
                                 "prices its instruction stream (PMC counts) at the issue rates tools/valu_rate "
                                 "measures at 8 waves/SIMD: valu_roofline_frac is that VALU time over the launch time"},
            "device_ms_timed_region": dev_ms,
            "summaries_time_comx_comy": summaries,
        }
        sim.close()
        if world == 1 and not args.no_survey_literal:
            out["survey_literal_lattice"] = survey_literal(pb, n, args.steps, args.warmup)
        if world == 1 and not args.no_streamlined:
            out["streamlined"] = streamlined_leg(pb, n, args.pitch, args.steps, args.warmup)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.pitch, args.cpu_seconds)
        # the ONE JSON line goes last: push out whatever C libraries (RCCL's version banner) still hold
        # in stdio buffers first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
