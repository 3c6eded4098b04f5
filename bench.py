#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the particle-robot update loop on MI355X: the headline line.

  python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2]): 10^6 oscillating bots on a square lattice (pitch 0.155:
jammed and dense for the whole run, ~57 candidate pairs per bot; DESIGN.md section 5 says why not the hexagonal one) in
the generalised arena (2048^2 grid, walls +-240), one light at (-230, 0), phase_std 0, dt 0.01, sort_interval 180.
A "step" is one timestep of the whole arena -- radius actuation + integration + neighbour forces + friction for every
bot, one fused k_force launch.  State is resident in HBM before the timed region.

ONE kernel per headline: `value`, `ms_per_step` and `roofline` all describe the SAME K launches of the exact kernel in
the form that writes everything the reference's collideD writes (particlebot_kernel_impl.cuh:828-830: both magnitude
sums, `attraction_sums: 1`).  `value` is over the wall clock of the K steps (barrier + device synchronisation on both
sides, MAX over ranks); `roofline.avg_launch_us` is the HIP-event time of those same K launches on the simulation's
stream / K, priced at SURVEY 8(d)'s 64 algorithmic bytes per particle-step.  The library's DEFAULT form for the
reference's default parameters (constrained_contraction 0: Sum|F_attr| has no reader and is not computed, 56 B) is
timed right after it and reported as `default_form`.

N > 1 (torch.distributed.run, one rank per GPU; a bare `python bench.py --gpus N` starts the N ranks itself as a child
process): a single arena does not shard (neighbour forces couple every cell each step), so every rank steps its own
arena -- an ensemble member with its own seed -- with no collective in the timed region; the per-arena summaries are
gathered over RCCL afterwards (`collective`).  At every N the line also carries BASELINE configs[3] as written -- 256 +
256 Monte-Carlo seeds of the obstacle and object-transport examples at full length, member k on rank k mod N, summary
rows gathered over RCCL -- end to end (`ensemble`).

The line stays under 4 KB; everything else goes to bench_detail.json (and stderr).  Every other leg -- streamlined
kernel, 8 x 10^6 bots, random blob, host round trip, the configs[3]/[4] workloads on their own -- is
tools/bench_legs.py.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import benchkit as K  # noqa: E402

MAX_LINE_BYTES = 4096
# the workload itself, under the names the parity tests use (tests/test_gpu_baseline_configs.py holds exactly this
# arena, built by these functions, bit-equal to the oracle)
LATTICE_PITCH, square_lattice, workload_params, make_sim = K.LATTICE_PITCH, K.square_lattice, K.workload_params, K.make_sim


def timed_arena(pb, args, rank, world, dist, torch, warm):
    """The headline region on every rank: W warm-up steps, then EXACTLY K steps between barriers, timed twice over the
    same launches -- wall clock (the contract's `value`) and HIP events on the simulation's stream (`roofline`)."""
    n = args.bots
    sim = K.make_sim(pb, n, K.LATTICE_PITCH, seed=1 + rank, force_sums=1)
    cfg, kernel = sim.config(), sim.force_kernel_name()
    assert cfg["force_variant"] == 2 and cfg["attraction_sums"] == 1, cfg   # the exact kernel, both magnitude sums
    prewarm = warm.run()
    sim.step(args.warmup)
    # With a process group the OPENING barrier is entered while the device still works (warm-up steps + a few ms of
    # the scratch arena, all asynchronous) and the synchronisation comes after it: an RCCL barrier leaves the device
    # idle for some hundred microseconds otherwise and the first timed steps would run at idle clocks.
    if dist is not None:
        warm.keep_busy(3.0)
        dist.barrier()
        warm.synchronize()
    sim.synchronize()
    if dist is not None:
        K.dev_sync(torch)
    s0 = sim.stats()
    # the K steps under two clocks: HIP events on the simulation's stream (`roofline`) and the host's clock from the
    # first launch to the drained stream (`value`; read inside pbSimStepTimedWall, next to the launches, so that the
    # interpreter's own call overhead -- ~20 us, 1 % of the driver's 20-step region -- is not billed to the device)
    done, dev_ms, wall_ms = sim.step_timed_wall(args.steps)
    wall = wall_ms * 1e-3   # this rank's K steps are complete (MAX over ranks below; then the closing barrier)
    if dist is not None:
        K.dev_sync(torch)
        dist.barrier()
    s1 = sim.stats()
    assert done == args.steps, (done, args.steps)
    launches = (s1["fused_launches"] - s0["fused_launches"]) + (s1["plain_launches"] - s0["plain_launches"])
    # a timed region under 50 ms of device time (the driver's --steps 20 is ~2 ms) is followed at once by one of
    # >= 100 ms of the same simulation, reported beside it (`long`)
    long = None
    if dev_ms < K.SHORT_MS:
        k = min(int(K.LONG_MS / max(dev_ms / done, 1e-6)) + 1, 400000)
        d2, ms2 = sim.step_timed(k)
        long = {"steps": d2, "avg_launch_us": ms2 * 1e3 / max(d2, 1)}
    cx, cy = sim.centroid()
    assert cx == cx and cy == cy, "simulation state went NaN: the benchmark workload is invalid"
    summary = [sim.time, cx, cy]
    sim.close()
    return {"wall": wall, "dev_ms": dev_ms, "launches": launches, "long": long, "summary": summary, "cfg": cfg,
            "kernel": kernel, "prewarm": prewarm}


def default_form_leg(pb, n, steps, warmup, warm):
    """The same arena in the library's default form (pbSimSetForceSums mode 0: no absForce_a, 56 B per particle-step)."""
    sim = K.make_sim(pb, n, K.LATTICE_PITCH, seed=1, force_sums=0)
    cfg, kernel = sim.config(), sim.force_kernel_name()
    t = K.timed_leg(sim, warm, warmup, steps)
    sim.close()
    us = t.get("us_per_step_long", t["us_per_step"])
    tr, why = K.matched_profile("latest_traffic.json", kernel)
    out = {"value": n / (us * 1e-6), "ms_per_step": us * 1e-3, "steps": t.get("steps_long", t["steps"]),
           "frac_at_56B": K.ALG_BYTES_DEAD_SUM * n / (us * 1e-6) / 1e9 / K.HBM_PEAK_GBS, "kernel": kernel.split("(")[0],
           "attraction_sums": cfg["attraction_sums"], "traffic": tr["hbm_bytes_per_launch"] if tr else None,
           "valu_frac_of_datasheet": K.valu_of_datasheet(tr, n, us)}
    return out, {"timed": t, "config": cfg, "kernel_signature": kernel, "profile": tr, "profile_dropped": why}


def configs3_end_to_end(rank, world, dist, torch, e2e_steps):
    """BASELINE configs[3] as written: 256 + 256 seeds over all ranks, full length, end to end (collective)."""
    r = K.ensemble_end_to_end("ensemble4", rank, world, dist, torch, None, 256, e2e_steps)
    if r is None:
        return None, None
    short = {"workload": "configs[3]: 256 obstacle + 256 object-transport seeds, 500/201 bots, member k on rank k mod N, "
                         "end to end (placement + upload + steps + RCCL gather)",
             "members_total": r["members_total"], "steps_per_member": r["steps_per_member"], "wall_s": r["wall_s"],
             "value_end_to_end": r["value_end_to_end"], "sims_per_s": r["sims_per_s_end_to_end"], "scaling": "strong",
             "rows_gathered": r["rows_gathered"]}
    return short, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--bots", type=int, default=1_000_000, help="bots per arena (the metric is quoted at 10^6)")
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="device time of scratch work right before every timed leg (clock ramp; never timed); 0 disables")
    ap.add_argument("--cpu-seconds", type=float, default=4.0,
                    help="time budget of the cpu_baseline sample on all usable cores (+ a quarter of it on one thread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ensemble", action="store_true", help="skip the configs[3] end-to-end run")
    ap.add_argument("--e2e-steps", type=int, default=None, help="bound the timesteps per member of the configs[3] run")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the long record goes")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even with one rank (the N>1 code path)")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0)
    ap.add_argument("--dry-run-device", action="store_true",
                    help="TEST ONLY (tests/test_bench_multirank.py): gloo, no GPU, the arena replaced by a counter")
    args = ap.parse_args()
    t_start, phases = time.perf_counter(), {}   # (detail record: seconds since start at the end of each phase)
    K.DRY = args.dry_run_device
    K.HEADLINE_VARIANT = 2
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (the two configs[3] batches must not share a hardware queue)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(K.spawn_ranks(args, __file__))
    rank, local_rank, world, dist, torch = K.init_ranks(args)

    import particlerobotsimulations_amd as pb
    dev = None
    if K.DRY:
        pb = K._DryPb
    elif dist is None:
        pb.legacy.cudaInit(0, None)
    else:
        dev = K.engine_device(local_rank)   # (torch.cuda.set_device chose it for torch; the engine is told itself)

    n = args.bots
    K.HEADLINE_FORCE_SUMS = 1   # the scratch arena runs the headline's kernel
    warm = K.DevicePrewarm(pb, min(n, 1_000_000), K.LATTICE_PITCH, args.prewarm_ms)
    h = timed_arena(pb, args, rank, world, dist, torch, warm)
    phases["headline"] = time.perf_counter() - t_start
    wall, summaries, coll = h["wall"], [h["summary"]], None
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=K.dist_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # the only data exchange of an ensemble of arenas: per-arena summaries (time, COMx, COMy), gathered
        mine = torch.tensor(h["summary"], dtype=torch.float64, device=K.dist_device())
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        summaries = [[float(x) for x in v.tolist()] for v in allv]
        coll = K.collective_info(dist, torch, local_rank, dev)
    warm.done()
    # (the arenas are closed before the ensemble: with their streams alive the two batches land on one hardware queue)
    ens, ens_detail = (None, None) if args.no_ensemble else configs3_end_to_end(rank, world, dist, torch, args.e2e_steps)
    phases["ensemble"] = time.perf_counter() - t_start

    if rank == 0:
        avg_us = h["dev_ms"] * 1e3 / max(h["launches"], 1)
        achieved = K.ALG_BYTES_PER_PARTICLE_STEP * n / (avg_us * 1e-6) / 1e9
        tr, why = K.matched_profile("latest_traffic_both_sums.json", h["kernel"]) if n == 1_000_000 else (None, "not 10^6 bots")
        roof = {"bound": "valu", "achieved": achieved, "peak": K.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / K.HBM_PEAK_GBS, "kernel": h["kernel"].split("(")[0], "avg_launch_us": avg_us,
                "launches": h["launches"], "alg_bytes_per_launch": K.ALG_BYTES_PER_PARTICLE_STEP * n,
                "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                "valu_frac_of_datasheet": K.valu_of_datasheet(tr, n, avg_us),
                "profile": ({"name": tr["profile"], "commit": (tr.get("build") or {}).get("commit"),
                             "sources_match": tr["sources_match"]} if tr else {"dropped": why})}
        if h["long"]:
            lu = h["long"]["avg_launch_us"]
            roof["long"] = dict(h["long"], frac=K.ALG_BYTES_PER_PARTICLE_STEP * n / (lu * 1e-6) / 1e9 / K.HBM_PEAK_GBS)
        out = {"metric": K.METRIC, "value": world * n * args.steps / wall, "unit": "particle-steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[2]: synthetic phototaxis arena, square lattice of oscillating bots at pitch "
                                      f"{K.LATTICE_PITCH}, one light at (-230,0), 2048^2 grid, walls +-240, phase_std 0",
                          "bots_per_gpu": n, "dt": 0.01, "sort_interval": 180.0,
                          "force_form": "exact k_force, both magnitude sums (everything collideD writes)",
                          "attraction_sums": h["cfg"]["attraction_sums"], "force_variant": h["cfg"]["force_variant"],
                          "lanes_per_bot": h["cfg"]["lanes_per_bot"],
                          "parallelism": "single arena" if world == 1 else f"{world} independent arenas, one per GPU"},
               "roofline": roof, "device_ms_timed_region": h["dev_ms"], "collective": coll}
        detail = {"headline": h, "summaries_time_comx_comy": summaries, "host": K.host_info(),
                  "build": K.loaded_build_stamp(), "profile_both_sums": tr, "ensemble": ens_detail}
        K.HEADLINE_FORCE_SUMS = 0
        warm = K.DevicePrewarm(pb, min(n, 1_000_000), K.LATTICE_PITCH, args.prewarm_ms)
        out["default_form"], detail["default_form"] = default_form_leg(pb, n, min(args.steps, 400),
                                                                       max(args.warmup, 100), warm)
        warm.done()
        phases["default_form"] = time.perf_counter() - t_start
        if ens is not None:
            out["ensemble"] = ens
        if world == 1 and not args.no_cpu_baseline:
            c = K.cpu_baseline(n, K.LATTICE_PITCH, args.cpu_seconds)
            detail["cpu_baseline"] = c
            out["cpu_baseline"] = {k: c[k] for k in ("value", "unit", "cores", "value_1_thread", "kind", "sample")}
            phases["cpu_baseline"] = time.perf_counter() - t_start
        detail["phases_end_s"] = phases
        out["detail"] = os.path.relpath(args.detail, ROOT) if args.detail else None
        # the driver reads an 8 KB tail: should the line ever outgrow its budget, the optional blocks go (they are in
        # the detail record), never the contract keys
        import json
        for drop in ("ensemble", "default_form", "collective"):
            if len(json.dumps(out)) + 1 < MAX_LINE_BYTES:
                break
            sys.stderr.write(f"bench.py: line over {MAX_LINE_BYTES} bytes, dropping `{drop}` (kept in {args.detail})\n")
            out.pop(drop, None)
        K.emit(out, detail, args.detail)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
