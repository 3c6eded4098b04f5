#!/usr/bin/env python3
"""bench_legs.py -- every measurement of the particle-robot update loop that is NOT the headline line of bench.py.

  python tools/bench_legs.py [--gpus N] [--steps K] [--warmup W] [--bots B] [--workload arena|ensemble4|ensemble5] ...

--workload arena (BASELINE configs[2], the 10^6-bot lattice): the arena in the library's default form, then the legs --
the same arena with both magnitude sums, the opt-in streamlined kernel, 8 x 10^6 bots, a random blob, the
survey-literal hex lattice, the host round trip, the shader clock, the configs[3] ensemble beside it.
--workload ensemble4 / ensemble5: BASELINE configs[3] / configs[4] as batched ensembles, member k on rank k mod N, the
summary rows gathered over RCCL, steady state and end to end.  --force-variant / --force-sums put ONE kernel under
tools/profile.sh.  One (long) JSON line on stdout; bench.py prints the short one the driver parses."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchkit as K
from benchkit import (ALG_BYTES_DEAD_SUM, ALG_BYTES_PER_PARTICLE_STEP, ENSEMBLE_WORKLOADS, FULL_RUN, HBM_PEAK_GBS,
                      LATTICE_PITCH, LONG_MS, PAIRS_PER_BOT_LATTICE, ROOT, SHORT_MS, VALU_PEAK_LANE_OPS, DevicePrewarm,
                      cpu_baseline, dev_sync, dist_device, emit, ensemble_batches, ensemble_end_to_end, host_info,
                      leg_fields, make_sim, profiled_traffic, timed_leg, valu_of_datasheet, workload_params)


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


def streamlined_leg(pb, n, pitch, steps, warmup, warm=None):
    """The opt-in streamlined force arithmetic (force variant 3; NOT bit-identical, DESIGN.md
    "Streamlined") on the same workload: its throughput over `steps` steps (device time, HIP events; pre-warmed,
    nothing between warm-up and timing), THEN its deviation from the exact kernel over one 10-step window from a
    common state."""
    import numpy as np
    fast = make_sim(pb, n, pitch, seed=1)
    fast.set_force_variant(3)
    t = timed_leg(fast, warm, warmup, steps)
    cx, cy = fast.centroid()
    # parity window: both kernels from the state the timed run ended in
    st = fast.get_state()
    ta = fast.time
    exact, fast2 = make_sim(pb, n, pitch, seed=1), make_sim(pb, n, pitch, seed=1)
    for sim, variant in ((exact, 2), (fast2, 3)):
        # (a fresh simulation object has no cell lists yet: both copies re-sort at their first step, from the
        #  same positions)
        sim.set_state(pos=st["pos"], vel=st["vel"], rad=st["rad"], phase=st["phase"], dead=st["dead"])
        sim.set_forces(st["absForce_a"] if st["absForce_a"] is not None else np.zeros(n, np.float32), st["absForce_r"])
        sim.time = ta
        sim.set_force_variant(variant)
    fast.close()
    exact.step(10)
    fast2.step(10)
    a, b = exact.get_state()["pos"].astype(np.float64), fast2.get_state()["pos"].astype(np.float64)
    exact.close()
    fast2.close()
    d = np.linalg.norm(b - a, axis=1)
    rel = d / np.maximum(np.linalg.norm(a, axis=1), 1.0)
    com = float(np.linalg.norm(a.mean(0) - b.mean(0)))
    out = leg_fields(t, n)
    us = out.get("us_per_step_long", out["us_per_step"])
    achieved = ALG_BYTES_PER_PARTICLE_STEP * n / (us * 1e-6) / 1e9
    out.update({"ms_per_step": out["us_per_step"] * 1e-3, "finite_at_end": bool(cx == cx and cy == cy),
            "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "kernel": "k_force_stream",
                         "over": "us_per_step_long" if "us_per_step_long" in out else "us_per_step"},
            "parity": {"against": "the exact kernel over 10 steps from the state the timed run ended in; the exact kernel is "
                                  "bit-identical to the CPU oracle on this very workload "
                                  "(tests/test_gpu_baseline_configs.py::test_bench_headline_workload_matches_oracle); "
                                  "oracle-side flip statistics at 10^6 bots and on blobs: tests/test_gpu_streamlined.py "
                                  "(bench.py may use oracle/ only in cpu_baseline)",
                       "vs_fma_bracket": fma_bracket_summary(),
                       "window_steps": 10, "max_abs_dpos": float(d.max()),
                       "median_abs_dpos": float(np.median(d)), "bots_beyond_1e-5_relative": int((rel > 1e-5).sum()),
                       "com_abs_dev": com},
            "note": "opt-in: pbSimSetForceVariant(sim, 3).  v_rsq/v_rcp/FMA arithmetic, |F_attr| taken from its "
                    "coefficient, contact terms added after attraction terms.  Not bit-identical; held to 1e-5 "
                    "relative over 10-step windows by tests/test_gpu_streamlined.py.  `value` of the line is the exact "
                    "kernel."})
    return out


def fma_bracket_summary():
    """The streamlined kernel against the FMA / __powf bracket of the reference's own arithmetic, from the committed
    record of tests/test_gpu_fma_bracket.py on MI355X (tests/golden/fma_bracket/hip_streamlined.json: teacher-forced
    10-step windows of every BASELINE config, the oracle as teacher, oracle/libpb_oracle_fma[_powf].so as the
    bracket).  bench.py itself runs no oracle code outside cpu_baseline; None if the record is absent."""
    path = os.path.join(ROOT, "tests", "golden", "fma_bracket", "hip_streamlined.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None
    tot = {}
    for case in rec["cases"].values():
        for cand, recs in case["candidates"].items():
            t = tot.setdefault(cand, {"flips": 0, "bot_windows": 0, "p99_worst": 0.0, "com_rel_worst": 0.0})
            for r in recs:
                w = r["window"]
                t["flips"] += w["flips"]
                t["bot_windows"] += case["bots"]
                t["p99_worst"] = max(t["p99_worst"], w["p99"])
                if case["case"] != "cfg3_arena_crop_10k":   # centred on the origin: |COM| ~ 0, relative figure meaningless
                    t["com_rel_worst"] = max(t["com_rel_worst"], w["com_rel"])
    rate = lambda c: (tot[c]["flips"] / max(tot[c]["bot_windows"], 1)) if c in tot else None
    bracket = max(r for r in (rate("fma"), rate("fma_powf")) if r is not None)
    return {"flip_rate_streamlined": rate("hip_streamlined"), "flip_rate_fma": rate("fma"),
            "flip_rate_fma_powf": rate("fma_powf"),
            "ratio": (rate("hip_streamlined") / bracket) if bracket > 0 else None,
            "flips_streamlined": tot["hip_streamlined"]["flips"], "flips_fma": tot["fma"]["flips"],
            "bot_windows": tot["hip_streamlined"]["bot_windows"],
            "p99_worst": {c: tot[c]["p99_worst"] for c in tot}, "com_rel_worst": {c: tot[c]["com_rel_worst"] for c in tot},
            "source": "tests/golden/fma_bracket/hip_streamlined.json (tests/test_gpu_fma_bracket.py on MI355X)",
            "note": "flip = a bot more than 1e-5 relative from the oracle after a teacher-forced 10-step window; the "
                    "bracket is the oracle's own source with its kernels FMA-contracted (what nvcc -fmad=true does "
                    "to the reference) and with exp2f(2*log2f(x)) for __powf: the reference's build-to-build spread"}


def both_sums_leg(pb, n, pitch, steps, warmup, warm=None):
    """The same workload with BOTH magnitude sums maintained (pbSimSetForceSums mode 1: what a batch with
    constrained_contraction set runs), so the line shows what leaving out the dead Sum|F_attr| is worth.
    Positions, velocities, radii, phases and absForce_r are bit-identical in the two modes
    (tests/test_gpu_dead_sum.py)."""
    sim = make_sim(pb, n, pitch, seed=1)
    sim.set_force_sums(1)
    t = timed_leg(sim, warm, warmup, steps)
    cfg = sim.config()
    sim.close()
    out = leg_fields(t, n)
    out.update({"attraction_sums": cfg["attraction_sums"], "dead_sum_form": cfg["dead_sum_form"]})
    return out


def host_round_trip_leg(pb, n, pitch, steps=10):
    """What the boundary costs a caller who does NOT keep the state on the device: pbSimSetState from host buffers
    (28 B per bot: pos, vel, rad, phase, dead), one step, pbSimGetState into host buffers (36 B per bot: the same
    plus the two force sums), every step, pageable numpy memory as a ctypes caller has it, buffers reused.  Never `value`: the
    class keeps the state resident between CSV dumps, as the reference does (particlebot.cpp:383-395 only reads
    back for dumpParticlebot).  DESIGN.md section 5 quotes this leg."""
    sim = make_sim(pb, n, pitch, seed=1)
    sim.step(50)
    st = sim.get_state()
    t_all = t_up = t_down = t_step = 0.0
    for i in range(steps + 2):
        t0 = time.perf_counter()
        sim.set_state(pos=st["pos"], vel=st["vel"], rad=st["rad"], phase=st["phase"], dead=st["dead"])
        sim.synchronize()
        t1 = time.perf_counter()
        sim.step(1)
        sim.synchronize()
        t2 = time.perf_counter()
        st = sim.get_state(out=st)  # the caller's buffers are reused: no fresh pages in the timed copies
        t3 = time.perf_counter()
        if i >= 2:
            t_up += t1 - t0
            t_step += t2 - t1
            t_down += t3 - t2
            t_all += t3 - t0
    sim.close()
    up_b, down_b = 28 * n, 36 * n
    return {"value": n * steps / t_all, "unit": "particle-steps/s", "steps": steps,
            "ms_per_step": 1e3 * t_all / steps, "upload_ms": 1e3 * t_up / steps, "step_ms": 1e3 * t_step / steps,
            "download_ms": 1e3 * t_down / steps, "upload_GBps": up_b * steps / t_up / 1e9,
            "download_GBps": down_b * steps / t_down / 1e9, "bytes_per_bot_per_step": 64,
            "note": "SetState + 1 step + GetState through pageable host buffers every step; never `value`"}




def hbm_target_note(n):
    """north_star's 60 % of the HBM roofline, restated in the unit that binds."""
    us = ALG_BYTES_PER_PARTICLE_STEP * n / (0.60 * HBM_PEAK_GBS * 1e9) * 1e6
    per_bot = VALU_PEAK_LANE_OPS * us * 1e-6 / n
    return (f"60 % of the HBM roofline = {0.6 * HBM_PEAK_GBS / 1e3:.1f} TB/s at 64 B per particle-step = {us:.1f} us per "
            f"step of {n} bots; at 100 % of the fp32 vector peak ({VALU_PEAK_LANE_OPS / 1e12:.1f} T lane-instructions/s) "
            f"that is {per_bot:.0f} VALU instructions per bot = {per_bot / PAIRS_PER_BOT_LATTICE:.0f} per candidate "
            f"pair ({PAIRS_PER_BOT_LATTICE} pairs per bot on this lattice) -- fewer than the one v_rsq_f32 and one "
            "v_rcp_f32 (4 issue slots each) plus the ~10 simple instructions the reference's pair law needs before "
            "any force is formed: the target is out of reach for this physics, and valu_frac_of_datasheet is the "
            "fraction to read")


SIMDS = 1024                 # 256 CUs x 4 SIMD-32 (MI355X_MICROARCH.md)
DATASHEET_CYC_SIMPLE = 2.0   # cycles per wave64 VALU instruction per SIMD ("v_fma_f32 (wave64) 2 cyc")
DATASHEET_CYC_TRANS = 8.0    # v_rcp/v_sqrt/v_rsq: quarter rate (8 lanes/clk; the guide's issue cost 8; tools/valu_rate measures 9.1-9.3)
NOMINAL_MHZ = 2400.0


def valu_roofline(tr, n, avg_launch_us, clock_mhz):
    """The force kernel's VALU instruction stream (PMC counts per wave from the committed profile)
    priced two ways against this run's launch time: (a) at the datasheet issue rate -- 2 cycles per
    wave64 instruction per SIMD-32, 8 for a quarter-rate transcendental -- at the shader clock MEASURED under this
    load (and, for reference, at the 2.4 GHz nominal clock); (b) at the rates tools/valu_rate measured
    on the profiled box at 8 waves per SIMD."""
    if not tr or "valu_insts_per_wave" not in tr or "trans_per_wave" not in tr:
        return None
    per_wave, trans = tr["valu_insts_per_wave"], tr["trans_per_wave"]
    waves_per_simd = (n / 64.0) / SIMDS
    cycles = ((per_wave - trans) * DATASHEET_CYC_SIMPLE + trans * DATASHEET_CYC_TRANS) * waves_per_simd
    out = {"valu_insts_per_wave": per_wave, "trans_per_wave": trans, "waves_per_simd": waves_per_simd,
           "datasheet_cycles_per_simd": cycles,
           "datasheet_rate": {"cycles_per_simple": DATASHEET_CYC_SIMPLE, "cycles_per_trans": DATASHEET_CYC_TRANS},
           "shader_clock_mhz_measured": clock_mhz,
           "frac_datasheet_at_nominal_clock": cycles / NOMINAL_MHZ / avg_launch_us,
           "frac_datasheet_at_measured_clock": (cycles / clock_mhz / avg_launch_us) if clock_mhz else None,
           "source": "profiles/latest_traffic.json (SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32, SQ_WAVES)"}
    if "ns_simple" in tr and "ns_trans" in tr:
        us = ((per_wave - trans) * tr["ns_simple"] + trans * tr["ns_trans"]) * waves_per_simd * 1e-3
        out["frac_microbenchmark_rate"] = us / avg_launch_us
        out["microbenchmark_rate"] = {"ns_simple": tr["ns_simple"], "ns_trans": tr["ns_trans"],
                                      "in_kernel_mhz": tr.get("valu_rate_mhz"),
                                      "source": "tools/valu_rate at 8 waves/SIMD on the profiled box"}
    for k in ("wave_cycle_split", "valu_lane_utilisation"):
        if k in tr:
            out[k] = tr[k]
    # the same figure as roofline.valu_frac_of_datasheet (instructions x 64 lanes / time / 78.6e12), under the name the
    # round-4 review used
    out["frac_of_datasheet"] = valu_of_datasheet(tr, n, avg_launch_us)
    return out


def measure_clock(pb, sim, ms_per_step, span=0.15):
    """Shader clock held while the force kernel runs: a sleeping sampler wave on its own stream spans
    `span` seconds of real time while the same simulation keeps stepping (1.3 x the span's worth of
    steps, so the sampler never sees an idle device; NOT part of `value`'s timed region)."""
    try:
        steps = int(span * 1.3 / max(ms_per_step * 1e-3, 1e-6)) + 50
        sim.step(200)                # the legs before this one may have let the clocks drop
        smp = pb.ClockSample(span)
        sim.step(steps)
        sim.synchronize()
        return smp.end(), span
    except Exception as e:  # diagnostic only
        return None, str(e)


def large_arena_leg(pb, pitch, warmup, steps, n=8_000_000, warm=None):
    """SURVEY 8(d) caveat 2: the same lattice at 8 x 10^6 bots (544 MB of state, beyond the 256 MiB
    Infinity Cache) to show the kernel's sensitivity to true HBM traffic."""
    sim = make_sim(pb, n, pitch, seed=1)
    t = timed_leg(sim, warm, warmup, steps)
    cx, cy = sim.centroid()
    cfg = sim.config()
    sim.close()
    out = leg_fields(t, n)
    us = out.get("us_per_step_long", out["us_per_step"])
    achieved = ALG_BYTES_PER_PARTICLE_STEP * n / (us * 1e-6) / 1e9
    out.update({"bots": n, "warmup": warmup, "us_per_step_per_1e6_bots": out["us_per_step"] / (n / 1e6),
                "state_bytes": 68 * n, "finite_at_end": bool(cx == cx and cy == cy),
                "force_variant": cfg["force_variant"], "lanes_per_bot": cfg["lanes_per_bot"],
                "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS,
                             "over": "us_per_step_long" if "us_per_step_long" in out else "us_per_step"},
                "note": "working set beyond the Infinity Cache: if HBM bound the step would cost > 8x the 10^6-bot one"})
    if "us_per_step_long" in out:
        out["us_per_step_per_1e6_bots_long"] = out["us_per_step_long"] / (n / 1e6)
    return out


class BlobPlacement:
    """The 10^6-bot random blob of the `random_blob` leg, grown on a host thread from the moment bench.py starts
    (4-14 s of one core, `pb_placement fastblob`) while the device legs before it run: host work that used to sit
    between device legs and leave the GPU idle for seconds."""

    def __init__(self, n):
        import threading
        self.n, self.pos, self.place_s, self.err = n, None, None, None
        self._t = threading.Thread(target=self._work, daemon=True)
        self._t.start()

    def _work(self):
        try:
            from particlerobotsimulations_amd import host
            t0 = time.perf_counter()
            h = host.HostSim(os.path.join(ROOT, "examples", "million_bot_blob.cfg"), engine="host", nCells=str(self.n))
            self.place_s = time.perf_counter() - t0
            self.pos = h.get("pos")
            h.close()
        except Exception as e:  # reported by the leg
            self.err = e

    def get(self):
        self._t.join()
        if self.err is not None:
            raise self.err
        return self.pos, self.place_s


def blob_leg(pb, n, steps, warmup, placement, warm=None):
    """the same arena holding a RANDOM BLOB of n bots grown by the reference's placement
    rule with the O(N) generator (`pb_placement fastblob`, Particlebot::placeFastBlob) instead of the
    lattice: the reference's own kind of initial state at a size its O(N^1.5) loop cannot reach."""
    import numpy as np
    pos, place_s = placement.get()
    sp, keep = workload_params(n, seed=1)
    sim = pb.Sim(sp, wall_half=240.0, keepalive=keep)
    sim.set_force_variant(2)
    sim.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                  phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    t = timed_leg(sim, warm, warmup, steps)
    st = sim.get_state()
    cx, cy = sim.centroid()
    sim.close()
    out = leg_fields(t, n)
    us = out.get("us_per_step_long", out["us_per_step"])
    achieved = ALG_BYTES_PER_PARTICLE_STEP * n / (us * 1e-6) / 1e9
    out.update({"bots": n, "placement": "pb_placement fastblob (examples/million_bot_blob.cfg), on a host thread "
                                        "beside the legs before this one", "placement_s": place_s,
                "warmup": warmup, "finite_at_end": bool(cx == cx and cy == cy),
                "bots_in_contact_frac": float((st["absForce_r"] > 0).mean()),
                "max_speed": float(np.abs(st["vel"]).max()),
                "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS,
                             "over": "us_per_step_long" if "us_per_step_long" in out else "us_per_step"}})
    return out




def measure_ensemble(pb, workload, members_per_gpu, steps, warmup, prewarm_ms, rank, world, dist, torch,
                     members_total=None, end_to_end=True, e2e_steps=None, strong_total=None, host_threads=0):
    """K timesteps of an ensemble workload on every rank (member k on rank k mod N), then the path's one
    exchange (the summary rows, over RCCL when there is a process group); then (end_to_end) the same ensemble run
    end to end at full length through the placement/stepping pipeline.  Collective: every rank calls
    it.  Returns (result dict on rank 0 else None, this rank's batches)."""
    import threading

    import numpy as np
    from particlerobotsimulations_amd import ensemble
    warm = DevicePrewarm(pb, 250_000, LATTICE_PITCH, prewarm_ms)
    batches = ensemble_batches(workload, rank, world, members_per_gpu, members_total)
    t_place = time.perf_counter()
    ens = [(K._DryLocalEnsemble if K.DRY else ensemble.LocalEnsemble)(cfg, over, common) for cfg, common, over, _ in batches]
    t_place = time.perf_counter() - t_place

    def drive(nsteps):
        """nsteps timesteps of every member; the batches of this rank run concurrently (one host
        thread per batch: each pbSim has its own HIP stream, ctypes releases the GIL)."""
        done = [0] * len(ens)

        def one(i):
            done[i] = ens[i].run_steps(nsteps)
        th = [threading.Thread(target=one, args=(i,)) for i in range(1, len(ens))]
        for t in th:
            t.start()
        one(0)
        for t in th:
            t.join()
        return done

    def barrier():
        for e in ens:
            e.synchronize()
        if dist is not None:
            dev_sync(torch)
            dist.barrier()

    def timed(nsteps):
        barrier()
        t0 = time.perf_counter()
        done = drive(nsteps)
        for e in ens:
            e.synchronize()
        wall = time.perf_counter() - t0   # this rank's steps are complete; MAX over ranks below; the barrier after it
        barrier()
        assert all(d == nsteps for d in done), (done, nsteps)
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device=dist_device())
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall

    prewarm = warm.run()
    drive(warmup)
    wall = timed(steps)
    # a timed region under 50 ms is followed by one of >= 100 ms (every rank takes the same decision: the wall
    # time is already the max over ranks)
    wall_long, steps_long = None, None
    if wall < SHORT_MS * 1e-3:
        steps_long = min(int(LONG_MS * 1e-3 / (wall / steps)) + 1, 200000)
        wall_long = timed(steps_long)
    warm.done()
    # the path's only exchange: every member's summary rows, gathered once over RCCL
    total_members = members_total if members_total is not None else members_per_gpu * world
    gathered = [ensemble.gather_summaries(e.rows, total_members, rank, world, dist,
                                          dist_device() if dist is not None else "cpu") for e in ens]
    bots = [e.n for e in ens]
    mine = [e.m for e in ens]
    for e in ens:
        e.close()
    e2e = strong = None
    if end_to_end:
        e2e = ensemble_end_to_end(workload, rank, world, dist, torch, members_per_gpu, members_total, e2e_steps,
                                  host_threads=host_threads)
        if strong_total is not None and members_total is None:
            strong = ensemble_end_to_end(workload, rank, world, dist, torch, None, strong_total, e2e_steps,
                                         host_threads=host_threads)
        # When the host is the limit (few cores per rank: the reference's placement rule costs 0.85-1.6 CPU-seconds per
        # 10^5-bot member) the same run is repeated with the O(N) generator (pb_placement fastblob, HISTORY.md 6c), so
        # that the line shows both what the reference's rule costs here and what the device can do.  Every rank takes
        # the same decision: rank 0's verdict is broadcast.
        fast = None
        if workload == "ensemble5":
            # (--dry-run-device: no timings, hence never host-bound, but the broadcast below still runs under gloo)
            host_bound = bool(e2e and e2e.get("bound_rank0") and any(b["bound"] == "host" for b in e2e["bound_rank0"]))
            if dist is not None:
                flag = torch.tensor([1 if host_bound else 0], dtype=torch.int32, device=dist_device())
                dist.broadcast(flag, src=0)
                host_bound = bool(flag.item())
            if host_bound:
                fast = ensemble_end_to_end(workload, rank, world, dist, torch, members_per_gpu, members_total, e2e_steps,
                                           host_threads=host_threads, extra_common={"pb_placement": "fastblob"})
    if rank != 0:
        return None, batches
    all_bots = sum(b * total_members for b in bots)      # bots stepped per timestep over all ranks
    achieved = ALG_BYTES_PER_PARTICLE_STEP * (all_bots / world) * steps / wall / 1e9
    last = [g[:, -1] for g in gathered]
    assert all(np.isfinite(l).all() for l in last), "an ensemble member went NaN"
    out = {
        "value": all_bots * steps / wall, "unit": "particle-steps/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": wall * 1e3 / steps, "scaling": "strong" if members_total is not None else "weak",
        "config": {"workload": f"{workload}: {ENSEMBLE_WORKLOADS[workload]}",
                   "members_per_gpu": (members_per_gpu * len(bots)) if members_total is None else None,
                   "members_total": total_members * len(bots),
                   "bots_per_member": bots, "dt": 0.01,
                   "parallelism": (f"member k -> rank k mod {world}; one batched pbSim per .cfg per GPU; "
                                   f"RCCL world size {dist.get_world_size()}" if dist is not None
                                   else "one GPU, no process group"),
                   "members_per_rank": [len(ensemble.shard(total_members, r, world)) * len(bots)
                                        for r in range(world)]},
        "placement_s": t_place, "device_prewarm": prewarm,
        "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_resident (<= 1024-bot members) / k_force (larger members)",
                     "note": "64 algorithmic bytes per particle-step over WALL time of the timed region "
                             "(host-driven schedule included); small members are latency-bound (DESIGN.md section 6)"},
        "summaries_last_row_time_comx_comy_dist": [[[float(x) for x in r] for r in l[:4]] for l in last],
        "summary_rows_gathered": [list(g.shape) for g in gathered],
    }
    if wall_long is not None:
        out.update(value_long=all_bots * steps_long / wall_long, steps_long=steps_long,
                   ms_per_step_long=wall_long * 1e3 / steps_long)
    if e2e is not None:
        out["end_to_end"] = e2e
        out["value_end_to_end"] = e2e["value_end_to_end"]
        out["sims_per_s_end_to_end"] = e2e["sims_per_s_end_to_end"]
    if strong is not None:
        out["strong_end_to_end"] = strong
    if end_to_end and workload == "ensemble5" and e2e is not None and not K.DRY:
        out["end_to_end_bound"] = "host" if any(b["bound"] == "host" for b in e2e["bound_rank0"]) else "device"
        if fast is not None:
            out["end_to_end_fastblob"] = fast
    return out, batches


def run_ensemble_workload(args, rank, world, dist, torch):
    import particlerobotsimulations_amd as pb
    if K.DRY:
        pb = K._DryPb
    res, batches = measure_ensemble(pb, args.workload, args.members_per_gpu, args.steps, args.warmup, args.prewarm_ms,
                                    rank, world, dist, torch, members_total=args.members_total,
                                    end_to_end=not args.no_end_to_end, e2e_steps=args.e2e_steps,
                                    host_threads=args.host_threads)
    if rank == 0:
        out = {"metric": "particle-steps/sec at 10^6 bots; achieved HBM GB/s vs peak; 1/2/4/8-GPU ensemble",
               "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "host": host_info()}
        out.update(res)
        if K.ENSEMBLE_FORCE_VARIANT is not None:
            out["headline"] = False
            out["config"]["force_variant"] = K.ENSEMBLE_FORCE_VARIANT
            out["force_variant_note"] = ("pb_force_variant set for every member: 3 = the opt-in tolerance kernel for "
                                         "batches in the throughput form (not bit-identical; DESIGN.md section 4)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_ensemble(batches, min(args.cpu_seconds, 10.0))
        emit(out)


def cpu_baseline_ensemble(batches, budget_s):
    """The oracle (CPU port; the reference has no CPU path) on ONE member of each batch for a bounded
    time: placement excluded, timesteps only."""
    from oracle import orclib
    cores = orclib.usable_cpus()
    orclib.lib().orc_set_num_threads(cores)
    work, el = 0.0, 0.0
    parts = []
    for cfg, common, over, _ in batches:
        kv = dict(common)
        lines = over[0].split("\n")
        kv.update({lines[i]: lines[i + 1] for i in range(0, len(lines), 2)})
        P = orclib.OrcParams()
        L = orclib.lib()
        import ctypes as C
        L.orc_params_defaults(C.byref(P))
        L.orc_load_cfg(C.byref(P), os.fsencode(cfg))
        for k, v in kv.items():
            L.orc_set_param(C.byref(P), k.encode(), str(v).encode())
        L.orc_params_derive(C.byref(P), 0, 0.0)
        sim = orclib.Sim(P)
        sim.run(1)
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget_s / len(batches) and steps < 100000:
            sim.run(10)
            steps += 10
        dt = time.perf_counter() - t0
        work += float(P.nCells) * steps
        el += dt
        parts.append(f"{steps} steps of one {P.nCells}-bot member of {os.path.basename(cfg)}")
        sim.close()
    return {"value": work / el, "unit": "particle-steps/s", "cores": orclib.lib().orc_num_threads(), "kind": "port",
            "sample": "; ".join(parts) + " (OpenMP over bots; reported, not optimised)"}




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--bots", type=int, default=1_000_000)
    ap.add_argument("--pitch", type=float, default=LATTICE_PITCH)
    ap.add_argument("--workload", choices=["arena", "ensemble4", "ensemble5"], default="arena",
                    help="arena: the 10^6-bot headline (BASELINE configs[2]); ensemble4 / ensemble5: BASELINE "
                         "configs[3] / configs[4] as batched ensembles sharded member k -> rank k mod N")
    ap.add_argument("--members-per-gpu", type=int, default=None,
                    help="ensemble workloads: members per GPU and per .cfg (default 32 for ensemble4, 8 for ensemble5)")
    ap.add_argument("--members-total", type=int, default=None,
                    help="ensemble workloads: a FIXED number of members per .cfg over all GPUs (strong scaling: "
                         "BASELINE configs[3] is 256, configs[4] 1024) instead of --members-per-gpu per GPU (weak)")
    ap.add_argument("--e2e-steps", type=int, default=None,
                    help="bound the timesteps per member of the ensemble end-to-end run (default: the configuration's "
                         "full length, 120000 for ensemble4 and 12000 for ensemble5)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the ensemble end-to-end run")
    ap.add_argument("--sub-batch", type=int, default=None,
                    help="members per sub-batch of the end-to-end pipeline (default: FULL_RUN's value for the workload)")
    ap.add_argument("--lanes", type=int, default=None,
                    help="ensemble workloads, end to end: sub-batches stepped at the same time (default: 2 with the "
                         "automatic sub-batch, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-survey-literal", action="store_true")
    ap.add_argument("--no-streamlined", action="store_true")
    ap.add_argument("--no-both-sums", action="store_true")
    ap.add_argument("--no-host-round-trip", action="store_true")
    ap.add_argument("--no-large-arena", action="store_true")
    ap.add_argument("--no-clock", action="store_true")
    ap.add_argument("--no-blob", action="store_true")
    ap.add_argument("--no-ensemble-leg", action="store_true")
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="device time of scratch work before the measured simulation (clock ramp); 0 disables")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="time budget of the cpu_baseline sample")
    ap.add_argument("--force-variant", type=int, default=2, choices=[0, 1, 2, 3],
                    help="force kernel of the arena workload (default 2, the exact kernel = the headline).  3 = the "
                         "opt-in streamlined kernel: for profiling it with tools/profile.sh; the line then says "
                         "`headline: false`")
    ap.add_argument("--force-sums", type=int, default=0, choices=[0, 1],
                    help="1: the arena keeps BOTH magnitude sums (pbSimSetForceSums mode 1: everything collideD "
                         "writes, impl.cuh:828-830) -- for profiling that kernel with tools/profile.sh; the line then "
                         "says `headline: false`")
    ap.add_argument("--dry-run-device", action="store_true",
                    help="TEST ONLY (tests/test_bench_multirank.py): no GPU is touched -- gloo process group, the arena "
                         "replaced by a counter, ensemble members placed for real but never stepped; the line says "
                         "dry_run: true and none of its numbers mean anything")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0,
                    help="seconds a rank waits for the others at the rendezvous before bench.py exits with code 2")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="producer threads per pipeline of the ensemble end-to-end runs (default: the rank's share of "
                         "the usable cores, pbHostGetResources)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: exercises the N>1 code path")
    args = ap.parse_args()
    K.DRY = args.dry_run_device
    # HIP gives a new stream the least-used of GPU_MAX_HW_QUEUES (default 4) hardware queues.  Next to the streams of
    # PyTorch and RCCL (any run with a process group) the two batches of the configs[3] leg -- one stream each, meant
    # to overlap -- landed on ONE queue and serialised: 1.68 s end to end instead of 0.95 (round 3, --force-dist).
    # With 6 or more queues they do not; nothing else in this file changes with it (measured).  Must be in the
    # environment before the HIP runtime initialises; the ranks torchrun starts inherit it.  Disclosed in `host`.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.sub_batch is not None and args.workload in FULL_RUN:
        FULL_RUN[args.workload]["sub_batch"] = args.sub_batch
    if args.lanes is not None and args.workload in FULL_RUN:
        FULL_RUN[args.workload]["lanes"] = args.lanes
    if args.members_per_gpu is None:
        args.members_per_gpu = 32 if args.workload == "ensemble4" else 8

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(K.spawn_ranks(args, __file__))
    rank, local_rank, world, dist, torch = K.init_ranks(args)

    import particlerobotsimulations_amd as pb

    if K.DRY:
        pb = K._DryPb
    elif dist is None:
        pb.legacy.cudaInit(0, None)
    else:
        K.engine_device(local_rank)  # (torch.cuda.set_device chose it for torch; the engine is told itself)

    if args.workload != "arena":
        if args.force_variant != 2:
            K.ENSEMBLE_FORCE_VARIANT = args.force_variant
        run_ensemble_workload(args, rank, world, dist, torch)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    K.HEADLINE_VARIANT = args.force_variant
    K.HEADLINE_FORCE_SUMS = args.force_sums
    n = args.bots
    # host work for a later leg starts now, on its own thread, so that no device leg waits for it
    blob = BlobPlacement(n) if (rank == 0 and world == 1 and not args.no_blob) else None
    # one scratch arena for the whole run: stepped for >= prewarm_ms immediately before EVERY timed leg
    warm = DevicePrewarm(pb, min(n, 1_000_000), args.pitch, args.prewarm_ms)
    sim = make_sim(pb, n, args.pitch, seed=1 + rank)
    cfg = sim.config()
    assert cfg["force_variant"] == args.force_variant, cfg  # `value` is the exact kernel unless --force-variant says otherwise

    def barrier():
        sim.synchronize()
        if dist is not None:
            dev_sync(torch)
            dist.barrier()

    prewarm = warm.run()
    sim.step(args.warmup)
    # The timed region: exactly K steps with a barrier + device synchronisation on both sides, MAX over ranks.
    # With a process group the OPENING barrier is entered while the device still works -- the W warm-up steps and a
    # few ms more of the scratch arena, all asynchronous -- and the synchronisation comes after it: an RCCL barrier
    # leaves the device idle for some hundred microseconds otherwise, and the first timed steps then run at idle
    # clocks (measured with --force-dist, --steps 20: 1.80 ms of device time for the 20 steps instead of 1.58).  The
    # clock is read when this rank's K steps have completed, BEFORE the closing barrier (MAX over ranks is the time
    # at which the last rank finished; the barrier's own latency, ~0.2 ms, is not part of any rank's K steps).
    if dist is not None:
        warm.keep_busy(3.0)
        dist.barrier()
        warm.synchronize()
        sim.synchronize()
        dev_sync(torch)
    else:
        barrier()
    s0 = sim.stats()
    t0 = time.perf_counter()
    done, dev_ms = sim.step_timed(args.steps)
    sim.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    s1 = sim.stats()
    assert done == args.steps, (done, args.steps)
    # a timed region under 50 ms of device time (the driver's --steps 20 is ~2 ms) is followed at once by one of
    # >= 100 ms, reported beside `value` as value_long (per rank, no collective inside)
    long_steps, long_ms = 0, 0.0
    if dev_ms < SHORT_MS:
        long_steps, long_ms = sim.step_timed(min(int(LONG_MS / max(dev_ms / done, 1e-6)) + 1, 400000))

    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=dist_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # the only data exchange of an ensemble: per-arena summaries (time, COMx, COMy), gathered
        cx, cy = sim.centroid()
        mine = torch.tensor([sim.time, cx, cy], dtype=torch.float64, device=dist_device())
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        summaries = [[float(x) for x in v.tolist()] for v in allv]
    else:
        cx, cy = sim.centroid()
        summaries = [[sim.time, cx, cy]]
    assert cx == cx and cy == cy, "simulation state went NaN: the benchmark workload is invalid"

    # shader clock under the headline load, while the headline simulation is still alive
    clock_mhz, clock_span = (None, None)
    if rank == 0 and world == 1 and not args.no_clock:
        dev_launches = (s1["fused_launches"] - s0["fused_launches"]) + (s1["plain_launches"] - s0["plain_launches"])
        clock_mhz, clock_span = measure_clock(pb, sim, dev_ms / max(dev_launches, 1))
    sim.close()
    # BASELINE's "1/2/4/8-GPU ensemble": configs[3] (obstacle + object-transport seed ensembles) measured beside the
    # arena at every N, without touching `value`: (a) weak form, 32 + 32 members per GPU: K timesteps in steady state
    # (`value`) and the whole 120 000-step run end to end through the placement/stepping pipeline
    # (`value_end_to_end`); (b) strong form, configs[3] as written: 256 + 256 members in all, end to end.
    # Member k on rank k mod N, RCCL gather of the summary rows.  Collective: every rank runs it.
    # (The arena simulations are closed first: HIP maps streams onto four hardware queues, and with the arena's and the
    #  scratch arena's streams alive the two ensemble batches -- one stream each, meant to overlap -- landed on ONE queue
    #  and serialised: 14.5 us per step instead of 8.2.)
    ens_leg = None
    if not args.no_ensemble_leg:
        warm.done()
        ens_leg, _ = measure_ensemble(pb, "ensemble4", 32, min(max(args.steps, 200), 4000), 20, args.prewarm_ms, rank,
                                      world, dist, torch, end_to_end=not args.no_end_to_end, e2e_steps=args.e2e_steps,
                                      strong_total=256)
        # (rank 0 goes on to the both_sums leg -- `roofline.frac` -- at every world size; the other ranks have no leg left)
        warm = DevicePrewarm(pb, min(n, 1_000_000), args.pitch, args.prewarm_ms if rank == 0 else 0.0)
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
                       "force_variant": cfg["force_variant"], "force_kind": cfg["force_kind"],
                       "lanes_per_bot": cfg["lanes_per_bot"], "resident": cfg["resident"],
                       "attraction_sums": cfg["attraction_sums"], "dead_sum_form": cfg["dead_sum_form"],
                       "force_sums_note": "constrained_contraction = 0 (the reference's default): absForce_a has no "
                                          "reader and is not computed (pbSimSetForceSums mode 0); every array the "
                                          "reference reads or writes out is bit-identical either way",
                       "parallelism": "single arena" if world == 1 else f"{world} independent arenas, one per GPU"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # machine-readable: which kernel `frac` / `achieved` / `avg_launch_us` describe in THIS line
                         # (the same keys meant two kernels depending on the flags)
                         "frac_kernel": "arena_kernel_priced_at_64B",
                         "frac_is": "the kernel `value` runs (dead-sum form) priced at SURVEY 8(d)'s 64 B although it "
                                    "moves 56: the both_sums leg did not run, see frac_dead_sum",
                         "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                         "traffic_source": (f"profiles/latest_traffic.json ({tr['profile']}): {tr['method']}"
                                            if tr else None),
                         # the shipped default at the bytes it is accountable for (no absForce_a: 56 B)
                         "frac_dead_sum": ALG_BYTES_DEAD_SUM * n / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                         "frac_dead_sum_priced_at_64": achieved / HBM_PEAK_GBS,   # what rounds 1-4 called `frac`
                         "dead_sum": {"kernel": "k_force<false, true, 1, 1, false, false> (what `value` runs; "
                                                "profiles/latest_traffic.json)",
                                      "algorithmic_bytes_per_launch": ALG_BYTES_DEAD_SUM * n,
                                      "avg_launch_us": avg_launch_s * 1e6,
                                      "achieved": ALG_BYTES_DEAD_SUM * n / avg_launch_s / 1e9,
                                      "frac": ALG_BYTES_DEAD_SUM * n / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                                      "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                                      "valu_frac_of_datasheet": valu_of_datasheet(tr, n, avg_launch_s * 1e6)},
                         "valu_frac_of_datasheet": valu_of_datasheet(tr, n, avg_launch_s * 1e6),
                         "valu_frac_of_datasheet_is": "VALU instructions of the kernel `value` runs x 64 lanes / launch "
                                                      "time / 78.6e12 lane-instructions/s (fp32 vector peak): the "
                                                      "roofline that binds (`bound`)",
                         "hbm_target_note": hbm_target_note(n),
                         "valu": valu_roofline(tr, n, avg_launch_s * 1e6, clock_mhz),
                         "shader_clock_mhz": clock_mhz,
                         "shader_clock_source": ("s_memtime / s_memrealtime of a sampler wave on its own stream beside "
                                                 f"{clock_span:.2f} s more of the same steps (after the timed region)"
                                                 if clock_mhz else None),
                         "kernel": "k_force, fuse = 1 (forces of step n + radius/integration of step n+1)",
                         "launches": launches, "avg_launch_us": avg_launch_s * 1e6,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_PARTICLE_STEP * n,
                         "note": "achieved/peak/frac are the HBM accounting SURVEY 8(d) prescribes (64 algorithmic "
                                 "bytes per particle-step over the kernel's launch time); the kernel is VALU-issue "
                                 "bound, not HBM bound (~50 neighbour pairs per bot, each with 4 IEEE divisions and 2 "
                                 "IEEE square roots -- one fewer since Sum|F_attr| is only kept when something reads it --, DESIGN.md section 3): `valu` prices its instruction stream at the "
                                 "datasheet issue rate at the measured shader clock and at tools/valu_rate's rates; "
                                 "`traffic` (PMC) ~ algorithmic bytes, i.e. no wasted re-reads"},
            "headline": args.force_variant == 2 and not args.force_sums,
            "device_ms_timed_region": dev_ms,
            "device_prewarm": prewarm,
            "summaries_time_comx_comy": summaries,
            "host": host_info(),
        }
        out["roofline"]["alg_bytes_note"] = (
            "64 B per particle-step = read pos 8 + vel 8 + rad 4 + phase 4 + dead 4 + absForce_a 4 + absForce_r 4, write "
            "pos 8 + vel 8 + rad 4 + absForce_a 4 + absForce_r 4 (SURVEY 8(d)).  `frac` / `achieved` / `avg_launch_us` "
            "price the kernel that writes everything the reference's collideD writes "
            "(particlebot_kernel_impl.cuh:828-830; the both_sums leg, value_with_both_sums) at those 64 B; the form "
            "`value` runs does not touch absForce_a (no reader: constrained_contraction 0) and is priced at the 56 B it "
            "moves: frac_dead_sum / dead_sum")
        if long_steps:
            us_long = long_ms * 1e3 / long_steps
            out["value_long"] = n * long_steps / (long_ms * 1e-3) * world
            out["steps_long"] = long_steps
            out["roofline"]["avg_launch_us_dead_sum_long"] = us_long
            out["roofline"]["dead_sum"].update(avg_launch_us_long=us_long, frac_long=ALG_BYTES_DEAD_SUM * n / (us_long * 1e-6)
                                               / 1e9 / HBM_PEAK_GBS)
            out["roofline"]["frac_dead_sum_long"] = ALG_BYTES_DEAD_SUM * n / (us_long * 1e-6) / 1e9 / HBM_PEAK_GBS
            # (rounds 1-4 priced this kernel at 64 B and called it `frac`: kept for comparison with BENCH_r01 ... r04)
            out["roofline"]["frac_dead_sum_priced_at_64_long"] = (ALG_BYTES_PER_PARTICLE_STEP * n / (us_long * 1e-6) / 1e9
                                                                 / HBM_PEAK_GBS)
            out["value_long_note"] = (f"the {args.steps} timed steps were {dev_ms:.2f} ms of device time: the same "
                                      f"simulation stepped {long_steps} more steps right behind them (device time, "
                                      "rank 0's arena x n_gpus)")
        if ens_leg is not None:
            out["ensemble_leg"] = ens_leg
        if world == 1 and not args.no_large_arena:
            out["large_arena"] = large_arena_leg(pb, args.pitch, 20, min(args.steps, 200), warm=warm)
        if not args.no_both_sums:   # (rank 0 of any world size: `frac` is this leg's kernel)
            out["both_sums"] = both_sums_leg(pb, n, args.pitch, min(args.steps, 400), max(args.warmup, 100), warm=warm)
            # top-level, next to `value`: the same workload with the dead Sum|F_attr| computed all the same
            out["value_with_both_sums"] = out["both_sums"]["value"]
            us_b = out["both_sums"].get("us_per_step_long", out["both_sums"]["us_per_step"])
            steps_b = out["both_sums"].get("steps_long", out["both_sums"]["steps"])
            # LIKE FOR LIKE: `frac` is the kernel that writes everything collideD writes, at the 64 B it
            # moves; measured live (HIP events on the simulation's stream, one launch per step, pre-warmed)
            trb = profiled_traffic("latest_traffic_both_sums.json") if n == 1_000_000 else None
            ach_b = ALG_BYTES_PER_PARTICLE_STEP * n / (us_b * 1e-6) / 1e9
            r = out["roofline"]
            r.update({"achieved": ach_b, "frac": ach_b / HBM_PEAK_GBS, "frac_both_sums": ach_b / HBM_PEAK_GBS,
                      "frac_kernel": "both_sums_leg",
                      "frac_is": "the kernel that writes everything the reference's collideD writes -- both magnitude "
                                 "sums, k_force<false, true, 1, 1, false, true>, the both_sums leg "
                                 "(value_with_both_sums) -- at 64 B per particle-step; NOT the kernel `value` runs: "
                                 "that one is frac_dead_sum (56 B)",
                      "kernel": "k_force<false, true, 1, 1, false, true>, fuse = 1 (forces of step n + "
                                "radius/integration of step n+1; profiles/latest_traffic_both_sums.json)",
                      "launches": steps_b, "avg_launch_us": us_b,
                      "traffic": trb["hbm_bytes_per_launch"] if trb else None,
                      "traffic_source": (f"profiles/latest_traffic_both_sums.json ({trb['profile']}): {trb['method']}"
                                         if trb else None),
                      "valu_frac_of_datasheet_both_sums": valu_of_datasheet(trb, n, us_b)})
            out["config"]["force_sums_note"] += ("; like for like with the reference's collideD (which writes absForce_a "
                                                 "every step) see value_with_both_sums and roofline.frac")
        if world == 1 and not args.no_streamlined:
            out["streamlined"] = streamlined_leg(pb, n, args.pitch, args.steps, args.warmup, warm=warm)
        if world == 1 and not args.no_blob:
            out["random_blob"] = blob_leg(pb, n, min(args.steps, 600), args.warmup, blob, warm=warm)
        if world == 1 and not args.no_survey_literal:
            out["survey_literal_lattice"] = survey_literal(pb, n, args.steps, args.warmup)
        warm.done()
        if world == 1 and not args.no_host_round_trip and not K.DRY:
            out["host_round_trip"] = host_round_trip_leg(pb, n, args.pitch)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.pitch, args.cpu_seconds)
        emit(out)
    else:
        warm.done()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
