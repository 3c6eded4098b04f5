#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the particle-robot update loop on MI355X.

Workload (BASELINE.json configs[2]): 10^6 oscillating bots on a square
lattice (pitch 2*min_radius; see LATTICE_PITCH below) in the generalised arena (2048^2 grid, walls +-240), one light at
(-230, 0), phase_std 0, dt 0.01, sort_interval 180.  A "step" is one timestep of the whole arena:
radius actuation + integration + neighbour forces + friction for every bot (one fused kernel).
State is resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--bots B] [--workload arena|ensemble4|ensemble5]

N > 1 (launched by torch.distributed.run, one rank per GPU; a bare `python bench.py --gpus N` starts
the N ranks itself as a child process): the path does not shard a single arena (neighbour forces
couple every cell each step), so with --workload arena every rank runs its own independent arena --
an ensemble member with its own seed offset -- with no collective in the timed
region; rank 0 gathers the per-arena centroid summaries over RCCL afterwards.  --workload ensemble4 /
ensemble5 run BASELINE configs[3] / configs[4]: batched ensembles of the reference's example
configurations, member k on rank k mod N, the summary rows gathered over RCCL at the end.
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
# of the 64 B, absForce_a is 8 (read 4 + write 4): the shipped default form (no reader: constrained_contraction 0) does
# not touch it and is accountable for 56
ALG_BYTES_DEAD_SUM = 56.0
VALU_PEAK_LANE_OPS = 78.6e12        # fp32 vector peak in lane-instructions/s (157.3 TFLOP/s of FMA)
PAIRS_PER_BOT_LATTICE = 57          # candidate pairs per bot on the bench lattice (tests/model_divergence.py)


# ---- --dry-run-device (test-only) --------------------------------------------------------------------------------
# The multi-rank choreography of this file -- spawn_ranks, the rendezvous, barriers, the MAX all-reduce, the gather of
# summary rows, ONE JSON line from rank 0 -- must be exercised before the driver's SCALE run does it, and this
# container has no GPU.  With --dry-run-device the process group is gloo on CPU tensors and the device is replaced by
# the stand-ins below: the arena by a counter (_DrySim), an ensemble's stepping by the pipeline's dry-run consumer
# (pbEnsemblePipelineDryRun: members are PLACED for real, by the producer pool, and their rows are made from the
# checksum of the placed state).  Every number in a dry-run line is meaningless and the line says `dry_run: true`;
# tests/test_bench_multirank.py reads its structure.
DRY = False


def dev_sync(torch):
    if not DRY:
        torch.cuda.synchronize()


def dist_device():
    return "cpu" if DRY else "cuda"


class _DrySim:
    def __init__(self, params, wall_half=0.0, keepalive=None):
        self.n, self.time, self._steps, self._variant = int(params.nCells), 0.0, 0, 2

    def set_force_variant(self, v):
        self._variant = v

    def set_lanes_per_bot(self, lanes):
        pass

    set_resident = set_force_sums = set_lanes_per_bot

    def set_state(self, **kw):
        pass

    def step(self, k, *a):
        self._steps += k
        self.time += 0.01 * k
        return k

    def step_timed(self, k, *a):
        return self.step(k), 0.08 * k

    def synchronize(self):
        pass

    close = synchronize

    def centroid(self):
        return 0.0, 0.0

    def stats(self):
        return {"steps": self._steps, "fused_launches": self._steps, "plain_launches": 0, "state_launches": 0, "resorts": 0,
                "phase_updates": 0, "resident_launches": 0}

    def config(self):
        return {"force_variant": self._variant, "force_kind": 0, "lanes_per_bot": 1, "resident": 0, "attraction_sums": 0,
                "dead_sum_form": 1}


class _DryPb:
    Sim = _DrySim


class _DryLocalEnsemble:
    """ensemble.LocalEnsemble without a device: members placed by the pipeline, rows from their checksums"""

    def __init__(self, cfg, over, common):
        from particlerobotsimulations_amd import ensemble
        self._p = ensemble.PipelinedEnsemble(cfg, over, common)
        self.m = len(over)
        self._p.run_dry(0)
        self.n = getattr(self._p, "n", 0)

    def run_steps(self, k):
        return k

    def synchronize(self):
        pass

    @property
    def rows(self):
        return self._p.rows

    def close(self):
        self._p.close()


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


ENSEMBLE_FORCE_VARIANT = None   # --workload ensemble4|5 --force-variant V: the members' pb_force_variant key
HEADLINE_VARIANT = 2   # the exact kernel; --force-variant 3 (profiling the streamlined kernel) is flagged in the line
HEADLINE_FORCE_SUMS = 0  # --force-sums 1: the arena itself keeps both magnitude sums (profiling that kernel; `headline: false`)


def make_sim(pb, n, pitch, seed, lattice="square"):
    import numpy as np
    sp, keep = workload_params(n, seed=seed)
    sim = pb.Sim(sp, wall_half=240.0, keepalive=keep)
    sim.set_force_variant(HEADLINE_VARIANT)   # the exact kernel, whatever the environment says (legs that want 3 set it)
    if HEADLINE_FORCE_SUMS:
        sim.set_force_sums(1)
    sim.set_lanes_per_bot(0)
    sim.set_resident(0)
    pos = square_lattice(n, pitch) if lattice == "square" else hex_lattice(n, np.float32(pitch))
    sim.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                  phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
    return sim


LONG_MS = 100.0    # a timed region shorter than SHORT_MS of device time is followed by a second one of >= LONG_MS
SHORT_MS = 50.0


def timed_leg(sim, warm, warmup, steps):
    """The measurement protocol of every arena leg: >= 100 ms of the same kind of work on the
    scratch arena IMMEDIATELY before (clock ramp), W untimed warm-up steps, EXACTLY K timed steps between HIP
    events on the simulation's stream -- nothing else in between, no host copy, no allocation -- and, when those K
    steps were less than 50 ms of device time (the driver's --steps 20 is ~2 ms), a second region of >= 100 ms
    right behind it, reported as *_long."""
    prewarm = warm.run() if warm is not None else None
    sim.step(warmup)
    done, ms = sim.step_timed(steps)
    out = {"steps": done, "ms": ms, "us_per_step": ms * 1e3 / max(done, 1), "device_prewarm_ms": prewarm["ms"] if prewarm else 0.0}
    if ms < SHORT_MS and done > 0:
        k = min(int(LONG_MS / max(ms / done, 1e-6)) + 1, 400000)
        d2, ms2 = sim.step_timed(k)
        out.update(steps_long=d2, ms_long=ms2, us_per_step_long=ms2 * 1e3 / max(d2, 1))
    return out


def leg_fields(t, n):
    """value / us_per_step (+ *_long) of a timed_leg result for an n-bot arena."""
    f = {"value": n * t["steps"] / (t["ms"] * 1e-3), "unit": "particle-steps/s (device time)", "steps": t["steps"],
         "us_per_step": t["us_per_step"], "device_prewarm_ms": t["device_prewarm_ms"]}
    if "ms_long" in t:
        f.update(value_long=n * t["steps_long"] / (t["ms_long"] * 1e-3), steps_long=t["steps_long"],
                 us_per_step_long=t["us_per_step_long"])
    return f


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
    back for dumpParticlebot).  DESIGN.md section 6 quotes this leg."""
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


class DevicePrewarm:
    """The chip ramps its clocks over the first ~100 ms of load and drops them again when idle
    (measured: the first 20 steps after an idle spell run at 137 us, after 50 ms of the same kind of
    work at 115 us -- MI355X_MICROARCH.md "DVFS give-back" asks for seconds of back-to-back launches
    before quoting a kernel).  A short timed region (the driver's --steps 20 --warmup 5) would otherwise
    measure the ramp, not the kernel.  So a SCRATCH copy of the workload (its own simulation object,
    thrown away) is created up front and stepped for at least min_ms of device time immediately before
    the measured simulation's own W warm-up steps and exactly K timed ones."""

    def __init__(self, pb, n, pitch, min_ms):
        self.min_ms = min_ms
        self.scratch = make_sim(pb, n, pitch, seed=12345) if min_ms > 0 else None
        if self.scratch is not None:
            self.scratch.set_force_variant(2)
        self.info = {"ms": 0.0, "steps": 0}
        self.runs = 0

    def run(self):
        if self.scratch is None:
            return self.info
        self.runs += 1
        steps, ms = 0, 0.0
        while ms < self.min_ms and steps < 20000:
            d, m = self.scratch.step_timed(100)
            steps += d
            ms += m
        # (closed later, by done(): freeing its buffers here would leave the device idle for milliseconds)
        self.info = {"ms": ms, "steps": steps,
                     "what": "a scratch copy of the workload stepped right before the measured simulation's warm-up "
                             "steps, to bring the device out of its idle power state; not part of warmup/steps"}
        return self.info

    def keep_busy(self, ms):
        """Enqueue ~ms of scratch steps WITHOUT waiting for them (the device stays loaded while the host is elsewhere)."""
        if self.scratch is not None and self.info["steps"] > 0:
            per_step = self.info["ms"] / self.info["steps"]
            self.scratch.step(max(1, min(int(ms / max(per_step, 1e-3)), 2000)))

    def synchronize(self):
        if self.scratch is not None:
            self.scratch.synchronize()

    def done(self):
        if self.scratch is not None:
            self.scratch.close()
            self.scratch = None


def profiled_traffic(which="latest_traffic.json"):
    """HBM bytes per k_force launch and its VALU instruction counts from the committed rocprofv3 PMC
    passes of this same command (profiles/latest_traffic.json: the shipped dead-sum form;
    profiles/latest_traffic_both_sums.json: the form that keeps both magnitude sums; both written by
    tools/profile.sh); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", which)) as fh:
            return json.load(fh)
    except Exception:
        return None


def valu_of_datasheet(tr, n, avg_launch_us):
    """The kernel's VALU instruction stream as a fraction of the chip's fp32 vector peak, 78.6e12 lane-instructions
    per second (every instruction counted once, 64 lanes each, whatever its issue cost): the roofline that BINDS."""
    if not tr or "valu_insts_per_wave" not in tr:
        return None
    return n * tr["valu_insts_per_wave"] / (avg_launch_us * 1e-6) / VALU_PEAK_LANE_OPS


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


# ---- ensemble workloads (BASELINE configs[3] and configs[4]) -------------------------------------
ENSEMBLE_WORKLOADS = {
    # name -> list of batches: (cfg, common overrides, per-member override maker)
    "ensemble4": "examples/example_obstacle.cfg + examples/example_object_transport.cfg, Monte-Carlo seeds "
                 "1000+k (BASELINE configs[3]); per GPU one batched pbSim per .cfg, both driven concurrently",
    "ensemble5": "examples/example_dead_cells.cfg at nCells 100000, light (-40,0), dead fraction swept 0..0.40 "
                 "over the members (BASELINE configs[4]); per GPU one batched pbSim",
}


FULL_RUN = {   # the BASELINE configuration at full length: max_time, timesteps per member, sub-batch
    "ensemble4": {"max_time": "1200", "steps": 120000, "sub_batch": 0},    # 100 actuation cycles; members of 500 / 201 bots
    # 10 cycles; 10^5-bot members, placement 0.85-1.6 s each.  -1: one placement round of the producer pool per sub-batch
    # (31 members with PB_HOST_THREADS=32).  Round 3 measured, whole config on one GPU: 8 members per sub-batch 104.1 s
    # (66 us per step of 8 x 10^5 bots carries the full ramp and drain), 32 members 95.4 s (first sub-batch ready
    # after TWO placement rounds: 3.0 s of waiting), 64 members 28.0 s per quarter against 26.0 s with 32.
    "ensemble5": {"max_time": "120", "steps": 12000, "sub_batch": -1},
}


def ensemble_batches(workload, rank, world, members_per_gpu, members_total=None, max_time="1e9"):
    """[(cfg_path, common, [override text per local member], [global member ids])] for this rank.
    Global member k -> rank k mod world (ensemble.shard).  Weak form: N GPUs run N x members_per_gpu members per
    .cfg; strong form (members_total): a FIXED number of members per .cfg, whatever N is."""
    from particlerobotsimulations_amd import ensemble
    total = members_total if members_total is not None else members_per_gpu * world
    ids = ensemble.shard(total, rank, world)
    ex = lambda name: os.path.join(ROOT, "examples", name)
    big = {"max_time": max_time, "dump_interval": "6"}
    if ENSEMBLE_FORCE_VARIANT is not None:
        big["pb_force_variant"] = str(ENSEMBLE_FORCE_VARIANT)   # (--force-variant with an ensemble workload)
    if workload == "ensemble4":
        return [(ex("example_obstacle.cfg"), big, [f"seed\n{1000 + k}" for k in ids], ids),
                (ex("example_object_transport.cfg"), big, [f"seed\n{1000 + k}" for k in ids], ids)]
    common = dict(big, nCells="100000", light_x="-40", light_y="0")
    over = []
    for k in ids:
        f = 0.40 * (k % 64) / 63.0
        over.append(f"seed\n{1000 + k // 64}\nnDead\n{int(round(f * 100000))}")
    return [(ex("example_dead_cells.cfg"), common, over, ids)]


def pipeline_bound(tm, members):
    """host-bound or device-bound?  What the host needs for this rank's members with the producer threads it has
    (placement CPU-seconds / threads) against what the device needs (upload + stepping + read-backs)."""
    host_s = tm["placement_cpu_s"] / max(tm["host_threads"], 1)
    dev_s = tm["device_s"] + tm["upload_s"]
    return {"bound": "host" if host_s > dev_s else "device", "host_s": host_s, "device_s": dev_s,
            "placement_cpu_s_per_member": tm["placement_cpu_s"] / max(members, 1), "members": members,
            "producer_threads": tm["host_threads"], "device_waited_for_host_s": tm["placement_wait_s"],
            "producers_pinned_to_gpu_numa_node": bool(tm.get("pinned")), "numa_node": tm.get("numa_node", -1),
            "oversubscription": (tm["placement_thread_wall_s"] / tm["placement_cpu_s"]
                                 if tm.get("placement_cpu_s", 0) > 0 else None),
            "note": "host_s = placement CPU-seconds of this rank's members / its producer threads; bound = host when "
                    "that exceeds the device's time for them (the wall time is then placement, not stepping); "
                    "oversubscription = producers' wall time / CPU time (1 = every producer had a core)"}


def ensemble_end_to_end(workload, rank, world, dist, torch, members_per_gpu=None, members_total=None, max_steps=None,
                        host_threads=0, extra_common=None):
    """One ensemble run END TO END, as a user of bin/particlebot_ensemble experiences it: from the moment the
    members exist only as override strings to the moment rank 0 holds every member's summary rows -- host placement
    (overlapped with device stepping by the sub-batch pipeline, pbEnsemblePipeline*), state upload, every timestep
    of the configuration at FULL LENGTH (FULL_RUN; max_steps bounds it for quick tests), the dead-bot draws, the
    summary reductions and the one RCCL gather.  Wall clock between two barriers, max over ranks.
    Collective: every rank calls it.  Returns the result on rank 0, None elsewhere."""
    import threading

    import numpy as np
    from particlerobotsimulations_amd import ensemble
    full = FULL_RUN[workload]
    batches = ensemble_batches(workload, rank, world, members_per_gpu, members_total, max_time=full["max_time"])
    steps_cap = full["steps"] + 1 if max_steps is None else int(max_steps)

    def barrier():
        if dist is not None:
            dev_sync(torch)
            dist.barrier()
    barrier()
    t0 = time.perf_counter()
    res = ensemble.host_resources()
    if host_threads <= 0 and len(batches) > 1:
        # this rank's pipelines place at the same time: they share the rank's producer threads (one core stays
        # with the threads that drive the device) instead of each taking all of them
        host_threads = max(1, (res["host_threads"] - 1) // len(batches))
    common_extra = dict(extra_common or {})
    pipes = [ensemble.PipelinedEnsemble(cfg, over, dict(common, **common_extra), sub_batch=full["sub_batch"],
                                        host_threads=host_threads, lanes=full.get("lanes"))
             for cfg, common, over, _ in batches]   # placement starts here, on the producer threads
    done = [0] * len(pipes)
    errors = [None] * len(pipes)

    def one(i):
        try:
            done[i] = pipes[i].run_dry(steps_cap) if DRY else pipes[i].run(steps_cap)
        except BaseException as e:   # re-raised on the main thread below: a leg with a failed pipeline has no value
            errors[i] = e
    th = [threading.Thread(target=one, args=(i,)) for i in range(1, len(pipes))]
    for t in th:
        t.start()
    one(0)
    for t in th:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    assert all(d == done[0] and d > 0 for d in done), done
    total_members = members_total if members_total is not None else members_per_gpu * world
    gathered = [ensemble.gather_summaries(p.rows, total_members, rank, world, dist,
                                          dist_device() if dist is not None else "cpu") for p in pipes]
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=dist_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    timings = [p.timings for p in pipes]
    bots = [getattr(p, "n", 0) for p in pipes]
    for p in pipes:
        p.close()
    if rank != 0:
        return None
    steps = done[0]
    work = sum(b * total_members * d for b, d in zip(bots, done))
    assert all(np.isfinite(g[:, -1]).all() for g in gathered), "an ensemble member went NaN"
    return {"value_end_to_end": work / wall, "unit": "particle-steps/s (wall: placement + upload + steps + gather)",
            "wall_s": wall, "steps_per_member": steps, "members_total": total_members * len(pipes),
            "bots_per_member": bots, "sims_per_s_end_to_end": total_members * len(pipes) / wall,
            "scaling": "strong" if members_total is not None else "weak", "n_gpus": world,
            "rows_gathered": [list(g.shape) for g in gathered],
            "last_rows_time_comx_comy_dist": [[[float(x) for x in r] for r in g[:4, -1]] for g in gathered],
            "pipeline_rank0": timings,
            "bound_rank0": [pipeline_bound(tm, p.m) for tm, p in zip(timings, pipes)] if not DRY else None,
            "placement": (extra_common or {}).get("pb_placement", "reference rule (CONFIG_RANDOM, particlebot.cpp:612-748)"),
            "host_share_rank0": [tm["placement_wait_s"] / max(tm["wall_s"], 1e-9) for tm in timings],
            "note": "placement_wait_s is the time the device-driving thread waited for the host (the unhidden part of "
                    "placement); placement_cpu_s is what the host spent in all; FULL configuration length unless "
                    "steps_per_member says otherwise"}


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
    ens = [(_DryLocalEnsemble if DRY else ensemble.LocalEnsemble)(cfg, over, common) for cfg, common, over, _ in batches]
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
        # 10^5-bot member) the same run is repeated with the O(N) generator (pb_placement fastblob), so
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
                             "(host-driven schedule included); small members are latency-bound (DESIGN.md 6b)"},
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
    if end_to_end and workload == "ensemble5" and e2e is not None and not DRY:
        out["end_to_end_bound"] = "host" if any(b["bound"] == "host" for b in e2e["bound_rank0"]) else "device"
        if fast is not None:
            out["end_to_end_fastblob"] = fast
    return out, batches


def run_ensemble_workload(args, rank, world, dist, torch):
    import particlerobotsimulations_amd as pb
    if DRY:
        pb = _DryPb
    res, batches = measure_ensemble(pb, args.workload, args.members_per_gpu, args.steps, args.warmup, args.prewarm_ms,
                                    rank, world, dist, torch, members_total=args.members_total,
                                    end_to_end=not args.no_end_to_end, e2e_steps=args.e2e_steps,
                                    host_threads=args.host_threads)
    if rank == 0:
        out = {"metric": "particle-steps/sec at 10^6 bots; achieved HBM GB/s vs peak; 1/2/4/8-GPU ensemble",
               "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "host": host_info()}
        out.update(res)
        if ENSEMBLE_FORCE_VARIANT is not None:
            out["headline"] = False
            out["config"]["force_variant"] = ENSEMBLE_FORCE_VARIANT
            out["force_variant_note"] = ("pb_force_variant set for every member: 3 = the opt-in tolerance kernel for "
                                         "batches in the throughput form (not bit-identical; DESIGN.md section 8)")
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


def host_info():
    """What the host-side arithmetic of the path runs on: the libm whose powf the phase update's minimum rests on
    (tests/test_libm_pin.py checks its properties exhaustively) and the cores placement can use."""
    import ctypes as C
    from particlerobotsimulations_amd import host
    L = host.lib()
    L.pbHostLibcVersion.restype = C.c_char_p
    try:
        cpus = len(os.sched_getaffinity(0))
    except Exception:
        cpus = os.cpu_count()
    from particlerobotsimulations_amd import ensemble
    res = ensemble.host_resources()
    return {"glibc": L.pbHostLibcVersion().decode(), "cpus": cpus,
            "usable_cpus": res["usable_cpus"], "cgroup_cpu_quota": res["cgroup_cpus"] if res["cgroup_cpus"] > 0 else None,
            "ranks_per_node": res["local_world_size"], "host_threads": res["host_threads"],
            "gpu_numa_node": res["numa_node"], "gpu_numa_cpus": res["numa_cpus"], "pin_producers": bool(res["pin_producers"]),
            "host_threads_rule": res["rule"],
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}


_JSON_FD = None   # with a process group: the original stdout (fd 1 itself is pointed at stderr, see divert_stdout)


def divert_stdout():
    """RCCL prints a five-line version banner to stdout when its communicator is created (this build does so with
    NCCL_DEBUG unset; NCCL_DEBUG_FILE does not move it).  stdout carries ONE line, the JSON: everything else written to
    file descriptor 1 from here on -- by C libraries or by Python -- goes to stderr."""
    global _JSON_FD
    if _JSON_FD is None:
        try:
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            _JSON_FD = saved
        except OSError:
            _JSON_FD = None   # (no usable stderr: keep stdout as it is; the JSON line is still the last one)


def emit(out):
    if DRY:
        out["dry_run"] = True
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if _JSON_FD is not None:
        os.write(_JSON_FD, (json.dumps(out) + "\n").encode())
    else:
        print(json.dumps(out), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- as a CHILD process,
    before this one has made any HIP or torch.cuda call -- and pass its exit code on."""
    import subprocess
    import torch
    have = args.gpus if args.dry_run_device else torch.cuda.device_count()  # (does not initialise the GPU)
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible\n")
        return 2
    port = str(29500 + (os.getpid() % 400))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


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
    global DRY
    DRY = args.dry_run_device
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
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)
    dist = None
    torch = None
    if world > 1 or args.force_dist:
        # torch first: its bundled HIP runtime must be the one instance in the process
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        divert_stdout()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        import torch
        import torch.distributed as dist
        # The rendezvous has its own, short deadline: a rank that never arrives (died at start-up, wrong WORLD_SIZE)
        # must end the run with exit code 2 after --rendezvous-timeout seconds, not hang it for the process group's
        # collective timeout.  (Port MASTER_PORT + 1: the launcher's own store may sit on MASTER_PORT.)
        try:
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1, world, rank == 0,
                                  timeout=datetime.timedelta(seconds=args.rendezvous_timeout), wait_for_workers=True)
            store.set(f"rank{rank}", "here")
            store.wait([f"rank{r}" for r in range(world)], datetime.timedelta(seconds=args.rendezvous_timeout))
            if DRY:
                dist.init_process_group(backend="gloo", store=store, rank=rank, world_size=world)
            else:
                torch.cuda.set_device(local_rank)
                dist.init_process_group(backend="nccl", store=store, rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
            assert dist.get_world_size() == world
            dist.barrier()   # (the communicator and RCCL's kernels are set up here, not inside a measurement's first barrier)
        except Exception as e:
            sys.stderr.write(f"bench.py: rank {rank} of {world}: rendezvous failed ({type(e).__name__}: {e})\n")
            sys.stderr.flush()
            os._exit(2)

    import particlerobotsimulations_amd as pb

    if DRY:
        pb = _DryPb
    elif dist is None:
        pb.legacy.cudaInit(0, None)  # otherwise torch.cuda.set_device above already chose this rank's GPU

    if args.workload != "arena":
        global ENSEMBLE_FORCE_VARIANT
        if args.force_variant != 2:
            ENSEMBLE_FORCE_VARIANT = args.force_variant
        run_ensemble_workload(args, rank, world, dist, torch)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    global HEADLINE_VARIANT, HEADLINE_FORCE_SUMS
    HEADLINE_VARIANT = args.force_variant
    HEADLINE_FORCE_SUMS = args.force_sums
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
                                 "IEEE square roots -- one fewer since Sum|F_attr| is only kept when something reads it --, DESIGN.md section 5): `valu` prices its instruction stream at the "
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
        if world == 1 and not args.no_host_round_trip and not DRY:
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
