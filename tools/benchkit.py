#!/usr/bin/env python3
"""benchkit -- what bench.py (the headline line) and tools/bench_legs.py (every other leg) share: the synthetic
phototaxis arena of BASELINE configs[2], the timing protocol (pre-warm, W warm-up steps, EXACTLY K timed steps between
HIP events on the simulation's stream), the CPU baseline, the configs[3]/[4] ensemble runs, the process-group set-up
and the one-JSON-line emitter."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

METRIC = "particle-steps/sec at 10^6 bots; achieved HBM GB/s vs peak; 1/2/4/8-GPU ensemble"
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
        self.n, self.time, self._steps, self._variant, self._sums = int(params.nCells), 0.0, 0, 2, 0

    def set_force_variant(self, v):
        self._variant = v

    def set_lanes_per_bot(self, lanes):
        pass

    set_resident = set_lanes_per_bot

    def set_force_sums(self, mode):
        self._sums = int(mode)

    def force_kernel_name(self):
        return f"k_force<dry run, sums {self._sums}>()"

    def set_state(self, **kw):
        pass

    def step(self, k, *a):
        self._steps += k
        self.time += 0.01 * k
        return k

    def step_timed(self, k, *a):
        return self.step(k), 0.08 * k

    def step_timed_wall(self, k, *a):
        return self.step(k), 0.08 * k, 0.0805 * k

    def synchronize(self):
        pass

    close = synchronize

    def centroid(self):
        return 0.0, 0.0

    def stats(self):
        return {"steps": self._steps, "fused_launches": self._steps, "plain_launches": 0, "state_launches": 0, "resorts": 0,
                "phase_updates": 0, "resident_launches": 0}

    def config(self):
        return {"force_variant": self._variant, "force_kind": 0, "lanes_per_bot": 1, "resident": 0,
                "attraction_sums": self._sums, "dead_sum_form": 1 - self._sums}


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


def make_sim(pb, n, pitch, seed, lattice="square", force_sums=None):
    import numpy as np
    sp, keep = workload_params(n, seed=seed)
    sim = pb.Sim(sp, wall_half=240.0, keepalive=keep)
    sim.set_force_variant(HEADLINE_VARIANT)   # the exact kernel, whatever the environment says (legs that want 3 set it)
    if HEADLINE_FORCE_SUMS if force_sums is None else force_sums:
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


def loaded_build_stamp():
    """lib/build_stamp.json of the libraries this process loads (csrc/Makefile writes it); None if absent."""
    try:
        with open(os.path.join(ROOT, "particlerobotsimulations_amd", "lib", "build_stamp.json")) as fh:
            return json.load(fh)
    except Exception:
        return None


def matched_profile(which, loaded_signature, stamp=None):
    """(counters, reason): the committed PMC profile `which` if it is a profile of the kernel THIS library launches --
    same rocprofv3 signature (template arguments and argument types, pbSimForceKernelName) -- else (None, why).  A
    profile of another signature is another kernel's counters: roofline.traffic / valu are dropped rather than quoted.
    The counters carry `sources_match`: whether the profile was taken on a build of the same force-kernel sources
    (lib/build_stamp.json) as the loaded one."""
    tr = profiled_traffic(which)
    if tr is None:
        return None, f"profiles/{which} absent"
    have = tr.get("kernel_signature")
    if not have:
        return None, f"profiles/{which} records no kernel signature (written before round 6)"
    if " ".join(have.split()) != " ".join((loaded_signature or "").split()):
        return None, f"profiles/{which} is a profile of `{have[:60]}...`, not of the loaded library's kernel"
    stamp = loaded_build_stamp() if stamp is None else stamp
    built = (tr.get("build") or {}).get("kernel_sources_sha16")
    tr = dict(tr, sources_match=bool(stamp and built and built == stamp.get("kernel_sources_sha16")))
    return tr, None


def valu_of_datasheet(tr, n, avg_launch_us):
    """The kernel's VALU instruction stream as a fraction of the chip's fp32 vector peak, 78.6e12 lane-instructions
    per second (every instruction counted once, 64 lanes each, whatever its issue cost): the roofline that BINDS."""
    if not tr or "valu_insts_per_wave" not in tr:
        return None
    return n * tr["valu_insts_per_wave"] / (avg_launch_us * 1e-6) / VALU_PEAK_LANE_OPS

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
            "placements_run": tm.get("placements_run"), "members_that_took_a_shared_placement": tm.get("placements_shared"),
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


def write_all(fd, data):
    """os.write until every byte is out (a pipe may take a line in pieces)"""
    view = memoryview(data)
    while len(view):
        view = view[os.write(fd, view):]


def emit(out, detail=None, detail_path=None):
    """ONE JSON line on stdout (`out`); `detail` (everything else a leg measured) goes to detail_path and to stderr."""
    if DRY:
        out["dry_run"] = True
    if detail is not None and detail_path:
        text = json.dumps(dict(detail, line=out), indent=1)
        try:
            with open(detail_path, "w") as fh:
                fh.write(text + "\n")
        except OSError as e:
            sys.stderr.write(f"bench: cannot write {detail_path}: {e}\n")
        sys.stderr.write(text + "\n")
        sys.stderr.flush()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    write_all(_JSON_FD if _JSON_FD is not None else 1, (json.dumps(out) + "\n").encode())


def spawn_ranks(args, script):
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
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(script)] + sys.argv[1:]
    return subprocess.call(cmd)


def init_ranks(args):
    """(rank, local_rank, world, dist, torch): RANK / LOCAL_RANK / WORLD_SIZE from the launcher; with more than one rank
    (or --force-dist) a process group -- RCCL on the GPUs, gloo under --dry-run-device.  Exits with code 2 when the
    launcher's WORLD_SIZE contradicts --gpus or a rank never arrives at the rendezvous."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)
    if world == 1 and not args.force_dist:
        return rank, local_rank, world, None, None
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
            # LOCAL_RANK is the device index unless the launcher already narrowed each rank's view to its own GPU
            # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank): then every rank sees one device, index 0
            global DEVICE_INDEX
            ndev = torch.cuda.device_count()
            DEVICE_INDEX = local_rank if local_rank < ndev else local_rank % max(ndev, 1)
            torch.cuda.set_device(DEVICE_INDEX)
            dist.init_process_group(backend="nccl", store=store, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", DEVICE_INDEX))
        assert dist.get_world_size() == world
        dist.barrier()   # (the communicator and RCCL's kernels are set up here, not inside a measurement's first barrier)
    except Exception as e:
        sys.stderr.write(f"bench.py: rank {rank} of {world}: rendezvous failed ({type(e).__name__}: {e})\n")
        sys.stderr.flush()
        os._exit(2)
    return rank, local_rank, world, dist, torch


DEVICE_INDEX = None   # init_ranks: the device index of this rank as the process sees it


def engine_device(local_rank):
    """Point the ENGINE (libparticlebot_hip.so) at this rank's GPU through its own C-ABI and return the device it then
    reports.  torch.cuda.set_device chose the device for torch's view of the HIP runtime; if the two libraries ever
    resolved to two copies of the runtime, every rank's simulations would silently land on device 0 -- so the engine is
    told explicitly, and what it answers is what `collective.local_rank_device` reports."""
    if DRY:
        return -1
    import ctypes as C
    from particlerobotsimulations_amd import _capi
    want = int(local_rank) if DEVICE_INDEX is None else int(DEVICE_INDEX)
    _capi.check(_capi.lib().pbSetDevice(want), "pbSetDevice")
    dev = C.c_int(-1)
    _capi.check(_capi.lib().pbGetDevice(C.byref(dev)), "pbGetDevice")
    if dev.value != want:
        raise RuntimeError(f"the engine sits on device {dev.value}, this rank's device is {want} (LOCAL_RANK {local_rank})")
    return dev.value


def collective_info(dist, torch, local_rank, dev=None):
    """What the process group itself reports -- backend, rank count, and which device every rank's ENGINE sits on
    (engine_device; index and PCI bus id, gathered over that same group) -- so that "RCCL saw N ranks on N distinct
    GPUs" is checkable from the line, also when a launcher narrowed every rank's view to one device (index 0 everywhere)."""
    if dist is None:
        return None
    if dev is None:
        dev = -1 if DRY else int(torch.cuda.current_device())
    bus = b""
    if not DRY:
        import ctypes as C
        from particlerobotsimulations_amd import _capi
        buf = C.create_string_buffer(32)
        if _capi.lib().pbDevicePciBusId(int(dev), buf, len(buf)) == 0:
            bus = buf.value[:16]
    mine = torch.tensor([local_rank, dev] + list(bus.ljust(16, b"\0")), dtype=torch.int64, device=dist_device())
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    rows = [[int(x) for x in v.tolist()] for v in allv]
    buses = [bytes(r[2:]).rstrip(b"\0").decode(errors="replace") for r in rows]
    out = {"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "torch_backend": dist.get_backend(),
           "ranks": dist.get_world_size(), "local_rank_device": [r[:2] for r in rows]}
    if any(buses):
        out["pci_bus_ids"] = buses
        out["distinct_gpus"] = len(set(buses))
    return out
