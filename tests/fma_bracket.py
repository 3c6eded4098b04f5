"""The FMA / __powf BRACKET: how far does a legitimately different build of the reference's own
arithmetic drift from the oracle in a teacher-forced window?

Why this exists.  BASELINE.json's north star asks that results "match the reference CUDA path's
centre-of-mass trajectory and per-particle positions ... to within 1e-5 relative fp32 tolerance".
The reference is built by nvcc -O3 with the default -fmad=true (/root/reference/Makefile:79-89 sets
no -fmad=false) and evaluates two powers with the fast intrinsic __powf
(particlebot_kernel_impl.cuh:586,589).  Which products nvcc fuses and what bits __powf returns cannot
be known in this image (no nvcc, no GPU of that make), so "the reference CUDA path" is a FAMILY of
results.  This module measures two members of the family against the oracle (which is the
contraction-free, x*x member): oracle/libpb_oracle_fma.so (kernel functions contracted) and
oracle/libpb_oracle_fma_powf.so (contracted + exp2f(2*log2f(x)) at the __powf sites); see
oracle/Makefile and the header of oracle/pb_oracle.c.  The same windows are then used to measure the
streamlined GPU kernel (force variant 3, tests/test_gpu_fma_bracket.py), which is the only kernel
of the product that is not bit-identical to the oracle.

Everything here is test infrastructure (it imports the oracle).  Procedure, per case:
  * the exact oracle (the "teacher") runs from t = 0;
  * at each epoch E the candidate is given the teacher's complete state (positions, velocities,
    radii, phases, dead flags, both force sums, the stale sort order, the fp32 clock);
  * teacher and candidate step 10 times: per-particle relative deviation |p - p_ref| / |p_ref|
    (median, p99, max), number of bots beyond 1e-5 ("flips"), relative deviation of the centre of
    mass, largest absolute displacement;
  * both keep stepping WITHOUT re-synchronisation up to HORIZON steps and the first step at which
    the max / p99 / centre-of-mass deviation exceeds 1e-5 is recorded (the horizon at which the
    tolerance breaks; the dynamics are chaotic).
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = lambda name: os.path.join(ROOT, "examples", name)

RTOL = 1e-5     # BASELINE.json north_star
WINDOW = 10     # teacher-forced windows of <= 10 steps
HORIZON = 100   # how far the un-resynchronised continuation is followed
SYNC_KEYS = ("pos", "vel", "rad", "phase", "dead", "absForce_a", "absForce_r", "hash", "index")

EPOCHS_FULL = (50, 200, 400, 1195, 1400, 3000)  # 1195: the window crosses the phase update at step 1200
EPOCHS_SHORT = (50, 400, 1195)
EPOCHS_RIM = (0, 110, 400, 1195, 1400, 3000)   # 0: the placed blob's first ten steps against the moved obstacles


def _cfg(orc, name, **over):
    over.setdefault("phase_std", 0.0)  # cuRAND noise is unpinned; strict runs switch it off
    over.setdefault("max_time", 1e9)
    return orc.load_cfg(EX(name), **over), None


def _arena_crop(orc, n):
    """BASELINE configs[2] (bench.py's headline workload: square lattice at pitch 0.155 under a far
    light) cropped to n bots; grid and walls as the reference's defaults, which hold 10^4 bots."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    P = orc.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0, light_y=0.0)
    return P, {"pos": bench.square_lattice(n, bench.LATTICE_PITCH)}


# name -> (builder(orc) -> (OrcParams, initial arrays or None = the reference's placement), epochs,
#          BASELINE.json config it stands for)
CASES = {
    "cfg1_example_300": (lambda orc: _cfg(orc, "example.cfg"), EPOCHS_FULL, "configs[0]: examples/example.cfg verbatim"),
    "cfg2a_dead_cells_100": (lambda orc: _cfg(orc, "example_dead_cells.cfg"), EPOCHS_FULL,
                             "configs[1] as shipped: examples/example_dead_cells.cfg (100 bots, 20 dead)"),
    "cfg2b_dead_cells_10k": (lambda orc: _cfg(orc, "example_dead_cells.cfg", nCells=10000, nDead=2000), EPOCHS_FULL,
                             "configs[1] as BASELINE.json words it: the same file at 10^4 bots, 20 % dead"),
    "cfg3_arena_crop_10k": (lambda orc: _arena_crop(orc, 10000), EPOCHS_FULL,
                            "configs[2]: the 10^6-bot bench lattice, a 10^4-bot crop"),
    "cfg4_obstacle_500": (lambda orc: _cfg(orc, "example_obstacle.cfg", seed=1000), EPOCHS_FULL,
                          "configs[3]: examples/example_obstacle.cfg, ensemble member seed 1000"),
    "cfg4_object_transport_201": (lambda orc: _cfg(orc, "example_object_transport.cfg", seed=1000), EPOCHS_FULL,
                                  "configs[3]: examples/example_object_transport.cfg, ensemble member seed 1000"),
    "cfg5_member_1e5_dead20": (lambda orc: _cfg(orc, "example_dead_cells.cfg", nCells=100000, nDead=20000, seed=4002,
                                                light_x=-40.0, light_y=0.0), EPOCHS_SHORT,
                               "configs[4]: one sweep member, 10^5 bots, 20 % dead, light at (-40, 0)"),
}
def _rim_obstacles(orc, name, **over):
    """The reference's obstacle examples verbatim leave their blob 2-3 units away from the obstacles: the first bot of
    example_obstacle.cfg (seed 1000) touches a circle after ~220 000 steps, of example_gap.cfg a wall after ~390 000
    (measured with the oracle) -- BASELINE configs[3]'s 120 000 steps never get there.  So that the obstacle force
    sites are exercised in 10-step windows, these cases MOVE the obstacles to the rim of the placed blob (everything
    else is the file): circles onto its leftmost / topmost / bottom-most bot, overlapping by 0.02; the two walls of
    the gap file against the blob's left rim, the upper wall's inner corner diagonally off the leftmost bot."""
    P, _ = _cfg(orc, name, **over)
    tmp = orc.Sim(P, reset=True)
    pos, rad = tmp.get("pos"), tmp.get("rad")
    tmp.close()
    if P.n_cir_obstacles:
        picks = [int(np.argmin(pos[:, 0])), int(np.argmax(pos[:, 1])), int(np.argmin(pos[:, 1]))]
        dirs = [(-1.0, 0.0), (0.0, 1.0), (0.0, -1.0)]
        for k in range(min(int(P.n_cir_obstacles), 3)):
            i, (ux, uy) = picks[k], dirs[k]
            reach = float(P.r_cir_obs[k]) + float(rad[i]) - 0.02
            P.x_cir_obs[k] = np.float32(pos[i, 0] + ux * reach)
            P.y_cir_obs[k] = np.float32(pos[i, 1] + uy * reach)
    if P.nobstacles:
        # the blob's leftmost bot j sits DIAGONALLY off the upper wall's inner corner, 0.02 inside its reach (the corner
        # branch, impl.cuh:757-779, is the one with powf); the bots above it meet that wall's right face; the lower
        # wall keeps the file's gap width below the corner
        j = int(np.argmin(pos[:, 0]))
        x2 = np.float32(pos[j, 0] - 0.04)
        gap = float(P.y1obs[1]) - float(P.y2obs[0])
        for k in range(int(P.nobstacles)):
            w = float(P.x2obs[k]) - float(P.x1obs[k])
            P.x2obs[k] = x2
            P.x1obs[k] = np.float32(x2 - w)
        P.y1obs[1] = np.float32(pos[j, 1] + 0.04)
        P.y2obs[0] = np.float32(float(P.y1obs[1]) - gap)
    return P, None


# Round 5: the cases that reach the DEVICE powf sites (obstacle and shadow tests, impl.cuh:214-229, 704-779).
CASES.update({
    "cfg4_obstacle_500_rim": (lambda orc: _rim_obstacles(orc, "example_obstacle.cfg", seed=1000), EPOCHS_RIM,
                              "configs[3]'s example_obstacle.cfg, seed 1000, its three circles moved onto the rim of the "
                              "placed blob (the file's own course is not reached within configs[3]'s 120 000 steps)"),
    "cfg_gap_1000_rim": (lambda orc: _rim_obstacles(orc, "example_gap.cfg"), EPOCHS_RIM,
                         "examples/example_gap.cfg, its two walls moved against the placed blob: faces and corners"),
    "cfg4_obstacle_500_late": (lambda orc: _cfg(orc, "example_obstacle.cfg", seed=1000), (230000, 300000, 450000),
                               "configs[3]'s example_obstacle.cfg verbatim, seed 1000, at the steps at which its blob is "
                               "on the obstacle course (1-7 bots on a circle; the oracle walks 450 000 steps: fixture "
                               "only, ~3 min)"),
    "cfg_gap_1000": (lambda orc: _cfg(orc, "example_gap.cfg"), EPOCHS_FULL,
                     "examples/example_gap.cfg verbatim (1000 bots, two rectangular obstacles: faces and corners)"),
    "cfg4_obstacle_500_shadow": (lambda orc: _cfg(orc, "example_obstacle.cfg", seed=1000, light_shadow=1), EPOCHS_FULL,
                                 "configs[3]'s obstacle course with light_shadow 1: every phase update runs "
                                 "checkIntersectionCircle for every bot (epoch 1195 crosses one)"),
})
# cases whose obstacles the device-powf bracket members (orclib.DEVPOWF_VARIANTS) can act on
DEVPOWF_CASES = ("cfg4_obstacle_500", "cfg_gap_1000", "cfg4_obstacle_500_shadow", "cfg4_obstacle_500_rim",
                 "cfg_gap_1000_rim", "cfg4_obstacle_500_late")
# measured by the fixture generator only (the oracle walks 450 000 steps first); the GPU session skips them
FIXTURE_ONLY_CASES = ("cfg4_obstacle_500_late",)
CHEAP_CASES = ("cfg1_example_300", "cfg2a_dead_cells_100", "cfg4_obstacle_500", "cfg4_object_transport_201")


def rel_dev(p, ref):
    p, ref = p.astype(np.float64), ref.astype(np.float64)
    return np.linalg.norm(p - ref, axis=1) / np.linalg.norm(ref, axis=1)


def window_stats(p, ref):
    dev = rel_dev(p, ref)
    dabs = np.linalg.norm(p.astype(np.float64) - ref.astype(np.float64), axis=1)
    cp, cr = p.astype(np.float64).mean(0), ref.astype(np.float64).mean(0)
    # com_rel is relative to |COM|, as the north star words it; a blob centred on the origin (the
    # arena lattice) makes that ill-conditioned, so the absolute figure is kept next to it
    return {"median": float(np.median(dev)), "p99": float(np.quantile(dev, 0.99)), "max": float(dev.max()),
            "flips": int((dev > RTOL).sum()), "com_rel": float(np.linalg.norm(cp - cr) / np.linalg.norm(cr)),
            "com_abs": float(np.linalg.norm(cp - cr)), "max_abs": float(dabs.max())}


class OracleCandidate:
    """A bracket build of the oracle as the candidate."""

    def __init__(self, orc, P, variant):
        self.name = variant
        self.sim = orc.Sim(P, reset=False, variant=variant)

    def start(self, init):
        for k, a in init.items():
            self.sim.set(k, a)

    def sync(self, teacher):
        for k in SYNC_KEYS:
            self.sim.set(k, teacher.view(k))
        self.sim.time = teacher.time

    def step(self, k):
        self.sim.run(k)

    def pos(self):
        return self.sim.view("pos")

    def walk(self, k):
        """the teacher has walked k steps towards the next epoch (nothing to do: sync() is complete)"""

    def resync(self, teacher):
        """end of an un-resynchronised continuation (nothing to do: sync() is complete)"""

    def close(self):
        self.sim.close()


class HipCandidate:
    """The product's streamlined force kernel (pbSimSetForceVariant(sim, 3)) as the candidate.  Between
    epochs the simulation walks with the exact kernel (variant 2), whose positions must equal the
    teacher's bit for bit when the window starts -- the engine's own slot order, cell table and fp32
    clock are then the teacher's by construction."""
    name = "hip_streamlined"

    def __init__(self, pb, P, variant=3):
        from helpers import simparams_from_orc
        sp, keep = simparams_from_orc(P)
        self.g = pb.Sim(sp, keepalive=keep)
        self.g.set_lanes_per_bot(1)   # the streamlined kernel only exists in the one-bot-per-lane form
        self.g.set_resident(1)        # ... and small simulations would otherwise take the resident kernel
        self.variant = variant
        self._p = None

    def start(self, init):
        self.g.set_state(**init)

    def walk(self, k):
        self.g.set_force_variant(2)
        assert self.g.step(k) == k

    def sync(self, teacher):
        from helpers import assert_bit_equal
        st = self.g.get_state()
        for key in ("pos", "vel", "rad", "phase"):
            assert_bit_equal(st[key], teacher.view(key), f"exact kernel at the start of the window: {key}")
        assert self.g.time == teacher.time
        self.g.set_force_variant(self.variant)
        assert self.g.config()["force_variant"] == self.variant

    def step(self, k):
        assert self.g.step(k) == k
        self._p = None

    def pos(self):
        if self._p is None:
            self._p = self.g.get_state()["pos"]
        return self._p

    def resync(self, teacher):
        self.g.set_state(pos=teacher.view("pos"), vel=teacher.view("vel"), rad=teacher.view("rad"),
                         phase=teacher.view("phase"))
        self.g.set_forces(teacher.view("absForce_a"), teacher.view("absForce_r"))
        assert self.g.time == teacher.time

    def close(self):
        self.g.close()


def make_teacher(orc, name, candidates_factory):
    """The reference's placement and its dead-bot draw at t = 0 (particlebot.cpp:178-194) both read
    libc's one global rand() stream, seeded when a simulation is created (main.cpp:929).  A
    throw-away oracle simulation performs both; teacher and candidates are then started from the
    resulting arrays with nDead = 0 (nothing left to draw; nDead has no other reader unless it is
    -1), so that no candidate can disturb the stream and the GPU engine can be handed the same dead
    flags before its first step."""
    build, epochs, what = CASES[name]
    P, init = build(orc)
    if init is None:
        tmp = orc.Sim(P, reset=True)
        init = {k: tmp.get(k) for k in ("pos", "vel", "rad", "phase", "dead")}
        if P.nDead > 0:
            tmp.run(1)
            init["dead"] = tmp.get("dead")
            assert int(init["dead"].sum()) == P.nDead
        tmp.close()
    else:
        tmp = orc.Sim(P, reset=True, hex=True)  # radii, phases of a fresh simulation
        full = {k: tmp.get(k) for k in ("pos", "vel", "rad", "phase", "dead")}
        tmp.close()
        full.update(init)
        init = full
    if P.nDead > 0:
        P.nDead = 0
    cands = candidates_factory(P)
    teacher = orc.Sim(P, reset=False)
    for k, a in init.items():
        teacher.set(k, a)
    for c in cands:
        c.start(init)
    return P, teacher, cands, epochs, what


def measure_case(orc, name, candidates_factory, epochs=None, horizon=HORIZON):
    """-> {candidate name: [per-epoch dict]}; see the module docstring."""
    P, teacher, cands, case_epochs, what = make_teacher(orc, name, candidates_factory)
    epochs = case_epochs if epochs is None else epochs
    out = {c.name: [] for c in cands}
    step = 0
    for E in epochs:
        teacher.run(E - step)
        for c in cands:
            c.walk(E - step)
            c.sync(teacher)
        rec = {c.name: {"epoch": E, "break_max": None, "break_p99": None, "break_com": None} for c in cands}
        for k in range(1, horizon + 1):
            teacher.run(1)
            ref = teacher.view("pos")
            for c in cands:
                c.step(1)
                r = rec[c.name]
                need_full = k == WINDOW
                if need_full or r["break_max"] is None or r["break_p99"] is None or r["break_com"] is None:
                    s = window_stats(c.pos(), ref)
                    if need_full:
                        r["window"] = s
                    if r["break_max"] is None and s["max"] > RTOL:
                        r["break_max"] = k
                    if r["break_p99"] is None and s["p99"] > RTOL:
                        r["break_p99"] = k
                    if r["break_com"] is None and s["com_rel"] > RTOL:
                        r["break_com"] = k
        for c in cands:
            c.resync(teacher)
            out[c.name].append(rec[c.name])
        step = E + horizon
    n = int(P.nCells)
    for c in cands:
        c.close()
    teacher.close()
    return {"case": name, "what": what, "bots": n, "window": WINDOW, "horizon": horizon, "rtol": RTOL,
            "candidates": out}


def summarise(result):
    """one row per candidate: worst-case window statistics over the epochs and the shortest horizons"""
    rows = {}
    for cname, recs in result["candidates"].items():
        w = [r["window"] for r in recs]
        brk = lambda key: min((r[key] for r in recs if r[key] is not None), default=None)
        rows[cname] = {"median_max": max(x["median"] for x in w), "p99_max": max(x["p99"] for x in w),
                       "max_max": max(x["max"] for x in w), "flips_total": sum(x["flips"] for x in w),
                       "flips_worst": max(x["flips"] for x in w), "bot_windows": result["bots"] * len(w),
                       "com_rel_max": max(x["com_rel"] for x in w), "max_abs_max": max(x["max_abs"] for x in w),
                       "first_break_max": brk("break_max"), "first_break_p99": brk("break_p99"),
                       "first_break_com": brk("break_com")}
    return rows


def format_rows(result):
    lines = []
    for cname, r in summarise(result).items():
        lines.append(f"{result['case']:28s} {cname:10s} bots {result['bots']:6d}: 10-step window worst of "
                     f"{len(result['candidates'][cname])} epochs: median {r['median_max']:.2g} p99 {r['p99_max']:.2g} "
                     f"max {r['max_max']:.2g} COM {r['com_rel_max']:.2g}; bots beyond 1e-5: {r['flips_total']} of "
                     f"{r['bot_windows']} bot-windows (worst window {r['flips_worst']}); 1e-5 breaks after "
                     f"max:{r['first_break_max']} p99:{r['first_break_p99']} COM:{r['first_break_com']} steps "
                     f"(None = not within {result['horizon']})")
    return lines
