"""GPU tests of the C++ host side: class Particlebot (both engines), the headless runner and the CSV
dump / reload, against the oracle's restatement of the reference class.  CSV files are compared
BYTE FOR BYTE: that covers the .cfg loader, srand + random placement, the dead-bot draw, every
kernel, the fp32 dump schedule and the %f formatting in one go."""
import os
import subprocess

import numpy as np
import pytest

from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = lambda name: os.path.join(ROOT, "examples", name)


@pytest.fixture(scope="module")
def host():
    from particlerobotsimulations_amd import host
    host.lib()
    return host


def oracle_csv(orc, cfg, path, **over):
    """display() loop of main.cpp:354-361 on the oracle: dump, then update, until update says stop."""
    P = orc.load_cfg(cfg, **over)
    sim = orc.Sim(P)
    open(path, "w").close()
    while True:
        sim.dump(path)
        if sim.update():
            break
    return sim


def product_csv(host, cfg, path, engine, batch=True, **over):
    sim = host.HostSim(cfg, engine=engine, **over)
    open(path, "w").close()
    while True:
        sim.dump(path)
        if sim.finished:
            break
        if batch:
            # everything up to (not past) the next dump row in one call: the runner's batching
            if sim.advance(sim.steps_until_dump()) == 0:
                break
        else:
            sim.update()
    return sim


CASES = [
    ("example_dead_cells.cfg", dict(max_time=1.3, testing=1, dump_interval=0.5)),
    ("example.cfg", dict(max_time=0.8, testing=1, dump_interval=0.25, phase_std=0)),
    ("example_object_transport.cfg", dict(max_time=12.5, testing=0, dump_interval=6)),
    ("example_obstacle.cfg", dict(max_time=0.6, testing=1, dump_interval=0.3)),
    ("example_gap.cfg", dict(max_time=0.4, testing=1, dump_interval=0.2)),
]


@pytest.mark.parametrize("engine", ["fused", "legacy"])
@pytest.mark.parametrize("cfg,over", CASES, ids=[c[0] for c in CASES])
def test_csv_byte_identical(host, orc, tmp_path, cfg, over, engine):
    a, b = str(tmp_path / "oracle.csv"), str(tmp_path / f"{engine}.csv")
    osim = oracle_csv(orc, EX(cfg), a, **over)
    gsim = product_csv(host, EX(cfg), b, engine, batch=(engine == "fused"), **{k: str(v) for k, v in over.items()})
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert len(da) > 200 and da.count(b"\n") >= 3
    assert da == db
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(gsim.get(k), osim.get(k), k)
    assert_bit_equal(gsim.get("dead"), osim.get("dead"), "dead")


def test_dead_draw_uses_libc_rand_after_placement(host, orc):
    """particlebot.cpp:178-194: the nDead bots are drawn with rand() continuing the stream that the
    placement consumed; a different time_to_dead moves the draw to a later step."""
    over = dict(max_time=100, time_to_dead=0.05)
    # product first: creating it calls srand() as main.cpp:929 does, which would rewind the
    # process-global libc stream the ORACLE draws its dead set from
    gsim = host.HostSim(EX("example_dead_cells.cfg"), **{k: str(v) for k, v in over.items()})
    P = orc.load_cfg(EX("example_dead_cells.cfg"), **over)
    osim = orc.Sim(P)
    for _ in range(4):
        osim.update()
    gsim.advance(4)
    assert gsim.get("dead").sum() == 0
    for _ in range(8):
        osim.update()
    gsim.advance(8)
    assert gsim.get("dead").sum() == 20
    assert_bit_equal(gsim.get("dead"), osim.get("dead"), "dead set")
    assert_bit_equal(gsim.get("pos"), osim.get("pos"), "pos")


def test_update_one_step_at_a_time_equals_batched(host):
    a = host.HostSim(EX("example.cfg"), max_time="100")
    b = host.HostSim(EX("example.cfg"), max_time="100")
    for _ in range(60):
        a.update()
    assert b.advance(60) == 60
    assert a.time == b.time
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(a.get(k), b.get(k), k)


def test_load_from_file_resume(host, orc, tmp_path):
    """loadFromFile (particlebot.cpp:369-411): the last complete row of a testing=1 CSV restores
    time, pos, vel, rad (6-decimal text, so the resumed run is compared with an oracle resumed from
    the same file, not with the uninterrupted run)."""
    csv = str(tmp_path / "ckpt.csv")
    over = dict(max_time=0.35, testing=1, dump_interval=0.1, phase_std=0)
    product_csv(host, EX("example.cfg"), csv, "fused", **{k: str(v) for k, v in over.items()})
    P = orc.load_cfg(EX("example.cfg"), **over)
    P.max_time = 1e9
    osim = orc.Sim(P)
    assert osim.load_from_file(csv) == 0
    osim.force_sort_once()  # the product sorts on its first step; the reference would not (see oracle)
    gsim = host.HostSim(EX("example.cfg"), max_time="1e9", testing="1", phase_std="0")
    gsim.load_from_file(csv)
    assert gsim.time == osim.time and gsim.time > 0.25
    for k in ("pos", "vel", "rad"):
        assert_bit_equal(gsim.get(k), osim.get(k), "restored " + k)
    osim.run(25)
    gsim.advance(25)
    for k in ("pos", "vel", "rad"):
        assert_bit_equal(gsim.get(k), osim.get(k), "resumed " + k)


def test_set_get_array(host):
    s = host.HostSim(EX("example_dead_cells.cfg"))
    pos = s.get("pos")
    pos[10:20] += np.float32(0.5)
    s.set("pos", pos[10:20], start=10)
    assert_bit_equal(s.get("pos"), pos, "pos after setArray")
    rad = np.full(100, 0.1, np.float32)
    s.set("rad", rad)
    assert_bit_equal(s.get("rad"), rad, "rad")


@pytest.mark.parametrize("engine", ["fused", "legacy"])
def test_set_array_subrange_after_stepping_leaves_other_bots_alone(host, engine):
    """setArray(array, data, start, count) on a simulation that has already stepped touches only the
    bots [start, start + count) (particlebot.cpp:834-867): every other bot keeps its CURRENT device
    state, not a stale host copy."""
    s = host.HostSim(EX("example_dead_cells.cfg"), engine=engine, max_time="1e9")
    s.advance(150)                      # through a re-sort: slot order != original order
    before = {k: s.get(k) for k in ("pos", "vel", "rad", "phase")}
    assert np.abs(before["vel"]).max() > 0
    new_pos = before["pos"][40:43] + np.float32(0.25)
    new_vel = np.full((1, 2), 0.5, np.float32)
    new_rad = np.full(5, 0.1, np.float32)
    new_phase = np.full(2, -1.5, np.float32)
    s.set("pos", new_pos, start=40)
    s.set("vel", new_vel, start=7)
    s.set("rad", new_rad, start=95)
    s.set("phase", new_phase, start=0)
    want = {k: v.copy() for k, v in before.items()}
    want["pos"][40:43] = new_pos
    want["vel"][7:8] = new_vel
    want["rad"][95:100] = new_rad
    want["phase"][0:2] = new_phase
    for k in want:
        assert_bit_equal(s.get(k), want[k], f"{engine}: {k} after ranged setArray")
    s.advance(10)                       # and the simulation carries on from there
    assert np.isfinite(s.get("pos")).all()


def test_headless_runner_binary(orc, tmp_path):
    """particlebot_run <cfg>: the reference's main() minus the window, end to end."""
    exe = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_run")
    out = tmp_path / "run.csv"
    over = dict(max_time=2.5, dump_interval=1, testing=1, csv_filename=str(out))
    args = [exe, EX("example_obstacle.cfg"), "--quiet"]
    for k, v in over.items():
        args += ["--set", k, str(v)]
    subprocess.check_call(args, cwd=str(tmp_path), timeout=300)
    ref = str(tmp_path / "oracle.csv")
    oracle_csv(orc, EX("example_obstacle.cfg"), ref, **{k: v for k, v in over.items() if k != "csv_filename"})
    assert open(ref, "rb").read() == out.read_bytes()


def test_ensemble_members_equal_individual_runs(host, tmp_path):
    """a batched ensemble (one pbSim, one launch per timestep for all members) gives
    every member exactly the state a stand-alone run with that seed gives: placement, dead-bot draw
    (example_dead_cells: 20 of 100, drawn at t = 0 from each member's own stream), phase noise."""
    import ctypes as C
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example_dead_cells.cfg")
    common = {"max_time": "1.55", "dump_interval": "0.5"}
    seeds = [6666, 1, 2, 3, 4, 5, 77, 123456]
    rows, steps, states = ensemble.run_local(cfg, [f"seed\n{s}" for s in seeds], common, final_state=True)
    assert rows.shape[0] == len(seeds) and rows.shape[2] == 4 and steps >= 155
    assert np.allclose(rows[:, 0, 0], 0.0) and np.all(np.diff(rows[0, :, 0]) > 0)
    # the same members one at a time through class Particlebot
    for k, s in enumerate(seeds):
        solo = host.HostSim(cfg, seed=str(s), **common)
        coms = []
        while True:
            if solo.steps_until_dump(1) >= 1 and abs(solo.time - rows[k, len(coms), 0]) < 1e-7:
                coms.append(solo.get("pos").astype(np.float64).mean(0))
            if solo.finished or len(coms) == rows.shape[1]:
                break
            solo.advance(1)
        assert len(coms) == rows.shape[1]
        assert np.abs(np.array(coms) - rows[k, :, 1:3]).max() < 1e-6, (k, coms, rows[k])
        while not solo.finished:
            solo.advance(1)
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(states[k][key], solo.get(key), f"member {k} final {key}")
    # different seeds really are different blobs
    assert len({tuple(np.round(r[-1, 1:3], 5)) for r in rows}) == len(seeds)


def test_checkpoint_resume_with_xorwow_noise_restores_the_generator(host, tmp_path):
    """A checkpoint of a `pb_rng curand` run records the generator kind; resuming into a simulation
    created WITHOUT the key switches the generator and replays the draws: the resumed run is
    bit-identical to the uninterrupted one across later phase updates (noise on)."""
    ck = str(tmp_path / "xw.pbck")
    over = dict(max_time="1e9", phase_update_interval="3")
    a = host.HostSim(EX("example_dead_cells.cfg"), pb_rng="curand", **over)
    a.advance(450)          # phase updates at t = 0 and 3: one Box-Muller pair consumed
    a.save_checkpoint(ck)
    a.advance(500)          # updates at 6 (fresh pair) and 9 (cached value)
    b = host.HostSim(EX("example_dead_cells.cfg"), reset=False, **over)   # default generator
    b.load_checkpoint(ck)
    b.advance(500)
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(a.get(k), b.get(k), k)
    c = host.HostSim(EX("example_dead_cells.cfg"), **over)               # the default generator's run differs
    c.advance(950)
    assert not np.array_equal(c.get("phase"), a.get("phase"))


def test_exact_checkpoint_resume(host, tmp_path):
    """saveCheckpoint/loadCheckpoint keep what the reference's CSV resume loses (phase,
    dead flags, noise-draw counter, the generator, contact forces and the STALE slot layout), so a
    resumed run is bit-identical to the uninterrupted one -- across a phase update with noise, with
    dead bots drawn after the checkpoint."""
    ck = str(tmp_path / "run.pbck")
    over = dict(max_time="1e9", time_to_dead="9.0", sort_interval="5.0")
    a = host.HostSim(EX("example_dead_cells.cfg"), **over)
    a.advance(700)          # t = 7: two re-sorts behind us, stale lists in use
    a.save_checkpoint(ck)
    a.advance(900)          # dead draw at t = 9, phase update (noise) at t = 12, more re-sorts
    b = host.HostSim(EX("example_dead_cells.cfg"), reset=False, **over)
    b.load_checkpoint(ck)
    assert abs(b.time - 7.0) < 1e-3
    b.advance(900)
    assert a.time == b.time
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(a.get(k), b.get(k), k)
    assert_bit_equal(a.get("dead"), b.get("dead"), "dead")
    assert a.get("dead").sum() == 20
    # a truncated file is rejected
    open(ck, "r+b").truncate(100)
    c = host.HostSim(EX("example_dead_cells.cfg"), reset=False, **over)
    with pytest.raises(OSError):
        c.load_checkpoint(ck)


def test_million_bot_cfg_runs_headless(tmp_path):
    """examples/million_bots.cfg through the runner binary: generalised arena + square-lattice
    placement + one fused kernel per step; the CSV centroid stays at the lattice centre."""
    exe = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_run")
    out = tmp_path / "m.csv"
    subprocess.check_call([exe, EX("million_bots.cfg"), "--quiet", "--set", "max_time", "0.5", "--set",
                           "dump_interval", "0.25", "--set", "csv_filename", str(out)], cwd=str(tmp_path), timeout=600)
    rows = [r for r in out.read_text().strip().split("\n")[2:]]
    assert len(rows) >= 3
    last = [float(x) for x in rows[-1].strip(",").split(",")]
    assert abs(last[1]) < 1e-3 and abs(last[2]) < 1e-3 and 229.9 < last[3] < 230.1


@pytest.mark.parametrize("trial", range(6))
def test_random_cfg_files_end_to_end(host, orc, tmp_path, trial):
    """Loader + class (placement, dead-bot draw, dump schedule) + fused engine on randomly written
    .cfg files, against the oracle's loader + whole-simulation object: the CSV is byte-identical and
    so is every state array at the end."""
    rng = np.random.default_rng(700 + trial)
    n = int(rng.integers(40, 400))
    lines = ["# random configuration %d" % trial,
             "nCells", str(n),
             "nDead", str(int(rng.integers(0, n // 3)) if trial % 3 else (-1 if trial == 3 else 0)),
             "seed", str(int(rng.integers(1, 99999))),
             "light_x", f"{rng.uniform(-6, 6):.3f}", "light_y", f"{rng.uniform(-6, 6):.3f}",
             "time_to_dead", f"{rng.uniform(0.02, 0.5):.3f}",
             "spring", f"{rng.uniform(300, 2000):.1f}", "damping", f"{rng.uniform(1, 20):.2f}",
             "shear", f"{rng.uniform(5, 50):.2f}", "friction", f"{rng.uniform(0.1, 0.8):.3f}",
             "phase_std", f"{rng.choice([0.0, 0.4, 0.9])}", "rise_period", f"{rng.choice([1, 2, 3])}",
             "dump_interval", f"{rng.choice([0.25, 0.5])}", "testing", str(trial % 2),
             "max_time", f"{rng.uniform(1.0, 1.6):.2f}",
             "csv_filename", "unused.csv"]
    if trial % 2:
        lines += ["n_cir_obstacles", "2", "x_cir_obs", "2.5 6.0", "y_cir_obs", "0.4 -0.8", "r_cir_obs", "0.35 0.3",
                  "light_shadow", str(1 + trial % 2)]
    if trial == 3:
        lines += ["massFactor", "2.0", "attractionFactor", "0.4", "radFactor", "1.5", "frictionFactor", "1.3"]
    cfg = tmp_path / f"random{trial}.cfg"
    cfg.write_text("\n".join(lines) + "\n")
    a, b = str(tmp_path / "oracle.csv"), str(tmp_path / "product.csv")
    gsim = product_csv(host, str(cfg), b, "fused")          # product first (it calls srand())
    osim = oracle_csv(orc, str(cfg), a)
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert da.count(b"\n") >= 3 and da == db
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(gsim.get(k), osim.get(k), k)
    assert_bit_equal(gsim.get("dead"), osim.get("dead"), "dead")


def test_cxx_rccl_ensemble_runner_equals_python_layer(tmp_path):
    """bin/particlebot_ensemble (C++ only: pbEnsemble* + one ncclAllGather over RCCL, here a world of
    one rank) writes the same summary rows as ensemble.run_local for the same members."""
    import json
    from particlerobotsimulations_amd import ensemble
    exe = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_ensemble")
    cfg = EX("example_dead_cells.cfg")
    out = tmp_path / "rows.f32"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([exe, cfg, "--members", "6", "--seed0", "500", "--set", "max_time", "1.55", "--set",
                        "dump_interval", "0.5", "--sweep", "nDead", "0", "10", "20", "--out", str(out)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]      # ONE line on stdout (RCCL's version banner goes to stderr)
    info = json.loads(lines[0])
    assert info["members"] == 6 and info["n_gpus"] == 1 and info["bots_per_member"] == 100
    got = np.fromfile(out, np.float32).reshape(6, info["rows_per_member"], 4)
    over = [ensemble.member_overrides(k, 500, ("nDead", ["0", "10", "20"])) for k in range(6)]
    rows, steps = ensemble.run_local(cfg, over, {"max_time": "1.55", "dump_interval": "0.5"})
    assert steps == info["steps_per_member"]
    assert np.array_equal(got, rows)


def _render_expected(P, pos, rad, dead, size, cx, cy, half, light_radius=0.25):
    """Independent numpy rasteriser of the headless frame (stand-in for display() +
    updateCol_k, particlebot_kernel_impl.cuh:401-443), painter's order: rectangles, circles, light,
    bots by index; float32 arithmetic as in Particlebot::writeFramePPM."""
    f = np.float32
    img = np.full((size, size, 3), 245, np.uint8)
    scale = f(0.5) * f(size) / f(half)
    px = lambda x: f(0.5) * f(size) - (f(x) - f(cx)) * scale
    py = lambda y: f(0.5) * f(size) - (f(y) - f(cy)) * scale
    yy, xx = np.mgrid[0:size, 0:size]
    xc, yc = xx.astype(f) + f(0.5), yy.astype(f) + f(0.5)

    def disc(x, y, r, col):
        ax, ay, pr = px(x), py(y), f(r) * scale
        x0, x1 = max(0, int(np.floor(ax - pr))), min(size - 1, int(np.ceil(ax + pr)))
        y0, y1 = max(0, int(np.floor(ay - pr))), min(size - 1, int(np.ceil(ay + pr)))
        if x0 > x1 or y0 > y1:
            return
        dx, dy = xc[y0:y1 + 1, x0:x1 + 1] - ax, yc[y0:y1 + 1, x0:x1 + 1] - ay
        m = (dx * dx + dy * dy) <= pr * pr
        img[y0:y1 + 1, x0:x1 + 1][m] = col

    for k in range(P.nobstacles):
        xa, xb = px(P.x2obs[k]), px(P.x1obs[k])
        ya, yb = py(P.y2obs[k]), py(P.y1obs[k])
        img[max(0, int(np.floor(ya))):min(size - 1, int(np.ceil(yb))) + 1,
            max(0, int(np.floor(xa))):min(size - 1, int(np.ceil(xb))) + 1] = 110
    for k in range(P.n_cir_obstacles):
        disc(P.x_cir_obs[k], P.y_cir_obs[k], P.r_cir_obs[k], (110, 110, 110))
    disc(P.light_x, P.light_y, light_radius, (250, 210, 40))
    span = f(P.max_radius) - f(P.min_radius)
    for i in range(len(rad)):
        r = f(rad[i])
        col = (0, 0, 0)
        if not dead[i]:
            g = (f(P.max_radius) - r) / span
            b = (r - f(P.min_radius)) / span
            G = min(f(255), max(f(0), f(20) + f(180) * g * g))
            B = min(f(255), max(f(0), f(30) + f(180) * np.sqrt(max(f(0), b), dtype=f)))
            col = (30, int(G), int(B))
        disc(pos[i, 0], pos[i, 1], r, col)
    return img


@pytest.mark.parametrize("cfg,steps", [("example_dead_cells.cfg", 150), ("example_obstacle.cfg", 260),
                                       ("example_gap.cfg", 120)])
def test_headless_frame_of_gpu_state_equals_render_of_oracle_state(host, orc, tmp_path, cfg, steps):
    """the frame the product writes from the fused engine's state after `steps`
    timesteps (radii mid-actuation, dead bots, circle / rectangle obstacles, the light) is, pixel for
    pixel, what an independent rasteriser draws from the ORACLE's state at the same step."""
    over = dict(max_time="1e9")
    sim = host.HostSim(EX(cfg), **over)
    sim.advance(steps)
    path = str(tmp_path / "f.ppm")
    size, half, cam_x = 500, 6.5, 2.0   # blob around (5, 0), light at x = -2 ... -5, obstacles between
    c = host.load_config(EX(cfg))
    sim.write_frame(path, size=size, center=(cam_x, 0.0), half_extent=half)
    raw = open(path, "rb").read()
    header = f"P6\n{size} {size}\n255\n".encode()
    got = np.frombuffer(raw[len(header):], np.uint8).reshape(size, size, 3)
    P = orc.load_cfg(EX(cfg), max_time=1e9)
    osim = orc.Sim(P)
    osim.run(steps)
    assert_bit_equal(sim.get("pos"), osim.get("pos"), "state the frame is drawn from")
    want = _render_expected(P, osim.get("pos"), osim.get("rad"), osim.get("dead"), size, cam_x, 0.0, half,
                            light_radius=c.light_radius)
    assert np.array_equal(got, want), int((got != want).any(axis=2).sum())
    # the frame shows what it should: bots of several radii (several blues), grey obstacles if any
    assert len({tuple(p) for p in got.reshape(-1, 3)[::7]}) > 5
    if c.nobstacles or c.n_cir_obstacles:
        assert ((got == 110).all(axis=2)).sum() > 50
