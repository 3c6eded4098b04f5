"""BASELINE configs[3] and configs[4] at FULL SIZE inside `-m gpu` (until now they ran at full
size only in bench.py), through the product's own multi-GPU runner bin/particlebot_ensemble (world of one rank, the
real RCCL gather):

  * configs[3]: examples/example_obstacle.cfg and examples/example_object_transport.cfg, 256 Monte-Carlo seeds each,
    120 000 timesteps per member (max_time 1200 = 100 actuation cycles), phase noise on;
  * configs[4]: a 64-member slice of the dead-fraction sweep -- examples/example_dead_cells.cfg at 10^5 bots, light at
    (-40, 0), 8 dead fractions 0 ... 0.40 x 8 seeds, 12 000 timesteps (10 cycles), the reference's placement rule.

Size-independent properties: every summary row finite, the row count and clock of the reference's dump gate
(particlebot.cpp:309-310), every member a different blob, progress toward the light declining monotonically with the
dead fraction (results/cfg5_full_sweep.md); and three randomly chosen members of each run are replayed stand-alone on
the CPU oracle over their first 300 timesteps: the same rows to 2e-6 (the device reduces the centre of mass in double
precision; the member's state is bit-identical, tests/test_gpu_baseline_configs.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

from test_gpu_baseline_configs import EX, dump_due, oracle_member

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENS = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_ensemble")


def run_ensemble(tmp_path, cfg, members, sets, sweep=None, extra=(), port="29441"):
    out = tmp_path / "rows.bin"
    cmd = [ENS, EX(cfg), "--members", str(members), "--seed0", "1000", "--out", str(out)]
    for k, v in sets.items():
        cmd += ["--set", k, str(v)]
    if sweep:
        cmd += ["--sweep", sweep[0]] + [str(v) for v in sweep[1]]
    cmd += list(extra)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines   # stdout is rank 0's one JSON line
    info = json.loads(lines[0])
    rows = np.fromfile(out, np.float32).reshape(members, info["rows_per_member"], 4)
    return info, rows


def expected_row_times(max_time, di, dt=0.01):
    """the fp32 clock and dump gate of the reference's display() loop (main.cpp:354-361, particlebot.cpp:174-176,309)"""
    f = np.float32
    t, out, steps = f(0.0), [], 0
    while True:
        if dump_due(t, di):
            out.append(t)
        if t > f(max_time):
            break
        t = f(t + f(dt))
        steps += 1
    return np.array(out, np.float32), steps   # (0.01f > 0.01: the fp32 clock passes 1200 after 119 915 steps)


def replay_three_members(orc, rng, cfg, members, over_of, di, rows):
    """three random members stand-alone on the oracle over their first 300 timesteps"""
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    for k in sorted(rng.choice(members, 3, replace=False)):
        over = dict(over_of(int(k)), max_time=3.0 - 0.005, dump_interval=float(di))  # 300 steps: t = 0 ... 2.99
        orows, osim = oracle_member(orc, EX(cfg), over, float(di))
        osim.close()
        m = len(orows)
        assert m >= 4, (k, m)
        assert np.array_equal(orows[:, 0].astype(np.float32), rows[k, :m, 0]), (k, orows[:, 0], rows[k, :m, 0])
        assert np.abs(orows[:, 1:] - rows[k, :m, 1:]).max() < 2e-6, (k, orows, rows[k, :m])


@pytest.mark.parametrize("cfg,bots", [("example_obstacle.cfg", 500), ("example_object_transport.cfg", 201)])
def test_config4_all_256_seeds_at_full_length(tmp_path, orc, cfg, bots):
    members, di = 256, 1.0
    info, rows = run_ensemble(tmp_path, cfg, members, {"max_time": "1200", "dump_interval": di})
    assert info["members"] == members and info["bots_per_member"] == bots and info["n_gpus"] == 1
    want_t, want_steps = expected_row_times(1200.0, di)
    assert info["steps_per_member"] == want_steps and 119000 < want_steps <= 120001
    assert rows.shape == (members, len(want_t), 4) and len(want_t) >= 1200
    assert np.isfinite(rows).all()
    assert all(np.array_equal(rows[k, :, 0], want_t) for k in range(members))   # one clock, the reference's
    assert len({tuple(np.round(r[0, 1:3], 5)) for r in rows}) == members         # 256 different blobs
    # every blob moved, on average toward the light (100 actuation cycles)
    progress = rows[:, 0, 3].astype(np.float64) - rows[:, -1, 3]
    assert progress.mean() > 0.2 and (progress > 0).mean() > 0.95, (progress.mean(), (progress > 0).mean())
    assert abs(info["progress_toward_light_mean"] - progress.mean()) < 1e-5
    print(f"{cfg}: 256 seeds x {info['steps_per_member']} steps in {info['wall_s']:.2f} s "
          f"({info['particle_steps_per_s']:.3g} particle-steps/s end to end); progress toward the light "
          f"{progress.mean():.4f} +- {progress.std():.4f}")
    replay_three_members(orc, np.random.default_rng(7), cfg, members, lambda k: dict(seed=1000 + k), di, rows)


def test_config5_slice_64_members_dead_fraction_trend(tmp_path, orc):
    members, di = 64, 1.0
    dead = [int(round(0.40 * i / 7 * 100000)) for i in range(8)]   # 0 ... 40 000 of 100 000
    sets = {"nCells": "100000", "light_x": "-40", "light_y": "0", "max_time": "120", "dump_interval": di}
    info, rows = run_ensemble(tmp_path, "example_dead_cells.cfg", members, sets, sweep=("nDead", dead),
                              extra=["--sub-batch", "-1"], port="29443")
    want_t, want_steps = expected_row_times(120.0, di)
    assert info["members"] == members and info["bots_per_member"] == 100000
    assert info["steps_per_member"] == want_steps and 11900 < want_steps <= 12001
    assert rows.shape == (members, len(want_t), 4) and np.isfinite(rows).all()
    assert all(np.array_equal(rows[k, :, 0], want_t) for k in range(members))
    # member k has dead count dead[k % 8] (the runner's --sweep rule) and seed 1000 + k
    progress = rows[:, 0, 3].astype(np.float64) - rows[:, -1, 3]
    by_f = np.array([progress[j::8].mean() for j in range(8)])
    spread = np.array([progress[j::8].std() for j in range(8)])
    print("dead fraction -> progress toward the light over 120 s (8 seeds each):",
          ", ".join(f"{d / 1e5:.3f}: {m:.5f} +- {s:.5f}" for d, m, s in zip(dead, by_f, spread)))
    assert np.all(np.diff(by_f) < 0), by_f                      # monotone decline (results/cfg5_full_sweep.md)
    assert 0.07 < by_f[0] < 0.10 and by_f[-1] < 0.5 * by_f[0]    # 0.0842 at f = 0 in the full sweep
    assert np.all(spread < 0.25 * np.abs(np.diff(by_f)).min() + 1e-3)
    pl = info["pipeline_rank0"]
    print(f"64 members x 10^5 bots x {info['steps_per_member']} steps in {info['wall_s']:.1f} s; pipeline {pl}")
    replay_three_members(orc, np.random.default_rng(8), "example_dead_cells.cfg", members,
                         lambda k: dict(seed=1000 + k, nDead=dead[k % 8], nCells=100000, light_x=-40.0, light_y=0.0),
                         di, rows)


def test_config5_slice_statistic_is_the_same_with_the_tolerance_kernel(tmp_path):
    """The same 64-member slice with `pb_force_variant 3` (the opt-in tolerance kernel, selected through the members'
    configuration): individual trajectories are not the exact kernels' after 12 000 free-running steps (chaos:
    DESIGN.md section 4), the sweep's STATISTIC is -- every per-fraction mean of the progress toward the light within
    three standard errors of the exact run's (measured: <= 2.9e-4 of 0.02-0.08, i.e. 0.5 of the seed spread), the same
    monotone decline, the fractions an order of magnitude further apart than the two realisations."""
    members, di = 64, 6.0
    dead = [int(round(0.40 * i / 7 * 100000)) for i in range(8)]
    sets = {"nCells": "100000", "light_x": "-40", "light_y": "0", "max_time": "120", "dump_interval": di}
    out = {}
    for name, extra in (("exact", {}), ("tolerance", {"pb_force_variant": "3"})):
        d = tmp_path / name
        d.mkdir()
        info, rows = run_ensemble(d, "example_dead_cells.cfg", members, dict(sets, **extra), sweep=("nDead", dead),
                                  extra=["--sub-batch", "-1"], port="29445" if name == "exact" else "29447")
        assert np.isfinite(rows).all()
        progress = rows[:, 0, 3].astype(np.float64) - rows[:, -1, 3]
        out[name] = (np.array([progress[j::8].mean() for j in range(8)]), np.array([progress[j::8].std() for j in range(8)]),
                     info["wall_s"], progress)
    (m_e, s_e, w_e, p_e), (m_t, s_t, w_t, p_t) = out["exact"], out["tolerance"]
    print("exact     :", np.round(m_e, 5), f"{w_e:.1f} s")
    print("tolerance :", np.round(m_t, 5), f"{w_t:.1f} s")
    assert not np.array_equal(p_e, p_t)                       # another kernel really ran
    assert np.all(np.diff(m_t) < 0)
    # two realisations of the same chaotic ensemble: the means of 8 seeds agree within 3 standard errors ...
    se = np.maximum(s_e, s_t) / np.sqrt(8.0)
    assert np.all(np.abs(m_t - m_e) <= 3.0 * se + 1e-4), (m_t - m_e, se)
    # ... the fractions stay apart by far more than that, and member by member the difference is inside the seed spread
    assert np.abs(m_t - m_e).max() <= 0.1 * np.abs(np.diff(m_e)).min()
    assert np.abs(p_t - p_e).max() <= 5 * s_e.max(), (np.abs(p_t - p_e).max(), s_e.max())


def test_cartesian_sweep_places_each_seed_once_and_members_are_the_oracles(tmp_path, orc):
    """`particlebot_ensemble --sweep nDead ... --cartesian` (round 6): the swept values form a grid and every grid point
    runs under each seed -- member k = grid point k mod G under seed seed0 + k / G, BASELINE configs[4]'s "64 points x
    16 seeds" in small -- so the members of one seed share ONE placement (and the generator state after it); every
    member still equals its own stand-alone oracle run."""
    di, dead = 6.0, [0, 40, 150, 320]
    sets = {"nCells": "2000", "light_x": "-9", "light_y": "0", "max_time": "6.1", "dump_interval": str(di)}
    info, rows = run_ensemble(tmp_path, "example_dead_cells.cfg", 12, sets, sweep=("nDead", dead), extra=("--cartesian",),
                              port="29453")
    assert info["members"] == 12 and info["bots_per_member"] == 2000
    assert info["collective"]["placements_run_rank0"] == 3           # three seeds, four dead fractions each
    over_of = lambda k: dict(seed=1000 + k // 4, nDead=dead[k % 4], nCells=2000, light_x=-9.0, light_y=0.0)
    # the same blob under the four fractions of a seed: identical at t = 0, apart afterwards
    assert np.array_equal(rows[0, 0, 1:3], rows[3, 0, 1:3]) and not np.array_equal(rows[0, -1, 1:3], rows[3, -1, 1:3])
    assert not np.array_equal(rows[0, 0, 1:3], rows[4, 0, 1:3])
    for k in range(12):
        orows, osim = oracle_member(orc, EX("example_dead_cells.cfg"), dict(over_of(k), max_time=6.1), di)
        assert np.array_equal(orows[:, 0].astype(np.float32), rows[k, :, 0]), k
        assert np.abs(orows[:, 1:] - rows[k, :, 1:]).max() < 2e-6, k
        osim.close()


def test_config5_whole_sweep_1024_members_reproduces_the_recorded_statistic(tmp_path, orc):
    """BASELINE configs[4] WHOLE under the driver's eyes (until now a builder soak): all 1 024
    members -- 64 dead fractions 0 ... 0.40 x 16 seeds of examples/example_dead_cells.cfg at 10^5 bots, light at
    (-40, 0), 12 000 timesteps -- in ONE invocation of bin/particlebot_ensemble (--sub-batch -1: two lanes of
    sub-batches, the RCCL gather with a world of one), ~85 s on one MI355X.  The members are those of
    results/cfg5_full_sweep.json (round 1: four runs of 256 members, seeds 1000+k / 2000+k / 3000+k / 4000+k, dead
    count nDead[k mod 64]; here `--sweep seed ...` next to `--sweep nDead ...`), and because every exact kernel of every
    round is bit-identical to the oracle the statistic must be the recorded one TO THE LAST DIGIT: every member's
    progress toward the light (fp32 difference of the distance column), hence every per-fraction mean.  Also: the
    reference's clock and row count, the monotone decline over all 64 fractions, a device-bound pipeline on a box with
    >= 16 usable CPUs, and three random members replayed on the oracle."""
    rec = json.load(open(os.path.join(ROOT, "results", "cfg5_full_sweep.json")))
    dead = [int(x) for x in rec["nDead"]]
    assert len(dead) == 64 and dead[0] == 0 and dead[-1] == 40000
    members = 1024
    seeds = [1000 * (k // 256 + 1) + k % 256 for k in range(members)]
    di = 1.0   # (the record was made with the file's own dump interval: the first and the last row are the same states)
    sets = {"nCells": "100000", "light_x": "-40", "light_y": "0", "max_time": "120", "dump_interval": str(di)}
    out = tmp_path / "rows.bin"
    cmd = [ENS, EX("example_dead_cells.cfg"), "--members", str(members), "--seed0", "1000", "--out", str(out), "--sub-batch", "-1"]
    for k, v in sets.items():
        cmd += ["--set", k, v]
    cmd += ["--sweep", "nDead"] + [str(d) for d in dead] + ["--sweep", "seed"] + [str(x) for x in seeds]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29449")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    info = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    rows = np.fromfile(out, np.float32).reshape(members, info["rows_per_member"], 4)
    want_t, want_steps = expected_row_times(120.0, di)
    assert info["members"] == members and info["bots_per_member"] == 100000 and info["n_gpus"] == 1
    assert info["steps_per_member"] == want_steps and 11900 < want_steps <= 12001
    assert rows.shape == (members, len(want_t), 4) and np.isfinite(rows).all()
    assert all(np.array_equal(rows[k, :, 0], want_t) for k in range(members))       # one clock, the reference's
    assert len({tuple(np.round(r[0, 1:3], 5)) for r in rows}) == members            # 1 024 different blobs
    progress = rows[:, 0, 3] - rows[:, -1, 3]                                        # fp32, as the record was made
    assert progress.dtype == np.float32
    by_f = np.zeros(64)
    worst = 0.0
    for j, d in enumerate(dead):
        mine = progress[j::64]                         # k = j, j + 64, ...: batch-major, as recorded
        want = np.array(rec["progress"][str(d)], np.float32)
        assert mine.shape == want.shape == (16,)
        worst = max(worst, float(np.abs(mine.astype(np.float64) - want).max()))
        assert np.array_equal(mine, want), (d, mine, want)
        by_f[j] = mine.astype(np.float64).mean()
        assert by_f[j] == want.astype(np.float64).mean()
    assert np.all(np.diff(by_f) < 0), by_f                                           # strictly monotone over all 64
    assert abs(by_f[0] - 0.08422) < 1e-4 and abs(by_f[-1] - 0.0215) < 1e-3           # results/cfg5_full_sweep.md
    pl = info["pipeline_rank0"]
    usable = orc.usable_cpus()
    print(f"configs[4] whole: 1 024 members x 10^5 bots x {want_steps} steps in {info['wall_s']:.1f} s "
          f"({info['particle_steps_per_s']:.3g} particle-steps/s end to end), pipeline {pl}; usable CPUs {usable}; "
          f"statistic equal to results/cfg5_full_sweep.json to the last digit (max |diff| {worst})")
    if usable >= 16:
        assert pl["bound"] == "device", pl
    assert pl["lanes"] == 2 and pl["sub_batches"] >= 30
    # what the communicator itself reported (ncclCommCount), and: 1 024 seeds = 1 024 placements, nothing shared
    c = info["collective"]
    assert c["backend"] == "rccl" and c["ranks"] == 1 and c["rank0_device"] == 0 and c["placements_run_rank0"] == 1024
    over_of = lambda k: dict(seed=seeds[k], nDead=dead[k % 64], nCells=100000, light_x=-40.0, light_y=0.0)
    replay_three_members(orc, np.random.default_rng(9), "example_dead_cells.cfg", members, over_of, di, rows)
