"""Command-line resume and checkpoints.  The reference's main() has a resume
branch (/root/reference main.cpp:940-957 -> particlebot.cpp:369-411) that restarts from the last CSV row; the
runners here also write EXACT checkpoints while they run, and a run that is killed and resumed from its last
checkpoint must equal the uninterrupted run bit for bit: the same CSV bytes, the same final state."""
import os
import signal
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_run")
ENS = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_ensemble")
EX = lambda name: os.path.join(ROOT, "examples", name)


def run(args, **kw):
    return subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)


def state_part(path):
    """a checkpoint file without the runner's trailer (which holds the CSV length and step counters)"""
    b = open(path, "rb").read()
    return b[:-24]


@pytest.mark.parametrize("cfg,over", [("example_obstacle.cfg", ["--set", "max_time", "30"]),
                                      ("example_dead_cells.cfg", ["--set", "max_time", "25", "--set", "time_to_dead", "3",
                                                                  "--set", "pb_rng", "curand"])])
def test_run_killed_and_resumed_equals_the_uninterrupted_run(tmp_path, cfg, over):
    base = [RUN, EX(cfg), "--quiet", "--set", "dump_interval", "1", "--set", "testing", "1"] + over
    a_csv, b_csv = str(tmp_path / "a.csv"), str(tmp_path / "b.csv")
    # A: uninterrupted (checkpointing all the same: taking checkpoints must not change anything)
    r = run(base + ["--set", "csv_filename", a_csv, "--checkpoint", str(tmp_path / "a.ck"), "--checkpoint-steps", "333",
                    "--final-checkpoint", str(tmp_path / "a.final")])
    assert r.returncode == 0, r.stderr
    # B: dies at step 1200 (a checkpoint every 333 steps: the last one is from step 999; rows up to t = 12 are in
    # the CSV by then and must be cut back), then resumes
    r = run(base + ["--set", "csv_filename", b_csv, "--checkpoint", str(tmp_path / "b.ck"), "--checkpoint-steps", "333",
                    "--stop-after-steps", "1200"])
    assert r.returncode == 9, (r.returncode, r.stderr)
    assert os.path.getsize(b_csv) > 0 and os.path.exists(tmp_path / "b.ck")
    r = run(base + ["--set", "csv_filename", b_csv, "--resume", str(tmp_path / "b.ck"), "--checkpoint",
                    str(tmp_path / "b.ck"), "--checkpoint-steps", "333", "--final-checkpoint", str(tmp_path / "b.final")])
    assert r.returncode == 0, r.stderr
    assert open(a_csv, "rb").read() == open(b_csv, "rb").read()
    assert state_part(tmp_path / "a.final") == state_part(tmp_path / "b.final")
    assert len(open(a_csv).read().splitlines()) >= 25


def test_run_really_killed_at_an_arbitrary_moment(tmp_path):
    """SIGKILL whenever: the wall-clock checkpoint cadence, a CSV with rows past the last checkpoint."""
    base = [RUN, EX("example_obstacle.cfg"), "--quiet", "--set", "nCells", "4000", "--set", "max_time", "400",
            "--set", "dump_interval", "2", "--set", "testing", "0"]
    a_csv, b_csv, ck = str(tmp_path / "a.csv"), str(tmp_path / "b.csv"), str(tmp_path / "b.ck")
    r = run(base + ["--set", "csv_filename", a_csv, "--final-checkpoint", str(tmp_path / "a.final")])
    assert r.returncode == 0, r.stderr
    p = subprocess.Popen(base + ["--set", "csv_filename", b_csv, "--checkpoint", ck, "--checkpoint-every", "0.05"],
                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    t0 = time.time()
    while not os.path.exists(ck) and p.poll() is None and time.time() - t0 < 120:
        time.sleep(0.005)
    time.sleep(0.12)
    killed = p.poll() is None
    p.send_signal(signal.SIGKILL)
    p.wait()
    assert os.path.exists(ck)
    r = run(base + ["--set", "csv_filename", b_csv, "--resume", ck, "--final-checkpoint", str(tmp_path / "b.final")])
    assert r.returncode == 0, r.stderr
    assert open(a_csv, "rb").read() == open(b_csv, "rb").read(), f"killed in mid-run: {killed}"
    assert state_part(tmp_path / "a.final") == state_part(tmp_path / "b.final")


def test_resume_from_the_csv_like_the_reference(tmp_path):
    """main.cpp:940-957: `cont` reads the last complete row of the run's own CSV and appends to it."""
    csv = str(tmp_path / "run.csv")
    base = [RUN, EX("example.cfg"), "--quiet", "--set", "csv_filename", csv, "--set", "testing", "1", "--set",
            "dump_interval", "0.5", "--set", "phase_std", "0"]
    assert run(base + ["--set", "max_time", "2"]).returncode == 0
    rows_before = open(csv).read().splitlines()
    assert run(base + ["--set", "max_time", "4", "--resume", csv]).returncode == 0
    rows = open(csv).read().splitlines()
    assert rows[:len(rows_before)] == rows_before and len(rows) > len(rows_before)
    def times(lines):   # (the dump writes header lines too; data rows start with the time)
        out = []
        for r in lines:
            try:
                out.append(float(r.split(",")[0]))
            except ValueError:
                pass
        return out
    t0, t1 = times(rows_before), times(rows[len(rows_before):])
    # (the resumed run dumps the row it restarted from again, as the reference's loop would)
    assert t1[0] == pytest.approx(t0[-1], abs=0.011) and t1[-1] >= 3.99 and t0[-1] <= 2.02


def test_resume_recognises_its_own_csv_however_it_is_spelled(tmp_path):
    """the same file named differently (./run.csv against run.csv, a symlink, an absolute path) is
    still the run's own CSV -- appended to, never reopened with "w+" -- and a resume from ANOTHER CSV refuses to
    truncate an existing csv_filename unless --overwrite-csv is given."""
    base = [RUN, EX("example.cfg"), "--quiet", "--set", "csv_filename", "run.csv", "--set", "testing", "1", "--set",
            "dump_interval", "0.5", "--set", "phase_std", "0"]
    assert run(base + ["--set", "max_time", "2"], cwd=tmp_path).returncode == 0
    before = open(tmp_path / "run.csv").read()
    os.symlink(tmp_path / "run.csv", tmp_path / "link.csv")
    for k, spelled in enumerate(["./run.csv", str(tmp_path / "run.csv"), "link.csv"]):
        assert run(base + ["--set", "max_time", str(3 + k), "--resume", spelled], cwd=tmp_path).returncode == 0
        now = open(tmp_path / "run.csv").read()
        assert now.startswith(before) and len(now) > len(before), spelled
        before = now
    # another CSV as the source: run.csv exists and is not it -> refused, untouched
    other = tmp_path / "other.csv"
    other.write_text(before)
    p = run(base + ["--set", "max_time", "7", "--resume", "other.csv"], cwd=tmp_path)
    assert p.returncode == 1 and "refusing to truncate" in p.stderr and open(tmp_path / "run.csv").read() == before
    p = run(base + ["--set", "max_time", "7", "--resume", "other.csv", "--overwrite-csv"], cwd=tmp_path)
    assert p.returncode == 0 and not open(tmp_path / "run.csv").read().startswith(before[:len(before) // 2])
    # ... and a csv_filename that does not exist yet is simply created
    p = run(base[:5] + ["fresh.csv"] + base[6:] + ["--set", "max_time", "7", "--resume", "other.csv"], cwd=tmp_path)
    assert p.returncode == 0 and os.path.getsize(tmp_path / "fresh.csv") > 0


def test_pipelined_ensemble_stopped_and_resumed_equals_the_uninterrupted_run(tmp_path):
    """A 16-member ensemble in sub-batches of 8 with checkpoints at every summary row: stopped after 1500 of its
    3000 steps (sub-batch 0 in mid-run, sub-batch 1 not started... both stop at the step cap), resumed: the same rows
    bit for bit, the same final states; resumed once more after everything has finished: rows straight from disk."""
    from helpers import assert_bit_equal
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example_obstacle.cfg")
    members = [f"seed\n{1000 + k}" for k in range(16)]
    common = {"max_time": "30", "dump_interval": "6", "time_to_dead": "7", "nDead": "40"}
    a = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=2, keep_final_states=True,
                                   checkpoint_dir=str(tmp_path / "a"))
    steps = a.run()
    assert steps in (3000, 3001) and a.rows.shape[1] == 7
    b = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=2, checkpoint_dir=str(tmp_path / "b"))
    assert b.run(1500) == 1500
    assert b.rows.shape[1] == 4          # t = 0, 0.01, 6, 12
    b.close()
    b2 = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=2, keep_final_states=True,
                                    checkpoint_dir=str(tmp_path / "b"), resume=True)
    assert b2.run() == steps
    assert np.array_equal(b2.rows.view(np.uint32), a.rows.view(np.uint32))
    sa, sb = a.final_states(), b2.final_states()
    for k in range(16):
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(sb[k][key], sa[k][key], f"member {k} {key}")
    assert b2.timings["placement_cpu_s"] < a.timings["placement_cpu_s"]      # restored, not placed again
    b2.close()
    b3 = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=2, checkpoint_dir=str(tmp_path / "b"),
                                    resume=True)
    assert b3.run() == steps and b3.timings["device_s"] == 0.0
    assert np.array_equal(b3.rows.view(np.uint32), a.rows.view(np.uint32))
    b3.close()
    a.close()
    # another decomposition must not pick the directory up
    with pytest.raises(RuntimeError):
        ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=4, host_threads=2, checkpoint_dir=str(tmp_path / "b"),
                                   resume=True)


def test_automatic_sub_batch_resumes_with_the_size_it_started_with(tmp_path):
    """sub_batch -1 follows the number of producer threads (pbEnsemblePipelineAutoSubBatch: 12-member sub-batches for
    3 threads and members this small, 8 for 2 threads); a sweep started with 3 threads and resumed with 2 keeps the
    12-member sub-batches of its checkpoint directory, and ends bit-identical to the uninterrupted run -- which here
    also steps its two sub-batches at the same time (lanes 2)."""
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example.cfg")
    members = [f"seed\n{3000 + k}" for k in range(20)]
    common = {"max_time": "18", "dump_interval": "6"}
    a = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=-1, host_threads=3)
    steps = a.run()
    assert a.timings["sub_batch"] == 12 and a.timings["sub_batches"] == 2 and a.timings["lanes"] == 2
    b = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=-1, host_threads=3, checkpoint_dir=str(tmp_path / "b"))
    assert b.run(700) == 700
    b.close()
    b2 = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=-1, host_threads=2, checkpoint_dir=str(tmp_path / "b"),
                                    resume=True)
    assert b2.run() == steps and b2.timings["sub_batch"] == 12
    assert np.array_equal(b2.rows.view(np.uint32), a.rows.view(np.uint32))
    b2.close()
    c = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=-1, host_threads=2)
    assert c.run(10) == 10 and c.timings["sub_batch"] == 8      # (what 2 threads choose on their own)
    c.close()
    a.close()


def test_cxx_ensemble_runner_killed_and_resumed(tmp_path):
    """bin/particlebot_ensemble --checkpoint DIR, SIGKILLed in mid-sweep, then --resume DIR: the gathered rows equal
    the uninterrupted sweep's bit for bit (world of one rank through the real RCCL calls)."""
    base = [ENS, EX("example_obstacle.cfg"), "--members", "12", "--sub-batch", "4", "--host-threads", "2", "--set",
            "nCells", "2500", "--set", "max_time", "240", "--set", "dump_interval", "6"]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29433")
    r = run(base + ["--out", str(tmp_path / "a.bin")], env=env)
    assert r.returncode == 0, r.stderr
    ck = tmp_path / "ck"
    p = subprocess.Popen(base + ["--checkpoint", str(ck), "--out", str(tmp_path / "b_killed.bin")], env=env,
                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    manifest = ck / "rank0of1" / "sub_000000.manifest"
    t0 = time.time()
    while not manifest.exists() and p.poll() is None and time.time() - t0 < 120:
        time.sleep(0.005)
    time.sleep(0.3)
    killed = p.poll() is None
    p.send_signal(signal.SIGKILL)
    p.wait()
    assert manifest.exists()
    r = run(base + ["--resume", str(ck), "--out", str(tmp_path / "b.bin")], env=env)
    assert r.returncode == 0, r.stderr
    a = np.fromfile(tmp_path / "a.bin", np.uint32)
    b = np.fromfile(tmp_path / "b.bin", np.uint32)
    assert a.size == b.size and a.size == 12 * 42 * 4 and np.array_equal(a, b), f"killed in mid-run: {killed}"
    assert '"resumed": true' in r.stdout


def test_payload_ensemble_with_xorwow_noise_resumes_exactly(tmp_path):
    """Object transport (payload mode, phase noise from the cuRAND-shaped XORWOW generator: per-bot generator states
    that must be rebuilt from seed, bot and draw count on resume) through the pipeline's checkpoints: stopped in
    mid-run, resumed, same rows and final states as the uninterrupted run."""
    from helpers import assert_bit_equal
    from particlerobotsimulations_amd import ensemble
    cfg = EX("example_object_transport.cfg")
    members = [f"seed\n{500 + k}" for k in range(6)]
    common = {"max_time": "40", "dump_interval": "6", "pb_rng": "curand", "phase_update_interval": "4"}
    a = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=3, host_threads=2, keep_final_states=True)
    steps = a.run()
    b = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=3, host_threads=2, checkpoint_dir=str(tmp_path / "ck"))
    assert b.run(2100) == 2100      # past five phase updates (five draws per bot) and three summary rows
    b.close()
    c = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=3, host_threads=2, keep_final_states=True,
                                   checkpoint_dir=str(tmp_path / "ck"), resume=True)
    assert c.run() == steps
    assert np.array_equal(c.rows.view(np.uint32), a.rows.view(np.uint32))
    for k, (sa, sc) in enumerate(zip(a.final_states(), c.final_states())):
        for key in ("pos", "vel", "rad"):
            assert_bit_equal(sc[key], sa[key], f"member {k} {key}")
    a.close(), c.close()


@pytest.mark.parametrize("cfg,extra", [("example_obstacle.cfg", ["--set", "max_time", "40"]),
                                       ("example_object_transport.cfg", ["--set", "max_time", "30", "--set", "phase_std", "0.4"]),
                                       ("example_dead_cells.cfg", ["--set", "max_time", "25", "--set", "nCells", "6000",
                                                                   "--set", "nDead", "900"])])
def test_ensemble_csv_dir_holds_the_reference_csv_of_every_member(tmp_path, cfg, extra, orc):
    """--csv-dir: member k's file is, byte for byte, the CSV the reference writes for that member run on its own with
    testing 0 (particlebot.cpp:303-367) -- its centroid columns are fp32 sums over the bots in order, which one lane per
    member reproduces on the device (pbSimCentroidSums), not the rows' accurately rounded mean.  Checked against
    particlebot_run (itself byte-identical to the oracle's dump: test_gpu_host.py) for three members, against the
    oracle's own dump for one, and in a split that steps two sub-batches at a time."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29451")
    common = ["--set", "dump_interval", "6", "--set", "testing", "0"] + extra
    d = tmp_path / "csv"
    r = run([ENS, EX(cfg), "--members", "7", "--seed0", "4100", "--sub-batch", "2", "--host-threads", "2", "--csv-dir", str(d)] + common,
            env=dict(env, PB_PIPELINE_LANES="2"))
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(d)) == [f"member_{k:06d}.csv" for k in range(7)]
    for k in (0, 3, 6):
        single = str(tmp_path / f"single_{k}.csv")
        r = run([RUN, EX(cfg), "--quiet", "--set", "seed", str(4100 + k), "--set", "csv_filename", single] + common)
        assert r.returncode == 0, r.stderr
        a, b = open(single, "rb").read(), open(d / f"member_{k:06d}.csv", "rb").read()
        assert a == b, (k, a[:200], b[:200])
        assert a.startswith(f"Seed, {4100 + k}\nTime,Centroid X, Centroid Y, Distance\n".encode()) and a.count(b"\n") >= 6
    # ... and the oracle's own dump of member 3
    kv = {}
    it = iter(common)
    for flag in it:
        name, value = next(it), next(it)
        kv[name] = float(value) if "." in value else int(value)
    from test_gpu_host import oracle_csv
    path = str(tmp_path / "oracle_3.csv")
    oracle_csv(orc, EX(cfg), path, seed=4103, **kv).close()
    assert open(path, "rb").read() == open(d / "member_000003.csv", "rb").read()
    # refused together with checkpoints
    r = run([ENS, EX(cfg), "--members", "2", "--csv-dir", str(d), "--checkpoint", str(tmp_path / "ck")] + common, env=env)
    assert r.returncode == 2 and "cannot be combined" in r.stderr
    # the Python front end (python -m particlerobotsimulations_amd.ensemble) drives the same pipeline
    if cfg == "example_obstacle.cfg":
        import sys
        d2 = tmp_path / "csv_py"
        r = run([sys.executable, "-m", "particlerobotsimulations_amd.ensemble", EX(cfg), "--members", "4", "--seed0", "4100",
                 "--sub-batch", "3", "--csv-dir", str(d2)] + common, env=dict(env, PYTHONPATH=ROOT), cwd=ROOT)
        assert r.returncode == 0, r.stderr
        for k in range(4):
            assert open(d2 / f"member_{k:06d}.csv", "rb").read() == open(d / f"member_{k:06d}.csv", "rb").read(), k


def test_csv_dir_never_writes_a_silently_shortened_file(tmp_path, capfd):
    """the member CSVs were written inside the `out && nrows < max_rows` branch, so a run with more dump
    rows than max_rows (4096 in the runner) stopped its files there and still returned success.  Now the run fails
    with a message when a row is due that the buffer cannot hold; with room it succeeds and the files are closed
    (checked) at the end of the run."""
    from particlerobotsimulations_amd import ensemble
    members = [f"seed\n{5200 + k}" for k in range(3)]
    common = {"max_time": "30", "dump_interval": "6", "testing": "0"}
    d = tmp_path / "short"
    d.mkdir()
    p = ensemble.PipelinedEnsemble(EX("example.cfg"), members, common, sub_batch=0, host_threads=2, max_rows=3,
                                   csv_dir=str(d))
    with pytest.raises(RuntimeError):
        p.run()
    p.close()
    # (refused BEFORE the first step -- the row count follows from dt, dump_interval and max_time -- not
    #  when row 4 falls due, possibly hours into the run)
    assert "nothing was stepped" in capfd.readouterr().err
    # the same without --csv-dir: rows past max_rows used to be dropped silently
    p = ensemble.PipelinedEnsemble(EX("example.cfg"), members, common, sub_batch=0, host_threads=2, max_rows=6)
    with pytest.raises(RuntimeError):
        p.run()
    p.close()
    assert "more than 6 summary rows" in capfd.readouterr().err
    p = ensemble.PipelinedEnsemble(EX("example.cfg"), members, common, sub_batch=0, host_threads=2, max_rows=7)
    assert p.run() > 0 and p.rows.shape[1] == 7          # rows at 0, 0.01, 6, 12, 18, 24, 30.0x: exactly what fits
    p.close()
    # a run bounded by max_steps needs only the rows of those steps (1 500 steps: t = 0, 0.01, 6, 12)
    p = ensemble.PipelinedEnsemble(EX("example.cfg"), members, dict(common, max_time="1e9"), sub_batch=0, host_threads=2,
                                   max_rows=4)
    assert p.run(1500) == 1500 and p.rows.shape[1] == 4
    p.close()
    d2 = tmp_path / "whole"
    d2.mkdir()
    p = ensemble.PipelinedEnsemble(EX("example.cfg"), members, common, sub_batch=0, host_threads=2, max_rows=16,
                                   csv_dir=str(d2))
    assert p.run() > 0
    text = open(d2 / "member_000001.csv").read()      # complete BEFORE close(): the run closed it
    p.close()
    assert text.startswith("Seed, 5201\n") and text.count("\n") == 2 + 7   # seed, header, rows at 0, 0.01 (the reference's gate passes both), 6 ... 30
