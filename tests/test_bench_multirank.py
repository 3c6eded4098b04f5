"""The N > 1 path of bench.py (the headline line) and tools/bench_legs.py (the ensemble workloads), run for real with two ranks before the driver's SCALE run does it (VERDICT round 3:
"has only ever executed with world size 1").  No GPU here: `--dry-run-device` puts gloo under the same
choreography -- spawn_ranks, the rendezvous, the opening/closing barriers, the MAX all-reduce of the wall time, the
gather of the members' summary rows, ONE JSON line from rank 0 on stdout -- and replaces the device by stand-ins
(the arena by a counter; ensemble members are placed for real by the producer pool and their rows made from the
checksum of the placed state, so a gathered row identifies its member).  The numbers of a dry-run line mean nothing;
its structure is what the driver will parse."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
LEGS = os.path.join(ROOT, "tools", "bench_legs.py")
MAX_LINE_BYTES = 4096   # bench.py's line must fit the driver's 8 KB stdout tail twice over


def run_legs(*args, **kw):
    return run_bench(*args, script=LEGS, **kw)


def run_bench(*args, env=None, timeout=600, script=BENCH):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, script, *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    return p


def one_json_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry ONE line, got {len(lines)}: {[l[:80] for l in lines]}"
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def two_rank_ensemble():
    return one_json_line(run_legs("--gpus", "2", "--workload", "ensemble4", "--e2e-steps", "200", "--steps", "50",
                                  "--dry-run-device"))


def test_two_rank_ensemble_line(two_rank_ensemble):
    d = two_rank_ensemble
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 50
    assert d["metric"].startswith("particle-steps/sec at 10^6 bots") and d["unit"] == "particle-steps/s"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32"
    cfg = d["config"]
    # member k -> rank k mod 2; 32 members per GPU and per .cfg, two .cfgs
    assert cfg["members_per_rank"] == [64, 64] and cfg["members_total"] == 128 and cfg["bots_per_member"] == [500, 201]
    assert "world size 2" in cfg["parallelism"]
    assert d["summary_rows_gathered"] == [[64, 3, 4], [64, 3, 4]]
    e2e = d["end_to_end"]
    assert e2e["n_gpus"] == 2 and e2e["members_total"] == 128 and e2e["rows_gathered"] == [[64, 3, 4], [64, 3, 4]]
    assert e2e["steps_per_member"] == 200
    # each rank sized its producer pool from ITS share of the host (LOCAL_WORLD_SIZE = 2 under the launcher)
    assert d["host"]["ranks_per_node"] == 2 and d["host"]["host_threads"] == max(1, d["host"]["usable_cpus"] // 2)


def test_gather_puts_every_member_in_its_place(two_rank_ensemble):
    """The same 64 members per .cfg on ONE rank (no process group): the rows rank 0 assembled from two ranks are
    the one-rank rows, member for member (a row is a function of the member's placed state)."""
    one = one_json_line(run_legs("--gpus", "1", "--workload", "ensemble4", "--members-per-gpu", "64", "--e2e-steps", "200",
                                 "--steps", "50", "--dry-run-device"))
    assert one["n_gpus"] == 1 and one["config"]["members_per_rank"] == [128]
    a = np.array(two_rank_ensemble["end_to_end"]["last_rows_time_comx_comy_dist"])
    b = np.array(one["end_to_end"]["last_rows_time_comx_comy_dist"])
    assert a.shape == b.shape == (2, 4, 4) and np.array_equal(a, b)
    a = np.array(two_rank_ensemble["summaries_last_row_time_comx_comy_dist"])
    b = np.array(one["summaries_last_row_time_comx_comy_dist"])
    assert np.array_equal(a, b) and len({tuple(r) for r in a.reshape(-1, 4)}) == 8  # eight different members


def check_headline_line(p, world):
    """bench.py's line: ONE line under 4 KB carrying the contract keys, one kernel per headline, the process group's
    own rank count, the configs[3] run; the long record goes to the --detail file."""
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) + 1 < MAX_LINE_BYTES, (len(lines), len(lines[0]))
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "default_form", "collective", "detail"):
        assert k in d, k
    assert d["dry_run"] is True and d["n_gpus"] == world and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["bots_per_gpu"] == 1_000_000 and d["config"]["attraction_sums"] == 1
    assert "model" not in d["config"] and len(d["config"]["workload"]) <= 200
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_us", "traffic", "valu_frac_of_datasheet"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["default_form"]["attraction_sums"] == 0 and "frac_at_56B" in d["default_form"]
    return d


def test_two_rank_arena_line(tmp_path):
    detail = tmp_path / "detail.json"
    p = run_bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run-device", "--e2e-steps", "100",
                  "--detail", str(detail))
    d = check_headline_line(p, 2)
    assert d["steps"] == 20 and d["warmup"] == 5 and "2 independent arenas" in d["config"]["parallelism"]
    # what the process group itself saw: two ranks, each on its own LOCAL_RANK
    c = d["collective"]
    assert c["backend"] == "gloo" and c["ranks"] == 2 and [x[0] for x in c["local_rank_device"]] == [0, 1]
    e = d["ensemble"]   # BASELINE configs[3] as written, strong form: 256 + 256 members over the two ranks
    assert e["members_total"] == 512 and e["scaling"] == "strong" and e["rows_gathered"] == [[256, 3, 4], [256, 3, 4]]
    assert e["steps_per_member"] == 100
    # rank-0-only, one-GPU-only legs stay out of a multi-rank line
    assert "cpu_baseline" not in d
    long = json.loads(detail.read_text())
    assert len(long["summaries_time_comx_comy"]) == 2 and long["line"] == d   # one summary per rank, gathered
    assert long["ensemble"]["n_gpus"] == 2 and "host" in long


def test_one_rank_line_has_no_process_group():
    d = check_headline_line(run_bench("--steps", "20", "--warmup", "5", "--dry-run-device", "--no-ensemble",
                                      "--no-cpu-baseline", "--detail", ""), 1)
    assert d["collective"] is None and "ensemble" not in d and d["detail"] is None
    assert d["config"]["parallelism"] == "single arena"


def test_emit_writes_every_byte_of_a_long_line():
    """benchkit.write_all loops over short writes: a pipe takes a 1 MB line in pieces."""
    code = ("import sys; sys.path.insert(0, %r); import benchkit as K; K.emit({'x': 'y' * 1000000})"
            % os.path.join(ROOT, "tools"))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, timeout=60)
    assert p.returncode == 0 and len(p.stdout) == len('{"x": ""}') + 1000000 + 1 and p.stdout.endswith(b'"}\n')


def test_launcher_started_ranks_and_strong_form():
    """The driver's own form: `python -m torch.distributed.run ... bench.py --gpus 2`; strong scaling by --members-total."""
    e = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), LEGS, "--gpus", "2", "--workload", "ensemble4",
           "--members-total", "10", "--e2e-steps", "50", "--steps", "20", "--dry-run-device"]
    p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    d = one_json_line(p)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["members_per_rank"] == [10, 10]
    assert d["end_to_end"]["rows_gathered"] == [[10, 3, 4], [10, 3, 4]]


def test_wrong_world_size_is_refused():
    p = run_bench("--gpus", "2", "--dry-run-device", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=3" in p.stderr


def test_a_rank_that_never_arrives_ends_the_run_with_code_2():
    """Rank 0 of a two-rank job whose rank 1 never starts: the rendezvous deadline (--rendezvous-timeout) ends the run
    with exit code 2 and a message instead of hanging."""
    port = str(29900 + os.getpid() % 90)
    t0 = time.time()
    p = run_bench("--gpus", "2", "--dry-run-device", "--rendezvous-timeout", "3",
                  env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port},
                  timeout=120)
    assert p.returncode == 2, (p.returncode, p.stderr[-2000:])
    assert "rendezvous failed" in p.stderr and time.time() - t0 < 60
    assert p.stdout.strip() == ""


def test_two_rank_ensemble5_line_runs_the_host_bound_broadcast():
    """--workload ensemble5 on two ranks (two 10^5-bot members per GPU, placed for real: a few CPU-seconds): the rank-0
    verdict `host-bound?` is broadcast to every rank before the optional fastblob re-run; under the dry-run device it
    is always `no`, but the collective itself must work."""
    d = one_json_line(run_legs("--gpus", "2", "--workload", "ensemble5", "--members-per-gpu", "2", "--e2e-steps", "20",
                               "--steps", "10", "--dry-run-device", timeout=900))
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["config"]["members_per_rank"] == [2, 2]
    assert d["config"]["bots_per_member"] == [100000] and d["end_to_end"]["rows_gathered"] == [[4, 3, 4]]
    assert "end_to_end_fastblob" not in d
