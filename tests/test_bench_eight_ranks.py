"""EIGHT ranks, before the driver's SCALE run starts them ("eight ranks have never run, even dry" --
the CPU tests stopped at world size 2 and 3 while every scaling statement of DESIGN.md section 7 is about 8).  What
depends on the rank count: the producer pool of a rank (usable CPUs // ranks of the node: 2 threads = 1 producer + the
device thread under the driver's 16-CPU quota), member k -> rank k mod 8 with BASELINE configs[3]'s 256 + 256 members,
the gather of 8 blocks of summary rows, and the C++ runner's TCP id exchange with seven clients under one deadline.
No GPU: `--dry-run-device` (gloo; members placed for real, rows made from the checksum of the placed state, so a
gathered row identifies its member) and `particlebot_ensemble --rendezvous-test`."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from test_bench_multirank import check_headline_line, one_json_line, run_bench, run_legs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_ensemble")


def fake_quota(tmp_path, cpus=16):
    """the driver's box: a cgroup quota of 16 CPUs on a 128-thread host (test hooks of pbHostGetResources)"""
    root = tmp_path / "cg"
    root.mkdir(exist_ok=True)
    (root / "cpu.max").write_text(f"{cpus * 100000} 100000\n")
    proc = tmp_path / "proc_self_cgroup"
    proc.write_text("0::/\n")
    return {"PB_CGROUP_ROOT": str(root), "PB_PROC_SELF_CGROUP": str(proc)}


@pytest.fixture(scope="module")
def eight_rank_configs3(tmp_path_factory):
    env = fake_quota(tmp_path_factory.mktemp("q"))
    t0 = time.time()
    d = one_json_line(run_legs("--gpus", "8", "--workload", "ensemble4", "--members-total", "256", "--e2e-steps", "100",
                               "--steps", "20", "--dry-run-device", env=env, timeout=900))
    d["_wall"] = time.time() - t0
    return d


def test_eight_rank_configs3_line(eight_rank_configs3):
    """BASELINE configs[3] as written (256 + 256 members in all) on eight ranks: ONE JSON line from rank 0."""
    d = eight_rank_configs3
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["metric"].startswith("particle-steps/sec at 10^6 bots") and d["unit"] == "particle-steps/s"
    cfg = d["config"]
    # member k -> rank k mod 8: 32 members per rank and per .cfg, two .cfgs
    assert cfg["members_per_rank"] == [64] * 8 and cfg["members_total"] == 512 and cfg["bots_per_member"] == [500, 201]
    assert "world size 8" in cfg["parallelism"]
    assert d["summary_rows_gathered"] == [[256, 3, 4], [256, 3, 4]]
    e2e = d["end_to_end"]
    assert e2e["n_gpus"] == 8 and e2e["members_total"] == 512 and e2e["rows_gathered"] == [[256, 3, 4], [256, 3, 4]]
    assert e2e["steps_per_member"] == 100
    # the pool of a rank: its eighth of min(hardware, affinity, the 16-CPU quota); on the driver's box that is 2
    h = d["host"]
    assert h["ranks_per_node"] == 8 and h["cgroup_cpu_quota"] == 16.0
    assert h["usable_cpus"] == min(16, h["cpus"]) and h["host_threads"] == max(1, h["usable_cpus"] // 8)
    assert "8 rank(s) per node" in h["host_threads_rule"]
    assert d["_wall"] < 180


def test_eight_rank_gather_puts_every_member_in_its_place(eight_rank_configs3, tmp_path):
    """The same 256 + 256 members on ONE rank: the rows rank 0 assembled from eight blocks are the one-rank rows,
    member for member."""
    one = one_json_line(run_legs("--gpus", "1", "--workload", "ensemble4", "--members-total", "256", "--e2e-steps", "100",
                                 "--steps", "20", "--dry-run-device", env=fake_quota(tmp_path), timeout=900))
    assert one["n_gpus"] == 1 and one["config"]["members_per_rank"] == [512]
    for key in ("last_rows_time_comx_comy_dist",):
        a, b = np.array(eight_rank_configs3["end_to_end"][key]), np.array(one["end_to_end"][key])
        assert a.shape == b.shape and np.array_equal(a, b)
    a = np.array(eight_rank_configs3["summaries_last_row_time_comx_comy_dist"])
    b = np.array(one["summaries_last_row_time_comx_comy_dist"])
    assert np.array_equal(a, b)
    # all rows of all members (not just the line's sample): the checksum rows differ member to member
    if "all_last_rows_sha1" in one:
        assert eight_rank_configs3["all_last_rows_sha1"] == one["all_last_rows_sha1"]


def test_eight_rank_configs4_line(tmp_path):
    """--workload ensemble5 on eight ranks, two 10^5-bot members per GPU (placed for real by one producer per rank)."""
    t0 = time.time()
    d = one_json_line(run_legs("--gpus", "8", "--workload", "ensemble5", "--members-per-gpu", "2", "--e2e-steps", "20",
                               "--steps", "10", "--dry-run-device", env=fake_quota(tmp_path), timeout=1200))
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["config"]["members_per_rank"] == [2] * 8 and d["config"]["bots_per_member"] == [100000]
    assert d["end_to_end"]["rows_gathered"] == [[16, 3, 4]] and d["end_to_end"]["members_total"] == 16
    assert "end_to_end_fastblob" not in d
    tm = d["end_to_end"]["pipeline_rank0"][0]
    # rank 0's pipeline: its share of the host minus the thread that drives the device, at most one per member
    assert tm["host_threads"] == min(2, max(1, d["host"]["host_threads"] - 1))
    assert time.time() - t0 < 180


def test_eight_rank_headline_line_is_small_and_names_eight_devices(tmp_path):
    """The driver's SCALE form of bench.py on eight ranks: the line obeys the 4 KB limit, the process group reports
    eight ranks, and every rank sits on its own LOCAL_RANK."""
    p = run_bench("--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run-device", "--e2e-steps", "50", "--detail",
                  str(tmp_path / "d.json"), env=fake_quota(tmp_path), timeout=900)
    d = check_headline_line(p, 8)
    c = d["collective"]
    assert c["ranks"] == 8 and sorted(x[0] for x in c["local_rank_device"]) == list(range(8))
    assert d["ensemble"]["members_total"] == 512 and d["ensemble"]["rows_gathered"] == [[256, 3, 4], [256, 3, 4]]
    long = json.loads((tmp_path / "d.json").read_text())
    assert len(long["summaries_time_comx_comy"]) == 8


def _id_exchange_eight_ranks_once():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = lambda r: dict(os.environ, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                         PB_RENDEZVOUS_PORT=str(port), PB_LAUNCH_TOKEN="eight-ranks")
    procs = [subprocess.Popen([EXE, "--rendezvous-test"], env=env(r), stdout=subprocess.PIPE, text=True)
             for r in (7, 3, 1, 5)]
    time.sleep(0.3)
    procs.append(subprocess.Popen([EXE, "--rendezvous-test"], env=env(0), stdout=subprocess.PIPE, text=True))
    time.sleep(0.2)
    try:
        c = socket.create_connection(("127.0.0.1", port), timeout=5)
        c.sendall(b"PBID" + (2).to_bytes(4, "little") + bytes(8))   # claims to be rank 2 of another launch
        time.sleep(0.2)
        c.close()
    except OSError:
        pass   # (rank 0 not listening yet on a loaded machine: the exchange itself is still checked below)
    procs += [subprocess.Popen([EXE, "--rendezvous-test"], env=env(r), stdout=subprocess.PIPE, text=True)
              for r in (2, 4, 6)]
    outs = [p.communicate(timeout=90)[0] for p in procs]
    return all(p.returncode == 0 for p in procs), outs


def test_cxx_runner_id_exchange_world_size_8_with_a_stray():
    """bin/particlebot_ensemble's rendezvous with SEVEN clients under one deadline, started before rank 0 serves, and
    one stray connection (right magic, wrong launch token) in between.  (The port is found by bind-and-close, which
    another process of a busy CI box may win before rank 0 binds it: one retry on a fresh port.)"""
    if not os.path.exists(EXE):
        pytest.skip("runner not built")
    ok, outs = _id_exchange_eight_ranks_once()
    if not ok:
        ok, outs = _id_exchange_eight_ranks_once()
    assert ok, outs
    assert sorted(o.strip() for o in outs) == [f"rendezvous-test rank {r} of 8: ok" for r in range(8)]
