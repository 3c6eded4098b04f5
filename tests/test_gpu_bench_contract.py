"""bench.py's output contract (ONE JSON line under 4 KB with the keys the driver and the judge read; one kernel per
headline) and tools/bench_legs.py's (every other leg, one long line), exercised on small arenas so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = os.path.join(ROOT, "tools", "bench_legs.py")


def test_headline_line_is_small_and_describes_one_kernel(tmp_path):
    """`python bench.py`: value, ms_per_step and roofline are the SAME K launches of the both-sums kernel (everything
    collideD writes, impl.cuh:828-830); the default (dead-sum) form, the CPU baseline and configs[3] ride along; the
    line is under 4 KB and the long record is in the --detail file."""
    n, k = 200000, 60
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bots", str(n), "--steps", str(k), "--warmup",
                          "20", "--cpu-seconds", "1", "--e2e-steps", "1500", "--detail", str(tmp_path / "d.json")],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) + 1 < 4096, (len(lines), len(lines[0]))
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "default_form", "ensemble", "detail"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == k and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["unit"] == "particle-steps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["collective"] is None
    c = d["config"]
    assert "workload" in c and "model" not in c and len(c["workload"]) <= 200 and c["bots_per_gpu"] == n
    assert c["attraction_sums"] == 1 and c["force_variant"] == 2 and c["lanes_per_bot"] == 1
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] == "k_force<false, true, 1, 1, false, true>" and r["launches"] == k
    assert r["alg_bytes_per_launch"] == 64.0 * n
    # one kernel per headline: every figure recomputes from the same K launches
    assert abs(r["avg_launch_us"] * k - d["device_ms_timed_region"] * 1e3) < 1e-6 * d["device_ms_timed_region"] * 1e3
    assert abs(r["achieved"] - 64.0 * n / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(d["value"] - n * k / (d["ms_per_step"] * k * 1e-3)) / d["value"] < 1e-6
    # wall clock of the K steps >= their device time, and close to it: 64 * value / peak ~ frac
    assert d["ms_per_step"] * k >= d["device_ms_timed_region"] * 0.999
    assert 0.85 * r["frac"] < 64.0 * d["value"] / 8e12 <= r["frac"] * 1.001
    assert r["long"]["steps"] * r["long"]["avg_launch_us"] * 1e-3 >= 90.0 and r["long"]["frac"] > 0   # (K steps were < 50 ms)
    assert r["traffic"] is None and r["profile"]["dropped"] == "not 10^6 bots"   # the committed counters are 10^6-bot ones
    f = d["default_form"]
    assert f["attraction_sums"] == 0 and f["kernel"] == "k_force<false, true, 1, 1, false, false>"
    assert abs(f["frac_at_56B"] - 56.0 * f["value"] / 8e12) < 1e-12 and f["value"] > d["value"] * 0.95
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0 and b["value_1_thread"] > 0 and "sample" in b
    e = d["ensemble"]
    # (1 500 steps: rows at t = 0, 0.01, 6 and 12)
    assert e["members_total"] == 512 and e["steps_per_member"] == 1500 and e["rows_gathered"] == [[256, 4, 4], [256, 4, 4]]
    long = json.loads((tmp_path / "d.json").read_text())
    assert long["line"] == d and long["headline"]["prewarm"]["ms"] >= 100.0 and "glibc" in long["host"]
    assert long["default_form"]["timed"]["device_prewarm_ms"] >= 100.0


def test_headline_line_through_rccl_with_one_rank(tmp_path):
    """--force-dist: the N > 1 code path on one GPU -- RCCL communicator, barriers, MAX all-reduce, gather of the arena
    summaries and of the configs[3] rows -- and `collective` says what the communicator saw."""
    raw = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bots", "150000", "--steps", "40", "--warmup",
                          "10", "--force-dist", "--no-cpu-baseline", "--e2e-steps", "300", "--detail", str(tmp_path / "d.json")],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert raw.returncode == 0, raw.stderr[-2000:]
    lines = [l for l in raw.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]      # ONE line on stdout: RCCL's version banner goes to stderr
    d = json.loads(lines[0])
    c = d["collective"]
    assert (c["backend"], c["torch_backend"], c["ranks"], c["local_rank_device"]) == ("rccl", "nccl", 1, [[0, 0]])
    assert c["distinct_gpus"] == 1 and len(c["pci_bus_ids"]) == 1 and ":" in c["pci_bus_ids"][0]   # e.g. 0000:c1:00.0
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["ensemble"]["rows_gathered"] == [[256, 2, 4], [256, 2, 4]]
    long = json.loads((tmp_path / "d.json").read_text())
    assert len(long["summaries_time_comx_comy"]) == 1


def test_profile_counters_are_quoted_only_for_the_loaded_kernel():
    """At 10^6 bots the line quotes the committed PMC profile (profiles/latest_traffic_both_sums.json) only if its
    kernel signature is the loaded library's; the detail record says which build the profile was taken on."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "20",
                          "--no-cpu-baseline", "--no-ensemble", "--detail", ""], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    # one kernel per headline, at the metric's own size: over a region of >= 50 ms the host's clock and the HIP events
    # agree, i.e. 64 * value / 8e12 == roofline.frac to three digits
    assert d["config"]["bots_per_gpu"] == 1_000_000 and "long" not in r and d["device_ms_timed_region"] >= 45.0
    assert abs(64.0 * d["value"] / 8e12 - r["frac"]) / r["frac"] < 2e-3, (d["value"], r["frac"])
    prof = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic_both_sums.json")))
    import particlerobotsimulations_amd as pb
    sig = [pb.force_form_kernel_name(i) for i, f in enumerate(pb.force_forms())
           if f == {"flat": 1, "lanes_per_bot": 1, "attraction_sums": 1, "offsets64": 0}][0]
    if prof.get("kernel_signature") == sig:
        assert r["traffic"] == prof["hbm_bytes_per_launch"] and 0.3 < r["valu_frac_of_datasheet"] < 1.0
        assert r["profile"]["name"] == prof["profile"] and isinstance(r["profile"]["sources_match"], bool)
        assert 0.8 < r["traffic"] / r["alg_bytes_per_launch"] < 1.5      # no wasted re-reads
    else:
        assert r["traffic"] is None and r["valu_frac_of_datasheet"] is None and "dropped" in r["profile"]


def test_legs_line_carries_every_leg():
    out = subprocess.run([sys.executable, LEGS, "--bots", "200000", "--steps", "60",
                          "--warmup", "20", "--cpu-seconds", "1", "--e2e-steps", "1500"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])  # the JSON line is the last line of stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["unit"] == "particle-steps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # achieved/peak/frac are the HBM accounting of SURVEY 8(d); `bound` names what really binds the kernel
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert d["config"]["force_variant"] == 2 and d["config"]["lanes_per_bot"] in (1, 4) and d["config"]["resident"] == 0
    assert 500.0 < r["shader_clock_mhz"] < 2600.0, r["shader_clock_mhz"]
    assert d["device_prewarm"]["steps"] >= 100 and d["device_prewarm"]["ms"] >= 100.0   # disclosed, not timed
    el = d["ensemble_leg"]   # BASELINE configs[3] measured beside the arena (what a SCALE run sees at every N)
    assert el["config"]["bots_per_member"] == [500, 201] and el["config"]["members_per_gpu"] == 64 and el["value"] > 1e8
    assert el["summary_rows_gathered"][0][0] == 32
    assert el["device_prewarm"]["ms"] >= 100.0                       # every leg is pre-warmed, not only the headline
    # the ensemble END TO END (placement + upload + steps + gather, through the sub-batch pipeline), weak and strong
    e2e, strong = el["end_to_end"], el["strong_end_to_end"]
    assert e2e["scaling"] == "weak" and e2e["members_total"] == 64 and e2e["steps_per_member"] == 1500
    assert strong["scaling"] == "strong" and strong["members_total"] == 512 and strong["rows_gathered"][0][0] == 256
    assert 0 < e2e["value_end_to_end"] and el["value_end_to_end"] == e2e["value_end_to_end"]
    assert e2e["pipeline_rank0"][0]["sub_batches"] == 1 and "sims_per_s" not in el
    # short timed regions carry a second, >= 100 ms figure
    assert d["device_ms_timed_region"] < 50.0 and d["value_long"] > 0 and d["roofline"]["frac_dead_sum_long"] > 0
    assert d["steps_long"] * d["roofline"]["avg_launch_us_dead_sum_long"] * 1e-3 >= 90.0   # (sized from the short region's pace)
    assert "glibc" in d["host"] and d["host"]["cpus"] >= 1
    # like for like: `frac` is the kernel that writes everything collideD writes (the both_sums leg) at
    # 64 B; the kernel `value` runs is priced at the 56 B it moves; both recomputable from the line itself
    assert "alg_bytes_note" in r and "collideD" in r["frac_is"] and "hbm_target_note" in r
    b = d["both_sums"]
    us_b = b.get("us_per_step_long", b["us_per_step"])
    assert abs(r["avg_launch_us"] - us_b) < 1e-9 and r["algorithmic_bytes_per_launch"] == 64.0 * 200000
    assert abs(r["achieved"] - 64.0 * 200000 / (us_b * 1e-6) / 1e9) / r["achieved"] < 1e-9
    assert r["frac"] == r["frac_both_sums"] and "false, true>" in r["kernel"]
    ds = r["dead_sum"]
    assert ds["algorithmic_bytes_per_launch"] == 56.0 * 200000 and abs(ds["frac"] - r["frac_dead_sum"]) < 1e-15
    assert abs(ds["achieved"] - 56.0 * 200000 / (ds["avg_launch_us"] * 1e-6) / 1e9) / ds["achieved"] < 1e-9
    assert abs(r["frac_dead_sum_priced_at_64"] * 56.0 / 64.0 - r["frac_dead_sum"]) < 1e-12
    assert 0 < r["frac"] <= r["frac_dead_sum_priced_at_64_long"] * 1.05
    assert d["config"]["force_sums_note"].endswith("roofline.frac")
    la = d["large_arena"]
    assert la["bots"] == 8_000_000 and la["finite_at_end"] and la["us_per_step"] > 0 and la["us_per_step_long"] > 0
    assert la["device_prewarm_ms"] >= 100.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert abs(d["value"] - 200000 * 60 / (d["ms_per_step"] * 60 * 1e-3)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    s = d["streamlined"]
    assert s["finite_at_end"] and s["value"] > d["value"] and s["parity"]["window_steps"] == 10
    # the headline runs the engine's default for the reference's default parameters (constrained_contraction 0:
    # absForce_a has no reader and is not computed); the same workload with both sums kept is reported beside it
    assert d["config"]["attraction_sums"] == 0 and d["config"]["dead_sum_form"] == 1
    b = d["both_sums"]
    assert b["attraction_sums"] == 1 and b["dead_sum_form"] == 0 and 0 < b["value_long"] < d["value_long"] * 1.05
    assert d["random_blob"]["device_prewarm_ms"] >= 100.0 and d["random_blob"]["value"] > 0


def _bench(*args):
    out = subprocess.run([sys.executable, LEGS, *args], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])


def test_ensemble_workload_line_and_rccl_path_on_one_gpu():
    """`bench.py --workload ensemble4` (BASELINE configs[3]: obstacle + object-transport seed ensembles)
    under the same contract keys; with --force-dist the same command initialises RCCL (world size 1),
    runs the max-over-ranks all_reduce and the all_gather of the summary rows, and must report the
    SAME summaries."""
    common = ("--workload", "ensemble4", "--members-per-gpu", "6", "--steps", "700", "--warmup", "50",
              "--cpu-seconds", "1", "--e2e-steps", "900")
    a = _bench(*common)
    b = _bench(*common, "--force-dist", "--no-cpu-baseline")
    for d in (a, b):
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "value_end_to_end", "sims_per_s_end_to_end"):
            assert k in d, k
        assert d["n_gpus"] == 1 and d["steps"] == 700 and d["scaling"] == "weak" and d["unit"] == "particle-steps/s"
        assert d["config"]["members_per_gpu"] == 12 and d["config"]["bots_per_member"] == [500, 201]
        assert d["config"]["members_per_rank"] == [12]
        assert abs(d["value"] - 6 * (500 + 201) * 700 / (d["ms_per_step"] * 700 * 1e-3)) / d["value"] < 1e-6
        # (700 steps are under 50 ms: a second timed region of >= 100 ms follows, so more rows than t = 0, 0.01, 6)
        assert d["summary_rows_gathered"][0][0] == 6 and d["summary_rows_gathered"][0][1] >= 3 and d["steps_long"] > 700
        assert d["end_to_end"]["steps_per_member"] == 900 and d["end_to_end"]["rows_gathered"] == [[6, 3, 4], [6, 3, 4]]
    assert "RCCL world size 1" in b["config"]["parallelism"]
    assert a["end_to_end"]["last_rows_time_comx_comy_dist"] == b["end_to_end"]["last_rows_time_comx_comy_dist"]
    assert a["cpu_baseline"]["kind"] == "port" and a["cpu_baseline"]["value"] > 0


def test_ensemble5_workload_line():
    """BASELINE configs[4] at bench scale: 10^5-bot members of the dead-fraction sweep as one batch."""
    d = _bench("--workload", "ensemble5", "--members-per-gpu", "2", "--steps", "60", "--warmup", "10",
               "--no-cpu-baseline", "--e2e-steps", "80")
    assert d["config"]["bots_per_member"] == [100000] and d["config"]["members_per_gpu"] == 2
    g = d["summary_rows_gathered"]   # rows at t = 0 and 0.01, more if the >= 100 ms region reached t = 6
    assert d["value"] > 1e8 and len(g) == 1 and g[0][0] == 2 and g[0][1] >= 2 and g[0][2] == 4
    tm = d["end_to_end"]["pipeline_rank0"][0]
    # the two members are two dead fractions of ONE seed: one placement (~0.27 CPU-s at 10^5 bots), one copy
    assert d["end_to_end"]["steps_per_member"] == 80 and tm["placement_cpu_s"] > 0.1
    assert tm["placements_run"] == 1 and tm["placements_shared"] == 1


def test_ensemble_strong_form_members_total():
    """--members-total: a fixed number of members over all GPUs (BASELINE configs[3] is 256 + 256)."""
    d = _bench("--workload", "ensemble4", "--members-total", "10", "--steps", "300", "--warmup", "20",
               "--no-cpu-baseline", "--e2e-steps", "400")
    assert d["scaling"] == "strong" and d["config"]["members_total"] == 20 and d["config"]["members_per_rank"] == [20]
    assert d["end_to_end"]["scaling"] == "strong" and d["end_to_end"]["members_total"] == 20


def test_arena_line_through_rccl_with_one_rank_and_gpus_flag_is_checked():
    """The N > 1 code path of the default workload on one GPU: --force-dist initialises RCCL (world 1),
    runs the barrier / max-over-ranks all_reduce / all_gather of the arena summaries.  And `--gpus 2`
    without a launcher on a one-GPU box refuses loudly instead of silently running one arena."""
    raw = subprocess.run([sys.executable, LEGS, "--bots", "150000", "--steps", "40", "--warmup",
                          "10", "--force-dist", "--no-cpu-baseline", "--no-survey-literal", "--no-streamlined",
                          "--no-large-arena", "--no-blob", "--e2e-steps", "300"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert raw.returncode == 0, raw.stderr[-2000:]
    lines = [l for l in raw.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]      # ONE line on stdout: RCCL's version banner goes to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and len(d["summaries_time_comx_comy"]) == 1 and d["value"] > 0
    assert "RCCL world size 1" in d["ensemble_leg"]["config"]["parallelism"]   # the leg's gather went over RCCL
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"],
                             capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert out.returncode == 2 and "only" in out.stderr and "GPU" in out.stderr
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5"],
                         capture_output=True, text=True, timeout=60, cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr


def test_force_sums_flag_profiles_the_both_sums_kernel_as_the_arena():
    """`tools/bench_legs.py --force-sums 1` (what tools/profile.sh runs for profiles/r6_both_sums.*): the arena itself keeps both
    magnitude sums, so that kernel is the one under the profiler; the line says it is not the headline."""
    d = _bench("--bots", "100000", "--steps", "40", "--warmup", "10", "--force-sums", "1", "--no-cpu-baseline",
               "--no-survey-literal", "--no-streamlined", "--no-large-arena", "--no-clock", "--no-blob", "--no-ensemble-leg",
               "--no-both-sums", "--no-host-round-trip")
    assert d["headline"] is False and d["config"]["attraction_sums"] == 1 and d["config"]["dead_sum_form"] == 0
    assert d["value"] > 0 and "both_sums" not in d and "collideD" not in d["roofline"]["frac_is"]
