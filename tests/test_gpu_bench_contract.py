"""bench.py's output contract (one JSON line, the keys the driver and the judge read), exercised on
a small arena so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bots", "200000", "--steps", "60",
                          "--warmup", "20", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900,
                         cwd=ROOT)
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
    la = d["large_arena"]
    assert la["bots"] == 8_000_000 and la["finite_at_end"] and la["us_per_step"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert abs(d["value"] - 200000 * 60 / (d["ms_per_step"] * 60 * 1e-3)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    s = d["streamlined"]
    assert s["finite_at_end"] and s["value"] > d["value"] and s["parity"]["window_steps"] == 10
