"""The oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer: every example, all three
light_shadow modes, a dead-bot draw, 1300 steps with frequent re-sorts, a CSV dump.  A checker with
memory errors or undefined behaviour would be a poor checker."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "odrv")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
           "-fopenmp", f"-I{ROOT}/oracle", f"{ROOT}/tests/sanitize_oracle_driver.c", f"{ROOT}/oracle/pb_oracle.c",
           "-lm", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "asan" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2")
    run = subprocess.run([exe, ROOT], capture_output=True, text=True, cwd=tmp_path, env=env, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "oracle sanitize done" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-2000:]
