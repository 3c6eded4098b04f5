"""The host-side C++ (class Particlebot in its HostOnly engine, .cfg loader, C wrappers, the ensemble pipeline's producer
pool with SHARED PLACEMENTS -- round 6's new concurrent code: groups claimed under a mutex, look-ahead placement, waits
on a condition variable, copies read without the lock) under AddressSanitizer + UndefinedBehaviorSanitizer and under
ThreadSanitizer: tools/sanitize/run.sh, no GPU needed (GPU sanitizers are not available on the pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("mode", ["address,undefined", "thread"])
def test_host_code_is_clean_under_sanitizers(tmp_path, mode):
    args = ["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh")] + (["thread"] if mode == "thread" else [])
    p = subprocess.run(args, capture_output=True, text=True, timeout=900, env=dict(os.environ, TMPDIR=str(tmp_path)))
    out = p.stdout + p.stderr
    if "cannot find -lasan" in out or "cannot find -ltsan" in out or "cannot find -lubsan" in out:
        pytest.skip("sanitizer runtime not installed")
    assert "asan driver done" in out, out[-3000:]
    for bad in ("AddressSanitizer", "ThreadSanitizer", "runtime error", "LeakSanitizer", "DIFFER"):
        assert bad not in out, out[-3000:]
    # three seeds x four dead fractions: three placements however many producers there are, the same members every time
    assert out.count("3 placements") == 3 and out.count("checksums equal") == 3
