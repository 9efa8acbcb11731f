"""AddressSanitizer + UBSan over the CPU-side native code (the product's host geometry through a fuzz
harness, every oracle routine once): tools/sanitize/run.sh.  GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_geometry_and_oracle_are_clean_under_asan_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    res = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "run.sh")], capture_output=True, text=True, timeout=600)
    if res.returncode != 0 and ("cannot find -lasan" in res.stderr or "libasan" in res.stderr and "No such file" in res.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "host geometry:" in res.stdout and "oracle: every routine clean" in res.stdout
    assert "ordered halves:" in res.stdout and "clean" in res.stdout
