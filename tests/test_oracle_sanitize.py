"""The oracle (plain-C restatement) under AddressSanitizer + UBSan: `make -C oracle sanitize`
builds oracle/kmer_oracle.c with oracle/oracle_selftest.c and runs it over the edge cases
(sanitizers on the CPU build only).  The self-test also checks that the
multi-threaded driver behind bench.py's all-cores cpu_baseline gives the sequential counters."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_selftest_under_asan_ubsan():
    if not shutil.which("gcc") or not shutil.which("make"):
        pytest.skip("no C toolchain")
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors="replace")
    if p.returncode != 0 and ("cannot find -lasan" in out or "libasan" in out and "No such file" in out):
        pytest.skip("this gcc has no sanitizer runtime")
    assert p.returncode == 0, out[-3000:]
    assert "oracle selftest ok" in out
