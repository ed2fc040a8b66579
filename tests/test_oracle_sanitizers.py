"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile, target asan): every entry point on
small inputs — three integrators forward and backward, path traces, environment light, uvgrad, light switching, ray
queries, sampler dump.  (GPU sanitizers are not available on the MI355X pool; the checker at least is clean.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "oracle_under_asan.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
