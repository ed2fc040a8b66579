"""The host side of libzdr_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer: zdr_api.cpp compiled by hipcc with
-fsanitize=address,undefined -fno-gpu-sanitize (host code only; the kernels' object file is the shipped one) and driven
through its GPU-free entry points — BVH builder, quad merge, plane records, argument checks.  (GPU sanitizers are not
available on the MI355X pool.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_builders_are_clean_under_asan_and_ubsan(tmp_path):
    from zdr_amd import build as hip_build
    try:
        hipcc = hip_build._hipcc()
    except RuntimeError:
        pytest.skip("no hipcc")
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip() if os.path.exists(clang) else ""
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no clang asan runtime")
    hip_build.build()                                       # zdr_kernels.o of the shipped library
    csrc = os.path.join(ROOT, "zdr_amd", "csrc")
    obj, lib = str(tmp_path / "zdr_api_asan.o"), str(tmp_path / "libzdr_hip_asan.so")
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer"]
    subprocess.run([hipcc, "-O1", "-g", *san, "-ffp-contract=off", "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-c", os.path.join(csrc, "zdr_api.cpp"), "-o", obj], check=True, capture_output=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *san, "-o", lib, obj, os.path.join(csrc, "zdr_kernels.o")], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "host_builder_under_asan.py"), lib], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
