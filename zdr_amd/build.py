"""Ahead-of-time build of libzdr_hip.so (gfx950 only; no JIT, no multi-arch fat binary).

``python -m zdr_amd.build`` or ``zdr_amd.build.build()``.  hipcc cross-compiles without a GPU;
the resulting .so sits in-tree next to the sources so that it travels with the repository
snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libzdr_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the zdr HIP back end needs ROCm (no CPU fallback exists)")


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp", ".h"))] + [
        os.path.join(ROOT, "include", "zdr.h")]


def source_hash() -> str:
    """sha256 over the kernel / host sources and the header, in name order: identifies the code a profile was taken on
    (profiles/pmc_traffic.json carries it; bench.py marks counter figures stale when it differs).  Works without git —
    the GPU box receives a snapshot, not a repository."""
    import hashlib
    h = hashlib.sha256()
    for path in _sources():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    hipcc = _hipcc()
    common = [f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs = []
    jobs = [
        # kernels: default contraction (FMA) for speed; NO fast-math (NaN policy, integrator.py:27).
        # -fno-slp-vectorize: hipcc otherwise packs scalar f32 math into v_pk_*_f32, which on gfx950
        # runs at the same FLOP rate as the scalar forms and pays extra moves to build register
        # pairs: measured 23.9 -> 17.7 ms on the cbox forward pass (profiles/r1_ab_flags.txt).
        ("zdr_kernels.hip", ["-O3", "-munsafe-fp-atomics", "-fno-slp-vectorize", *os.environ.get("ZDR_KERNEL_FLAGS", "").split()]),
        # host side: IEEE float32 for the per-triangle constants
        # (-D options of ZDR_KERNEL_FLAGS go to both halves: some macros size shared workspaces)
        ("zdr_api.cpp", ["-O2", "-ffp-contract=off", "-x", "hip", *[f for f in os.environ.get("ZDR_KERNEL_FLAGS", "").split() if f.startswith("-D")]]),
    ]
    procs = []
    for src, extra in jobs:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc, *extra, *common, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
