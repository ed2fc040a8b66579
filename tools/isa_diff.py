#!/usr/bin/env python3
"""Which kernels does a source change actually touch?  Compiles csrc/zdr_kernels.hip of a git revision and of the working tree to gfx950
assembly (hipcc -S, CPU only) and compares every kernel's instruction stream, labels normalised.  A kernel reported `identical` runs the
very same machine code: no timing is needed for it — and one that is NOT expected to change and does is the thing to time first.
(Round 4: a sampler change meant for the direct kernels shifted the register allocation of the BVH forward kernel and moved five spill
operations into its walk loop, +8 %; it had been A/B-timed on the Cornell box only.)
    python tools/isa_diff.py [rev]        rev defaults to HEAD; prints one line per kernel: identical | DIFFERENT (instructions, VALU, scratch ops old -> new)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-munsafe-fp-atomics", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only"]


def assemble(tree, out):
    subprocess.run([HIPCC, *FLAGS, "-I" + os.path.join(tree, "include"), "-I" + os.path.join(tree, "zdr_amd", "csrc"),
                    os.path.join(tree, "zdr_amd", "csrc", "zdr_kernels.hip"), "-o", out], check=True, capture_output=True)
    kernels, name = {}, None
    for line in open(out):
        m = re.match(r"^(_Z\w+):\s", line)
        if m and ".type" not in line:
            name = m.group(1); kernels[name] = []
        elif line.startswith(".Lfunc_end"):
            name = None
        elif name and line.startswith("\t") and not line.strip().startswith((";", ".")):
            kernels[name].append(re.sub(r"\.?L?BB\d+_\d+", "L", line.strip()))
    return kernels


def stats(body):
    return len(body), sum(l.startswith("v_") for l in body), sum(l.startswith("scratch_") for l in body)


rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
with tempfile.TemporaryDirectory() as tmp:
    old_tree = os.path.join(tmp, "old"); os.makedirs(old_tree)
    tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "zdr_amd/csrc", "include"], check=True, capture_output=True).stdout
    subprocess.run(["tar", "-x", "-C", old_tree], input=tar, check=True)
    old = assemble(old_tree, os.path.join(tmp, "old.s"))
    new = assemble(ROOT, os.path.join(tmp, "new.s"))
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0][:70]
changed = 0
for name in sorted(set(old) | set(new)):
    if name not in old or name not in new:
        print(f"{'ADDED' if name in new else 'REMOVED':10s} {demangle(name)}"); changed += 1
    elif old[name] == new[name]:
        print(f"identical  {demangle(name)}")
    else:
        changed += 1
        a, b = stats(old[name]), stats(new[name])
        print(f"DIFFERENT  {demangle(name)}   instructions {a[0]} -> {b[0]}, VALU {a[1]} -> {b[1]}, scratch ops {a[2]} -> {b[2]}")
print(f"{changed} of {len(set(old) | set(new))} kernels differ from {rev}")
