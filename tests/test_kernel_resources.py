"""The occupancy the persistent kernels are built for, read from the metadata of the BUILT library (no GPU needed): the code
object embedded in libzdr_hip.so is carved out and its AMDGPU notes are read with llvm-readelf.

Why a test: gfx950 hands out a CU's 160 KiB of LDS in 128 blocks of 1,280 bytes and its 512 VGPRs per SIMD lane in whole waves.
A few bytes of LDS or a few registers too many cost a whole wave per CU or per SIMD — silently: the kernels stay correct and get
5-10 % slower (profiles/r3_bwd_records_and_atomics.txt: 13,264 bytes of LDS were 11 blocks, 11 waves per CU, and the launch took
as long as the old kernel's)."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from zdr_amd import _native

LDS_BLOCK, LDS_BLOCKS_PER_CU = 1280, 128
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels():
    if not os.path.exists(READELF):
        pytest.skip("llvm-readelf not found")
    blob = open(_native.LIB_PATH, "rb").read()
    off = blob.find(b"\x7fELF\x02\x01\x01\x40")                  # ELF64, little endian, OS ABI 64 = AMDGPU HSA
    assert off > 0, "no gfx code object inside the library"
    e_shoff, = struct.unpack_from("<Q", blob, off + 0x28)
    e_shentsize, e_shnum = struct.unpack_from("<HH", blob, off + 0x3A)
    path = os.path.join(os.path.dirname(_native.LIB_PATH), "_gfx950_code_object.tmp")
    try:
        with open(path, "wb") as f:
            f.write(blob[off:off + e_shoff + e_shentsize * e_shnum])
        out = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
    finally:
        if os.path.exists(path):
            os.remove(path)
    found = {}
    for m in re.finditer(r"- \.agpr_count.*?(?=\n  - \.agpr_count|\Z)", out, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s*(\S+)", blk).group(1)
        found[name] = {k: int(re.search(r"\.%s:\s*(\d+)" % k, blk).group(1))
                       for k in ("group_segment_fixed_size", "vgpr_count", "sgpr_count", "private_segment_fixed_size")}
    assert found
    return found


def waves_per_cu(lds_bytes, vgprs):
    by_lds = LDS_BLOCKS_PER_CU // max(1, -(-lds_bytes // LDS_BLOCK))
    by_vgpr = 4 * min(8, 512 // max(1, -(-vgprs // 8) * 8))
    return min(by_lds, by_vgpr)


def pick(found, pattern):
    sel = {n: r for n, r in found.items() if re.search(pattern, n)}
    assert sel, pattern
    return sel


def test_backward_path_kernel_holds_sixteen_waves_per_cu_brute_force():
    """k_path_bwd<*, BruteAccel, *>: LDS is what decides — 8 blocks = 10,240 bytes per wave (100 pool slots, the scatter queue,
    the tile origins; neither seeds nor cotangents: accel.h) — and 128 VGPRs."""
    for name, r in pick(kernels(), r"k_path_bwdILi[01]E10BruteAccel").items():
        assert r["group_segment_fixed_size"] <= 8 * LDS_BLOCK, (name, r)
        assert r["vgpr_count"] <= 128, (name, r)
        assert waves_per_cu(r["group_segment_fixed_size"], r["vgpr_count"]) >= 16, (name, r)


def test_backward_path_kernel_holds_sixteen_waves_per_cu_bvh():
    """k_path_bwd<*, BvhAccel, *>: 8 blocks = 10,240 bytes per wave, 2,560 of which are the traversal stack the launcher adds
    (ZDR_BVH_LDS_STACK_BWD = 10 entries x 64 lanes x 4 bytes), and 128 VGPRs."""
    for name, r in pick(kernels(), r"k_path_bwdILi[01]E8BvhAccel").items():
        assert r["group_segment_fixed_size"] + 10 * 256 <= 8 * LDS_BLOCK, (name, r)
        assert r["vgpr_count"] <= 128, (name, r)


def test_forward_path_kernel_holds_four_waves_per_simd_without_spills():
    """k_path<cmj, BruteAccel>: 128 VGPRs at most and not a byte of scratch — the SGPR spills to VGPR lanes (320 v_readlane in
    the loop) went with the kernarg reload, and the two VGPRs that carried them with them."""
    for name, r in pick(kernels(), r"k_pathILi0E10BruteAccelLb0ELb0E").items():
        assert r["vgpr_count"] <= 128 and r["private_segment_fixed_size"] == 0, (name, r)


def test_backward_path_kernel_spills_nothing_brute_force():
    """k_path_bwd<cmj, BruteAccel>: its private segment is the overflow records of the pool (16 x 80 bytes + 16 links, rounded: 1,376 bytes)
    and NOTHING else.  The kernel sits exactly at 128 VGPRs; round 4 measured what four more live registers cost — four spilled dwords in
    the hot loop, 1,392 bytes, 10.7 -> 11.5 ms (+7 %) — with every other figure of this file unchanged.  A guard, not a tuning knob."""
    for name, r in pick(kernels(), r"k_path_bwdILi0E10BruteAccelLb0E").items():
        assert r["private_segment_fixed_size"] <= 1376, (name, r)
