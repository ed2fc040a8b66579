#!/usr/bin/env python3
"""Generates tests/golden/cbox_golden.npz from the CPU oracle (IEEE build).

The reference holds no golden vectors and cannot run, so these fixtures pin the
ORACLE against regressions and give the GPU tests a second, frozen target.  Inputs are the cbox
scene (tests/golden/assets), the seeded material B of SURVEY §8d at 64x64 texels, camera of
fd_validate.py:28-33.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from conftest import CBOX_CAMERA, cbox_models, fd_material_np
from zdr_amd import geometry

NTHREADS = 1       # pixels are independent and each is summed in sample order: the images do not depend on it; the gradients are float64 sums per thread

CASES = [  # name, integrator, W, spp, seed, tent
    ("collocated_32_spp1", "collocated", 32, 1, 0, True),
    ("direct_32_spp4", "direct", 32, 4, 0, True),
    ("path_32_spp4", "path", 32, 4, 0, True),
    ("path_24_spp16_box_seed7", "path", 24, 16, 7, False),
    # round 3: sizes at which whole-image statistics of a glossy material mean something (65,536 / 36,864 paths; the
    # cases above hold a few thousand and are compared path by path on the GPU, tests/test_golden.py)
    ("path_64_spp16", "path", 64, 16, 0, True),
    ("path_48_spp16_box_seed7", "path", 48, 16, 7, False),
]


def main():
    A = geometry.assemble(cbox_models())
    S = oracle.OracleScene.from_arrays(A)
    mat = fd_material_np(64, 1)
    out = {"material": mat}
    for name, integ, W, spp, seed, tent in CASES:
        p = oracle.make_params(integ, W, W, spp, seed, CBOX_CAMERA, mat.shape[:2], use_tent=tent, nthreads=NTHREADS)
        out[name + "/image"] = S.render_forward(p, mat)
        cot = np.ones((W, W, 4), np.float32)
        pb = oracle.make_params(integ, W, W, spp, seed + 1, CBOX_CAMERA, mat.shape[:2], use_tent=tent, nthreads=NTHREADS)
        out[name + "/grad"] = S.render_backward(pb, cot, mat)
    out["sampler_cmj_px24_py345_seed0_spp16"] = np.stack([oracle.sampler_dump(oracle.SAMPLER_CMJ, 24, 345, 0, 16, i, nvert=3) for i in range(16)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cbox_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
