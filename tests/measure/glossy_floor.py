#!/usr/bin/env python3
"""How far may the HIP kernels sit from the oracle on a GLOSSY material, measured against the oracle's own two builds?

On material B (roughness 0.3-0.9) last-ulp differences are amplified chaotically, so the image-level parity
bars are expressed in units of a ruler: what the SAME oracle source compiled with FMA contraction differs from its IEEE
build by.  The HIP build perturbs MORE operations than that ruler does — FMA contraction (other choices than gcc's), and
v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / device sin, cos at 1-2 ulp each — so its distance from the IEEE oracle is a multiple
of the ruler.  This tool measures that multiple: for many seeds it renders forward + backward on the GPU and with both CPU
builds, sets aside the paths that measurably took another branch (tests/gpu_util.Flips, all three builds), and prints the
ratio HIP-vs-IEEE / FMA-vs-IEEE of every robust statistic the parity assertions use.  tests/gpu_util.FLOOR_FACTOR quotes
the result (profiles/r3_glossy_floor_ratio.txt).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle
from gpu_util import Flips, grad_diff_stats, image_diff_stats, multi_light_arrays, oracle_params
from zdr_amd import geometry, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=16)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "glossy_floor_ratio.json"))
a = ap.parse_args()

CASES = [  # name, scene arrays, material, W, spp
    ("cbox B(256) 64x64 spp16", None, scenes.fd_material_np(256, 0), 64, 16),
    ("cbox B(64) 32x32 spp4 (golden size)", None, scenes.fd_material_np(64, 1), 32, 4),
    ("three lights B(256) 64x64 spp16", multi_light_arrays(), scenes.fd_material_np(256, 0), 64, 16),
    ("cbox B(1024) 96x96 spp16", None, scenes.fd_material_np(1024, 0), 96, 16),
]
rows = {}
for name, arrays, mat, W, spp in CASES:
    A = arrays if arrays is not None else geometry.assemble(scenes.cbox_models())
    scene = scenes.make_scene("path", arrays=A)
    S, Sf = oracle.OracleScene.from_arrays(A), oracle.OracleScene.from_arrays(A, variant="fma")
    m = torch.from_numpy(mat).cuda()
    ones = np.ones((W, W, 4), np.float32)
    acc = {k: [] for k in ("img frac_bad", "img mean_rel", "grad frac_bad", "grad rel_l1", "flips fwd", "flips bwd")}
    for seed in range(100, 100 + a.seeds):
        img = scene.render_forward(m, (W, W), spp, seed).cpu().numpy()[..., :3]
        g = torch.zeros_like(m); scene.render_backward(torch.from_numpy(ones).cuda(), g, m, (W, W), spp, seed); g = g.cpu().numpy()
        p, pb = oracle_params(scene, W, W, spp, seed, mat.shape[:2]), oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2])
        ref, flo = S.render_forward(p, mat)[..., :3], Sf.render_forward(p, mat)[..., :3]
        gref, gflo = S.render_backward(pb, ones, mat), Sf.render_backward(pb, ones, mat)
        ff = Flips(scene, S, Sf, mat, (W, W), spp, seed)
        fb = Flips(scene, S, Sf, mat, (W, W), spp, seed + 1, cot=ones)
        k = ~ff.pixels; t = ~fb.texels
        si, fi = image_diff_stats(img[k], ref[k]), image_diff_stats(flo[k], ref[k])
        sg, fg = grad_diff_stats(g[t], gref[t]), grad_diff_stats(gflo[t], gref[t])
        acc["img frac_bad"].append(si["frac_bad"] / max(fi["frac_bad"], 1e-12)); acc["img mean_rel"].append(si["mean_rel"] / fi["mean_rel"])
        acc["grad frac_bad"].append(sg["frac_bad"] / max(fg["frac_bad"], 1e-12)); acc["grad rel_l1"].append(sg["rel_l1"] / fg["rel_l1"])
        acc["flips fwd"].append(ff.count / max(ff.floor_count, 1)); acc["flips bwd"].append(fb.count / max(fb.floor_count, 1))
    rows[name] = {k: {"median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "max": float(np.max(v))} for k, v in acc.items()}
    print(f"== {name}  ({a.seeds} seeds): HIP-vs-IEEE / FMA-vs-IEEE, flipped paths of all three builds set aside")
    for k, r in rows[name].items():
        print(f"   {k:14s} median {r['median']:5.2f}   90th percentile {r['p90']:5.2f}   max {r['max']:5.2f}")
    sys.stdout.flush()
json.dump(rows, open(a.out, "w"), indent=1)
