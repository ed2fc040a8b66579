#!/usr/bin/env python3
"""README.md:21 of the reference: "atomic_fetch_add will become extremely slow" when the gradients concentrate on few
texels.  Times the path integrator's forward and PRB backward on cbox 512x512 spp 256 for material textures from 1x1
(every gradient lands on ONE texel) to 1024x1024, and checks the gradient sum against the 1024^2 run of the same
constant material (the total gradient of a constant material does not depend on the texture resolution)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--sizes", default="1,2,4,8,16,32,64,256,1024")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--integrator", default="path")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hotspot.json"))
a = ap.parse_args()
scene = make_scene(a.integrator)
W = a.res
ones = torch.ones((W, W, 4), device="cuda")
rows = []
ref_sum = None
for n in [int(s) for s in a.sizes.split(",")][::-1]:
    m = torch.empty((n, n, 4), device="cuda"); m[..., :3] = torch.tensor([0.6, 0.5, 0.4], device="cuda"); m[..., 3] = 0.5
    tf, tb = [], []
    for i in range(a.iters + 1):
        g = torch.zeros_like(m)
        torch.cuda.synchronize(); t = time.perf_counter(); scene.render_forward(m, (W, W), a.spp, i); torch.cuda.synchronize(); tf.append(time.perf_counter() - t)
        t = time.perf_counter(); scene.render_backward(ones, g, m, (W, W), a.spp, i); torch.cuda.synchronize(); tb.append(time.perf_counter() - t)
    scene.check()
    tot = g.double().sum(dim=(0, 1)).cpu().numpy()
    if ref_sum is None: ref_sum = tot
    r = {"texture": f"{n}x{n}", "cells": (n + 1) * (n + 1), "fwd_ms": round(min(tf[1:]) * 1e3, 3), "bwd_ms": round(min(tb[1:]) * 1e3, 3),
         "grad_sum": [float(v) for v in tot], "grad_sum_rel_to_1024": [float(v) for v in np.abs(tot - ref_sum) / np.abs(ref_sum)]}
    rows.append(r); print(r, flush=True)
json.dump({"workload": f"cbox {a.integrator} {W}x{W} spp={a.spp}, constant material (0.6, 0.5, 0.4, r 0.5)", "rows": rows}, open(a.out, "w"), indent=1)
