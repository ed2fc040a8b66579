#!/usr/bin/env python3
"""BASELINE configs[4]: ~1M-triangle synthetic scene (displaced height field + quad light), path + PRB."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import fd_material_np
from zdr_amd.scenes import make_scene, terrain_arrays, terrain_camera

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="tess", choices=["tess", "terrain"])
ap.add_argument("--n", type=int, default=0)          # tess: 30 n^2 triangles (183 -> 1.0M); terrain: 2 n^2 (707 -> 1.0M)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--iters", type=int, default=2)
a = ap.parse_args()
from zdr_amd.scenes import cbox_models, cbox_material_np
from zdr_amd import procedural
t0 = time.time()
A = procedural.tessellated_cbox(cbox_models(), n=a.n or 183) if a.scene == "tess" else terrain_arrays(n=a.n or 707)
t1 = time.time()
scene = make_scene("path", arrays=A)
if a.scene == "terrain": scene.camera = terrain_camera()
t2 = time.time()
print("scene:", scene.info(), f"gen {t1-t0:.1f}s build {t2-t1:.1f}s")
m = torch.from_numpy(cbox_material_np() if a.scene == "tess" else fd_material_np(1024, 0)).cuda()
W = a.res; ones = torch.ones((W, W, 4), device="cuda"); g = torch.zeros_like(m)
for which in ("fwd", "bwd"):
    ts = []
    for i in range(a.iters + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        if which == "fwd": img = scene.render_forward(m, (W, W), a.spp, i)
        else: scene.render_backward(ones, g, m, (W, W), a.spp, i)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"{which}: {min(ts[1:])*1e3:.2f} ms  {W*W*a.spp/min(ts[1:])/1e6:.1f} Msamples/s")
print("stats:", scene.render_stats(m, (W, W), 4))
print("image mean", img[..., :3].mean().item(), "nan", torch.isnan(img).any().item(), "grad sum", g.sum().item())
