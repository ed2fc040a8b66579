#!/usr/bin/env python3
"""Runs only the forward or only the backward pass of the cbox bench workload a few times
(for rocprofv3 --pmc passes and kernel experiments)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zdr_amd.scenes import cbox_material_np, fd_material_np
from zdr_amd.scenes import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="fwd", choices=["fwd", "bwd", "both"])
ap.add_argument("--integrator", default="path")
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--accel", default="auto")
ap.add_argument("--material", default="A")
ap.add_argument("--env", action="store_true", help="add a sun-and-sky environment map (the ENV kernel instantiations)")
a = ap.parse_args()
scene = make_scene(a.integrator, accel=a.accel)
if a.env:
    import numpy as np
    sky = np.random.default_rng(0).uniform(0.05, 0.6, (32, 64, 3)).astype(np.float32); sky[5:8, 40:44] = (300.0, 260.0, 200.0)
    scene.add_envmap(sky)
m = torch.from_numpy(cbox_material_np() if a.material == "A" else fd_material_np(1024, 0)).cuda()
W = a.res
ones = torch.ones((W, W, 4), device="cuda")
g = torch.zeros_like(m)
for which in (["fwd", "bwd"] if a.which == "both" else [a.which]):
    ts = []
    for i in range(a.iters + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        if which == "fwd":
            scene.render_forward(m, (W, W), a.spp, i)
        else:
            scene.render_backward(ones, g, m, (W, W), a.spp, i)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    best = min(ts[1:])
    print(f"{which}: {best*1e3:.3f} ms  {W*W*a.spp/best/1e6:.1f} Msamples/s")
