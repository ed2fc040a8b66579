#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs that fit one GPU (c1 runs on the oracle = the CPU case)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import cbox_material_np, cbox_models
from zdr_amd.scenes import make_scene
from zdr_amd import procedural

def timed(fn, iters=3):
    ts = []
    for i in range(iters + 1):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(i); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts[1:])

m = torch.from_numpy(cbox_material_np()).cuda()
out = {}
def run(name, scene, W, spp, backward):
    n = W * W * spp
    ones = torch.ones((W, W, 4), device="cuda"); g = torch.zeros_like(m)
    tf = timed(lambda i: scene.render_forward(m, (W, W), spp, i))
    r = {"fwd_ms": round(tf * 1e3, 3), "fwd_msamples_s": round(n / tf / 1e6, 1)}
    if backward:
        tb = timed(lambda i: scene.render_backward(ones, g, m, (W, W), spp, i))
        r.update({"bwd_ms": round(tb * 1e3, 3), "bwd_msamples_s": round(n / tb / 1e6, 1)})
    out[name] = r; print(name, r, flush=True)

run("c1 collocated 256x256 spp1 (GPU, for reference)", make_scene("collocated"), 256, 1, True)
run("c2 direct 512x512 spp64 fwd", make_scene("direct"), 512, 64, True)
run("c3 path 512x512 spp256", make_scene("path"), 512, 256, True)
run("c4-shape path 1024x1024 spp1024 on ONE GPU", make_scene("path"), 1024, 1024, True)
big = make_scene("path", arrays=procedural.tessellated_cbox(cbox_models(), n=183))
print(big.info())
run("c5 1M-tri tessellated cbox path 1024x1024 spp256", big, 1024, 256, True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
