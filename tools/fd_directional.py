#!/usr/bin/env python3
"""directional derivative of <W, I(material + t delta)> over the WHOLE image, AD
(<grad, delta> from one backward pass) against two-sided FD (eps = 0.01, same seed in both renders),
pooled over seeds.  Averaging over all pixels resolves the roughness derivative, whose per-pixel FD is
very noisy (a roughness change moves the VNDF samples)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import fd_material_np
from zdr_amd.scenes import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--spp", type=int, default=4096)
ap.add_argument("--seeds", type=int, default=32)
ap.add_argument("--rr-depth", type=int, default=2)
ap.add_argument("--max-depth", type=int, default=16)
ap.add_argument("--eps", type=float, default=0.01)
ap.add_argument("--scene", default="cbox", choices=["cbox", "tess1m", "env", "lights3"],
                help="tess1m: the 1,004,672-triangle tessellated cbox (BASELINE configs[4], BVH kernels); env: cbox + a sun-and-sky "
                     "environment map (the adjoint of the environment's light sampling and MIS); lights3: three emitters of different "
                     "sizes and colours plus a blocker (zdr_amd/scenes.py, multi_light_arrays)")
ap.add_argument("--integrator", default="path", choices=["path", "direct"])
ap.add_argument("--only", default="", help="comma list of diffuse,roughness,all")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fd_directional.json"))
a = ap.parse_args()
if a.scene == "tess1m":
    from zdr_amd.scenes import cbox_models
    from zdr_amd import procedural
    scene = make_scene(a.integrator, arrays=procedural.tessellated_cbox(cbox_models(), n=183))
elif a.scene == "lights3":
    from zdr_amd.scenes import multi_light_arrays
    scene = make_scene(a.integrator, arrays=multi_light_arrays())
else:
    scene = make_scene(a.integrator)
    if a.scene == "env":
        sky = np.random.default_rng(0).uniform(0.05, 0.6, (32, 64, 3)).astype(np.float32)    # tests/test_envmap.py, sun_sky()
        sky[5:8, 40:44] = (300.0, 260.0, 200.0)
        scene.add_envmap(sky)
scene.rr_depth, scene.max_depth = a.rr_depth, a.max_depth
W = a.res
material = torch.from_numpy(fd_material_np(1024, 0)).cuda()
g = torch.Generator(device="cuda").manual_seed(1)
wimg = torch.rand((W, W, 4), device="cuda", generator=g) + 0.5; wimg[..., 3] = 0
res = {}
for name, chans in (("diffuse", slice(0, 3)), ("roughness", slice(3, 4)), ("all", slice(0, 4))):
    delta = torch.zeros_like(material)
    delta[..., chans] = torch.rand(material[..., chans].shape, device="cuda", generator=g)
    eps = a.eps
    if a.only and name not in a.only.split(','): continue
    ad, fd = [], []
    for s in range(a.seeds):
        d = torch.zeros_like(material)
        scene.render_backward(wimg, d, material, (W, W), a.spp, 7000 + s)
        ad.append((d.double() * delta.double()).sum().item())
        ip = scene.render_forward(material + eps * delta, (W, W), a.spp, 3000 + s).double()
        im = scene.render_forward(material - eps * delta, (W, W), a.spp, 3000 + s).double()
        fd.append((((ip - im) * wimg.double()).sum() / (2 * eps)).item())
    ad, fd = np.array(ad), np.array(fd)
    rel = abs(ad.mean() - fd.mean()) / abs(fd.mean())
    sig = np.hypot(ad.std(ddof=1), fd.std(ddof=1)) / np.sqrt(a.seeds) / abs(fd.mean())
    res[name] = {"AD": ad.mean(), "AD_se": ad.std(ddof=1) / np.sqrt(a.seeds), "FD": fd.mean(), "FD_se": fd.std(ddof=1) / np.sqrt(a.seeds), "rel_err": rel, "one_sigma": sig}
    print(f"{name:9s}: AD = {ad.mean():.4f} +- {res[name]['AD_se']:.4f}  FD = {fd.mean():.4f} +- {res[name]['FD_se']:.4f}  rel-err {rel:.2e} (1 sigma {sig:.2e})", flush=True)
from zdr_amd import build as hip_build
json.dump({"csrc_sha256": hip_build.source_hash(), "scene": a.scene, "integrator": a.integrator, "res": W, "spp": a.spp, "seeds": a.seeds, "fd_eps": a.eps, "result": res}, open(a.out, "w"), indent=1)
