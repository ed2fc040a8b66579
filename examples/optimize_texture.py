#!/usr/bin/env python3
"""Inverse rendering with zdr_amd: recover the Cornell box's diffuse / roughness texture from a rendered target by gradient
descent — the workflow of the reference's example.py (render a ground truth, start from a random material, Adam on the
image loss through scene.render's PRB backward), on the assets this repository ships.

    python examples/optimize_texture.py --iters 200 --res 256 --spp 16 --out /tmp/zdr_example
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from PIL import Image

from zdr_amd import Camera, Scene, float3

ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


def load_material(diffuse_file, roughness_file):
    d = np.asarray(Image.open(diffuse_file))[..., :3]
    r = np.asarray(Image.open(roughness_file))[..., :1]
    return torch.from_numpy(np.ascontiguousarray((np.concatenate([d, r], -1).astype(np.float32) / 255.0) ** 2.2)).cuda()


def save_png(path, img):
    Image.fromarray((img[..., :3].clamp(0, 1) ** 0.454 * 255).to(torch.uint8).cpu().numpy()).save(path)


def run(iters=200, res=256, spp=16, tex=256, out=None, integrator="path", seed=0, verbose=True):
    scene = Scene([(os.path.join(ASSETS, "cboxuv.obj"), None, float3(0.0)),
                   (os.path.join(ASSETS, "cbox-light.obj"), None, float3(17, 12, 4))], integrator=integrator)
    scene.camera = Camera(fov=50 / 180 * 3.1415926, origin=float3(-0.2, 2.6, 6.0), target=float3(-0.2, 2.6, -2.5), up=float3(0.0, 1.0, 0.0))
    material_gt = load_material(os.path.join(ASSETS, "cboxd.png"), os.path.join(ASSETS, "cboxr.png"))
    image_gt = scene.render(material_gt, res=(res, res), spp=max(256, 4 * spp))          # seed defaults to 0
    rng = random.Random(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    material = torch.rand((tex, tex, 4), device="cuda", generator=g).requires_grad_()
    opt = torch.optim.Adam([material], lr=0.02)
    losses = []
    for it in range(iters):
        opt.zero_grad()
        image = scene.render(material, res=(res, res), spp=spp, seed=rng.randint(0, 2147483646))
        loss = (image[..., :3] - image_gt[..., :3]).abs().mean()
        loss.backward()
        opt.step()
        with torch.no_grad():
            material.clamp_(1e-3, 1.0)                                                     # roughness / albedo stay physical
        losses.append(float(loss))
        if verbose and (it % 20 == 0 or it == iters - 1):
            print(f"iteration {it:4d}  L1 image loss {losses[-1]:.5f}", flush=True)
    if out:
        os.makedirs(out, exist_ok=True)
        save_png(os.path.join(out, "target.png"), image_gt)
        save_png(os.path.join(out, "result.png"), scene.render(material.detach(), res=(res, res), spp=max(256, 4 * spp)))
        save_png(os.path.join(out, "texture_diffuse.png"), material.detach())
        duvdxy = scene.render_duvdxy(material.detach(), res=(res, res), spp=16)            # screen -> texture Jacobian (example.py)
        footprint = torch.det(duvdxy.reshape(res, res, 2, 2)).abs() * tex * tex
        Image.fromarray((footprint.clamp(0, 1) ** 0.454 * 255).to(torch.uint8).cpu().numpy()).save(os.path.join(out, "footprints.png"))
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--tex", type=int, default=256)
    ap.add_argument("--integrator", default="path")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    run(a.iters, a.res, a.spp, a.tex, a.out, a.integrator)
