#!/usr/bin/env python3
"""Automated counterpart of /root/reference/fd_validate.py: compares the PRB
gradient (AD) of ONE image pixel w.r.t. ONE texel with two-sided finite differences (FD) of the
forward render, eps = 0.01 (fd_validate.py:92), same seed in both FD renders (:72-81), single-pixel
backward for AD (:84-89), the reference's five seeds (:97) and spp = 2^0 .. 2^12 (:96,100).

Procedure as in the reference (:133-177): render the image at 1024^2 spp 128, draw the pixel by
importance sampling its brightness (light pixels masked), back-propagate that pixel, draw the texel by
importance sampling |grad|.  Only the 8x8 tile that contains the pixel is rendered in the sweeps (the
C-ABI's shard rectangle), which makes a high-sample tail affordable: the table is 'compared by eye' in
the reference (:114); the tail pools many seeds at 2^16 spp to put a number on the agreement.

Scene/material: cbox + the seeded interior material B of SURVEY §8d (FD needs texels away from 0 and 1,
fd_validate.py:93-94; the stock cboxr.png is 1.0 everywhere)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import fd_material_np
from zdr_amd.scenes import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--channel", default="any", choices=["any", "diffuse", "roughness"])
ap.add_argument("--tail-spp", type=int, default=1 << 16)
ap.add_argument("--tail-seeds", type=int, default=64)
ap.add_argument("--pick-seed", type=int, default=0)
ap.add_argument("--pick", default="sample", choices=["sample", "max"], help="texel: importance-sample |grad| (reference) or take the strongest texel of a 4096-spp gradient")
ap.add_argument("--eps", type=float, default=0.01, help="finite-difference step (fd_validate.py:92 uses 0.01)")
ap.add_argument("--skip-table", action="store_true", help="only the high-sample tail")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fd_validate.json"))
a = ap.parse_args()

RES = (1024, 1024)
FD_EPS = a.eps
SEEDS = [0, 12345, 853402567, 19260817, 948377263]
scene = make_scene("path")
material = torch.from_numpy(fd_material_np(1024, 0)).cuda()
gen = torch.Generator(device="cuda").manual_seed(a.pick_seed)

def pick(weights):
    flat = weights.flatten().clamp(min=0)
    i = torch.multinomial(flat, 1, generator=gen).item()
    return tuple(int(v) for v in np.unravel_index(i, tuple(weights.shape)))

# ---- choose the pixel and the texel like fd_validate.py:133-177
m = material.clone().requires_grad_()
I = scene.render(m, res=RES, spp=128)
black = material.clone(); black[..., :3] = 0
I_black = scene.render(black, res=RES, spp=128)
w = I.detach().clone(); w[(I.detach() == I_black).all(dim=-1)] = 0        # mask light pixels
imgidx = pick(w[..., 0:3])
I[imgidx].backward()
g = m.grad.abs()
if a.channel == "diffuse": g[..., 3] = 0
if a.channel == "roughness": g[..., :3] = 0
if a.pick == "max":   # a spp-128 gradient is mostly single stray paths: re-estimate the pixel's gradient before choosing
    yy, xx, _ = imgidx
    r0 = (xx // 8 * 8, yy // 8 * 8, xx // 8 * 8 + 8, yy // 8 * 8 + 8)
    cot0 = torch.zeros((RES[1], RES[0], 4), device="cuda"); cot0[imgidx] = 1.0
    g = torch.zeros_like(material)
    for s_ in range(16):
        scene.render_backward(cot0, g, material, RES, 4096, 100 + s_, rect=r0)
    g = (g / 16).abs()
    if a.channel == "diffuse": g[..., 3] = 0
    if a.channel == "roughness": g[..., :3] = 0
    texidx = tuple(int(v) for v in np.unravel_index(int(g.flatten().argmax().item()), tuple(g.shape)))
else:
    texidx = pick(g)
print("Image index:", imgidx, " pixel value:", I[imgidx].item())
print("Texture index:", texidx, " texel value:", material[texidx].item(), " texel gradient (spp 128):", m.grad[texidx].item())
assert FD_EPS <= material[texidx].item() <= 1 - FD_EPS

y, x, c = imgidx
rect = (x, y, x + 1, y + 1)      # only this pixel is read: render nothing else (the C-ABI's shard rectangle)

def fd_grad(spp, seed):
    vals = []
    for sgn in (-1, +1):
        mm = material.clone(); mm[texidx] += sgn * FD_EPS
        vals.append(scene.render_forward(mm, RES, spp, seed, rect=rect)[imgidx].item())
    return (vals[1] - vals[0]) / (2 * FD_EPS)

def ad_grad(spp, seed):
    cot = torch.zeros((RES[1], RES[0], 4), device="cuda"); cot[imgidx] = 1.0
    d = torch.zeros_like(material)
    scene.render_backward(cot, d, material, RES, spp, seed, rect=rect)   # uses seed + 1 like render.py:196
    return d[texidx].item()

rows = {"FD": [], "AD": []}
for name, fn in (() if a.skip_table else (("FD", fd_grad), ("AD", ad_grad))):
    print(f"{name}:  (rows: spp = 2^0 .. 2^12, columns: seeds {SEEDS})")
    for e in range(13):
        r = [fn(2 ** e, s) for s in SEEDS]
        rows[name].append(r)
        print(" ".join(f"{v: .6f}" for v in r))
t0 = time.time()
fd, ad = [], []
for s in range(a.tail_seeds):
    fd.append(fd_grad(a.tail_spp, 1000 + s)); ad.append(ad_grad(a.tail_spp, 500000 + s))
    if (s + 1) % 64 == 0:
        f_, a_ = np.array(fd), np.array(ad)
        print(f"  {s + 1} seeds ({time.time() - t0:.0f} s): FD {f_.mean():.6f} +- {f_.std(ddof=1) / np.sqrt(len(f_)):.6f}  AD {a_.mean():.6f} +- {a_.std(ddof=1) / np.sqrt(len(a_)):.6f}", flush=True)
fd, ad = np.array(fd), np.array(ad)
fd_m, ad_m = fd.mean(), ad.mean()
fd_se, ad_se = fd.std(ddof=1) / np.sqrt(len(fd)), ad.std(ddof=1) / np.sqrt(len(ad))
rel = abs(ad_m - fd_m) / abs(fd_m)
sigma = np.hypot(fd_se, ad_se) / abs(fd_m)
print(f"tail: {a.tail_seeds} seeds x 2^{int(np.log2(a.tail_spp))} spp ({time.time() - t0:.1f} s):  FD = {fd_m:.6f} +- {fd_se:.6f}   AD = {ad_m:.6f} +- {ad_se:.6f}")
print(f"grad rel-err |AD - FD| / |FD| = {rel:.3e}   (statistical resolution 1 sigma = {sigma:.3e})")
from zdr_amd import build as hip_build
json.dump({"csrc_sha256": hip_build.source_hash(), "imgidx": imgidx, "texidx": texidx, "fd_eps": FD_EPS, "table_spp": [2 ** e for e in range(13)], "seeds": SEEDS,
           "FD": rows["FD"], "AD": rows["AD"], "tail": {"spp": a.tail_spp, "seeds": a.tail_seeds, "FD": fd_m, "FD_se": fd_se, "AD": ad_m, "AD_se": ad_se,
           "rel_err": rel, "one_sigma": sigma}}, open(a.out, "w"), indent=1)
