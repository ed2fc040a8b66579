#!/usr/bin/env python3
"""Traversal-only benchmark on the 1M-triangle tessellated cbox: coherent (camera) and incoherent
(random interior) rays through zdr_trace_closest / zdr_trace_any."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import cbox_models, CBOX_CAMERA
from zdr_amd.scenes import make_scene
from zdr_amd import procedural

n = int(sys.argv[1]) if len(sys.argv) > 1 else 183
scene = make_scene("path", arrays=procedural.tessellated_cbox(cbox_models(), n=n))
print(scene.info())
N = 1 << 23
g = torch.Generator(device="cuda").manual_seed(0)
# incoherent: random origins inside the room, random directions
lo = torch.tensor([-2.8, 0.1, -5.5], device="cuda"); hi = torch.tensor([2.3, 5.0, -0.5], device="cuda")
o = lo + (hi - lo) * torch.rand((N, 3), device="cuda", generator=g)
d = torch.randn((N, 3), device="cuda", generator=g); d = d / d.norm(dim=1, keepdim=True)
inc = torch.cat([o, torch.zeros((N, 1), device="cuda"), d, torch.full((N, 1), 1e30, device="cuda")], 1).contiguous()
# coherent: pinhole camera rays in scanline order of 8x8 tiles
W = 2904   # multiple of 8, W*W >= N
ys, xs = torch.meshgrid(torch.arange(W, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
t = ((ys // 8) * (W // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
px = (2.0 / W * (xs + 0.5) - 1.0) * np.tan(0.5 * CBOX_CAMERA[0]); py = (2.0 / W * (ys + 0.5) - 1.0) * np.tan(0.5 * CBOX_CAMERA[0])
dirs = torch.stack([px, -py, -torch.ones_like(px)], -1).reshape(-1, 3); dirs = dirs / dirs.norm(dim=1, keepdim=True)
order = torch.argsort(t.reshape(-1))[:N]
coh = torch.cat([torch.tensor(CBOX_CAMERA[1], device="cuda").expand(N, 3), torch.zeros((N, 1), device="cuda"), dirs[order], torch.full((N, 1), 1e30, device="cuda")], 1).contiguous()
for name, rays in (("coherent", coh), ("incoherent", inc)):
    for kind in ("closest", "any"):
        r = rays.clone()
        if kind == "any": r[:, 3] = 1e-4; r[:, 7] = 2.0
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = scene.trace_closest(r) if kind == "closest" else scene.trace_any(r)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        hits = (out[0][:, 0] >= 0).float().mean().item() if kind == "closest" else (out != 0).float().mean().item()
        print(f"{name:10s} {kind:7s}: {N / best / 1e6:8.1f} Mrays/s  ({best * 1e3:.2f} ms, hit rate {hits:.3f})")
