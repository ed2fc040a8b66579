#!/usr/bin/env python3
"""Does ordering the bounce rays pay on the 1 M-triangle BVH?  

The bounce rays of the c5 workload (1,004,672-triangle tessellated cbox, cbox camera) are rebuilt from the kernels' own
per-path traces: zdr_path_dump gives, per camera sample, the sampled continuation direction `wi` of every vertex and
whether the path went on; the vertex positions follow by tracing (camera ray through the pixel centre -> p0, p0 + wi0
-> p1, ...).  Each bounce's rays are then traced with the product's k_trace
  (i)  in PATH ORDER — the order in which the persistent waves meet them: 8x8 pixel tile, then sample index, then pixel;
  (ii) SORTED by a coherence key, several keys tried:
        oct+morton21   direction octant (3 bits) << 21 | 21-bit Morton code of the origin      (the verdict's key)
        morton30       30-bit Morton code of the origin alone
        morton21+oct   21-bit Morton code of the origin << 3 | octant (origin-major)
        m15+dir9       15-bit Morton of the origin << 9 | direction quantised to 3 bits per axis
and the rates are reported in G rays/s, with the time torch.sort needs for the key as a lower bound of what ordering costs.
Shadow rays (vertex -> a uniform point on the light, any-hit) are measured the same way.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from zdr_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--n", type=int, default=scenes.TESS1M_N)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ray_order.json"))
a = ap.parse_args()
dev = torch.device("cuda")
scene = scenes.make_scene("path", arrays=scenes.tess1m_arrays(a.n))
print(scene.info(), flush=True)
m = torch.from_numpy(scenes.cbox_material_np()).to(dev)
W, spp = a.res, a.spp

# queries in path order: tile (row-major) -> sample -> pixel of the tile: what 64 consecutive lanes of a wave pick up
T = W // 8
tile, s, p = torch.meshgrid(torch.arange(T * T, device=dev), torch.arange(spp, device=dev), torch.arange(64, device=dev), indexing="ij")
x = (tile % T) * 8 + (p & 7); y = (tile // T) * 8 + (p >> 3)
q = torch.stack([x, y, s], -1).reshape(-1, 3).to(torch.int32).contiguous()
N = q.shape[0]
MAXV = 3
dump = scene.path_dump(m, q, (W, W), spp, 0, maxv=MAXV)
nvert = dump[:, 0].view(torch.int32)
v = dump[:, 8:].reshape(N, MAXV, 24)
flags = v[..., 4].contiguous().view(torch.int32)
went_on = (flags & 2) != 0
wi = v[..., 6:9]

# camera rays through the pixel centres -> p0
cam = scenes.CBOX_CAMERA
o = torch.tensor(cam[1], device=dev); tgt = torch.tensor(cam[2], device=dev); up = torch.tensor(cam[3], device=dev)
fwd = (tgt - o) / (tgt - o).norm(); right = torch.linalg.cross(fwd, up); right = right / right.norm(); upp = torch.linalg.cross(right, fwd)
tn = float(np.tan(0.5 * cam[0]))
px = (2.0 / W * (q[:, 0].float() + 0.5) - 1.0) * tn; py = (2.0 / W * (q[:, 1].float() + 0.5) - 1.0) * tn
d0 = right[None] * px[:, None] - upp[None] * py[:, None] + fwd[None]; d0 = d0 / d0.norm(dim=1, keepdim=True)


def rays_of(orig, dirs, tmin=1e-3, tmax=1e30):
    n = orig.shape[0]
    tm = tmax if torch.is_tensor(tmax) else torch.full((n,), tmax, device=dev)
    return torch.cat([orig, torch.full((n, 1), tmin, device=dev), dirs, tm[:, None]], 1).contiguous()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best, out


def part1by2(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


lo = torch.tensor(scene._arrays.verts[:, :3].min(0), device=dev) - 1e-3
hi = torch.tensor(scene._arrays.verts[:, :3].max(0), device=dev) + 1e-3


def morton(orig, bits):
    g = ((orig - lo) / (hi - lo)).clamp(0, 1 - 1e-7)
    c = (g * (1 << bits)).to(torch.int64)
    return part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)


def keys(orig, dirs):
    octant = ((dirs[:, 0] < 0).long() | ((dirs[:, 1] < 0).long() << 1) | ((dirs[:, 2] < 0).long() << 2))
    dq = ((dirs * 0.5 + 0.5).clamp(0, 1 - 1e-7) * 8).long()
    return {
        "oct+morton21": (octant << 21) | morton(orig, 7),
        "morton30": morton(orig, 10),
        "morton21+oct": (morton(orig, 7) << 3) | octant,
        "m15+dir9": (morton(orig, 5) << 9) | (dq[:, 0] | (dq[:, 1] << 3) | (dq[:, 2] << 6)),
    }


report = {"scene": scene.info(), "res": W, "spp": spp, "camera_samples": N, "bounces": {}}


def measure(name, rays, anyhit):
    fn = (lambda r: scene.trace_any(r)) if anyhit else (lambda r: scene.trace_closest(r))
    n = rays.shape[0]
    t_path, out = timed(lambda: fn(rays))
    hit = (out != 0).float().mean().item() if anyhit else (out[0][:, 0] >= 0).float().mean().item()
    row = {"rays": n, "hit_rate": round(hit, 4), "path_order": {"ms": round(t_path * 1e3, 3), "grays_s": round(n / t_path / 1e9, 3)}}
    print(f"{name:28s} {n / 1e6:7.2f} M rays  path order {n / t_path / 1e9:6.2f} G rays/s", flush=True)
    for kname, k in keys(rays[:, 0:3], rays[:, 4:7]).items():
        t_sort, perm = timed(lambda: torch.sort(k)[1])
        sorted_rays = rays[perm].contiguous()
        t_s, _ = timed(lambda: fn(sorted_rays))
        row[kname] = {"ms": round(t_s * 1e3, 3), "grays_s": round(n / t_s / 1e9, 3), "speedup": round(t_path / t_s, 3),
                      "torch_sort_ms": round(t_sort * 1e3, 3), "speedup_incl_torch_sort": round(t_path / (t_s + t_sort), 3)}
        print(f"    sorted by {kname:14s} {n / t_s / 1e9:6.2f} G rays/s   x{t_path / t_s:5.2f}   (torch.sort of the key: {t_sort * 1e3:.2f} ms, trace {t_s * 1e3:.2f} ms)", flush=True)
    # random order: the floor
    perm = torch.randperm(n, device=dev)
    t_r, _ = timed(lambda: fn(rays[perm].contiguous()))
    row["random_order"] = {"ms": round(t_r * 1e3, 3), "grays_s": round(n / t_r / 1e9, 3)}
    print(f"    random order            {n / t_r / 1e9:6.2f} G rays/s", flush=True)
    report["bounces"][name] = row
    return out


# bounce 0: the camera rays themselves (coherent by construction)
ip, bt = measure("camera rays (closest)", rays_of(o.expand(N, 3).contiguous(), d0, 0.0), False)
pos = o[None] + d0 * bt[:, 2:3]
alive = (ip[:, 0] == 0) & (nvert >= 1)
# the light: a uniform point on instance 1's triangles (two triangles of equal area on the cbox light)
A = scene._arrays
lt = A.tris[A.inst_tri_begin[1]:A.inst_tri_begin[2]]
lv = torch.tensor(A.verts[lt.reshape(-1), :3].reshape(-1, 3, 3), device=dev)
gen = torch.Generator(device=dev).manual_seed(0)
for k in range(MAXV):
    idx = torch.nonzero(alive & went_on[:, k] & (nvert > k)).squeeze(1)      # path order is kept: nonzero() is ascending
    sh_idx = torch.nonzero(alive & (nvert > k)).squeeze(1)
    if sh_idx.numel() > 1000:
        u = torch.rand((sh_idx.numel(), 2), device=dev, generator=gen); fl = u.sum(1) > 1; u[fl] = 1 - u[fl]
        tri = lv[torch.randint(0, lv.shape[0], (sh_idx.numel(),), device=dev, generator=gen)]
        pl = tri[:, 0] + (tri[:, 1] - tri[:, 0]) * u[:, :1] + (tri[:, 2] - tri[:, 0]) * u[:, 1:]
        dl = pl - pos[sh_idx]; dist = dl.norm(dim=1); dl = dl / dist[:, None]
        measure(f"shadow rays of vertex {k} (any)", rays_of(pos[sh_idx], dl, 1e-3, 0.9999 * dist), True)
    if idx.numel() < 1000:
        break
    r = rays_of(pos[idx], wi[idx, k].contiguous())
    ip_k, bt_k = measure(f"bounce {k + 1} rays (closest)", r, False)
    newpos = pos.clone(); newpos[idx] = pos[idx] + wi[idx, k] * bt_k[:, 2:3]
    hit_surface = torch.zeros(N, dtype=torch.bool, device=dev); hit_surface[idx] = ip_k[:, 0] == 0
    pos, alive = newpos, hit_surface

os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(report, open(a.out, "w"), indent=1)
print("wrote", a.out)
