#!/usr/bin/env python3
"""BASELINE configs[3] (cbox path 1024x1024 spp 1024, pixel-tiled over N GPUs) rehearsed on ONE GPU, as SURVEY §8e
prescribes: the N interleaved tile shards of the frame are rendered one after another (forward and PRB backward), each
timed on its own.  What N GPUs would need per step is the slowest shard plus the two 16 MiB all-reduces, so
  load-balance efficiency = sum of the shard times / (N x slowest shard)
is the scaling efficiency the compute side allows (the RCCL exchange itself cannot be measured on a one-GPU box).
Also checks at FULL size that the shards add up: every pixel holds the same samples whoever renders it (the image union
is bit-identical when the shard cuts the sample range into the same chunks, else equal up to float re-association of the
per-pixel sum), gradient to float re-association."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zdr_amd.scenes import cbox_material_np
from zdr_amd.scenes import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=1024)
ap.add_argument("--ranks", default="2,4,8")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shard_balance.json"))
a = ap.parse_args()
scene = make_scene("path")
m = torch.from_numpy(cbox_material_np()).cuda()
W, spp = a.res, a.spp
ones = torch.ones((W, W, 4), device="cuda")

def timed(fn):
    ts = []
    for i in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts[1:])

full = scene.render_forward(m, (W, W), spp, 5)
g_full = torch.zeros_like(m); scene.render_backward(ones, g_full, m, (W, W), spp, 5)
t_full = (timed(lambda: scene.render_forward(m, (W, W), spp, 5)), timed(lambda: scene.render_backward(ones, torch.zeros_like(m), m, (W, W), spp, 5)))
out = {"workload": f"cbox path {W}x{W} spp={spp} (BASELINE configs[3]), interleaved 8x8-tile shards rendered one after another on ONE MI355X",
       "unsharded_ms": {"fwd": round(t_full[0] * 1e3, 2), "bwd": round(t_full[1] * 1e3, 2)}, "ranks": {}}
for N in [int(x) for x in a.ranks.split(",")]:
    union = torch.zeros_like(full); g_sum = torch.zeros_like(m)
    fwd, bwd = [], []
    for r in range(N):
        scene.render_forward(m, (W, W), spp, 5, tile_shard=(r, N), out=union)
        scene.render_backward(ones, g_sum, m, (W, W), spp, 5, tile_shard=(r, N))
        fwd.append(timed(lambda: scene.render_forward(m, (W, W), spp, 5, tile_shard=(r, N))))
        bwd.append(timed(lambda: scene.render_backward(ones, torch.zeros_like(m), m, (W, W), spp, 5, tile_shard=(r, N))))
    step = [f + b for f, b in zip(fwd, bwd)]
    rel = float((g_sum - g_full).abs().sum() / g_full.abs().sum())
    out["ranks"][N] = {"fwd_ms": [round(t * 1e3, 2) for t in fwd], "bwd_ms": [round(t * 1e3, 2) for t in bwd],
                       "slowest_step_ms": round(max(step) * 1e3, 2), "sum_of_steps_ms": round(sum(step) * 1e3, 2),
                       "load_balance_efficiency": round(sum(step) / (N * max(step)), 4),
                       "speedup_the_compute_side_allows": round((t_full[0] + t_full[1]) / max(step), 2),
                       "image_union_bit_identical": bool(torch.equal(union, full)),
                       "image_union_max_rel_diff": float(((union - full).abs() / (full.abs() + 1e-6)).max()),   # a shard cuts a pixel's samples into other chunks: re-association of the same sum
                       "gradient_rel_l1_vs_unsharded": rel}
    print(N, out["ranks"][N], flush=True)
scene.check()
json.dump(out, open(a.out, "w"), indent=1)
