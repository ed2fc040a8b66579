#!/usr/bin/env python3
"""BASELINE configs[2] at FULL size (cbox, path, 512x512, spp 256): HIP forward image and PRB gradient against the
CPU oracle on the same seed, with the oracle's own IEEE-vs-FMA difference beside them as the fp32 floor."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from conftest import cbox_material_np, cbox_models
from gpu_util import make_scene, oracle_params, image_diff_stats
from zdr_amd import geometry

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--shards", type=int, default=1, help="render the GPU side as this many interleaved tile shards (BASELINE configs[3]) and compare their union / sum")
ap.add_argument("--scene", default="cbox", choices=["cbox", "tess1m"], help="tess1m: BASELINE configs[4], the 1,004,672-triangle tessellated Cornell box (BVH)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_size_parity.json"))
a = ap.parse_args()
W, spp, seed = a.res, a.spp, 7
mat = cbox_material_np()
if a.scene == "tess1m":
    from zdr_amd.scenes import tess1m_arrays
    A = tess1m_arrays()
    scene = make_scene("path", arrays=A)
    assert scene.info()["accel"] == "bvh"
else:
    A = geometry.assemble(cbox_models())
    scene = make_scene("path")
S, Sf = oracle.OracleScene.from_arrays(A), oracle.OracleScene.from_arrays(A, variant="fma")
m = torch.from_numpy(mat).cuda()
out = {"config": f"{'1,004,672-triangle tessellated cbox' if a.scene == 'tess1m' else 'cbox'} path {W}x{W} spp {spp} seed {seed}, cboxd/cboxr textures" + (f", GPU side = union of {a.shards} interleaved tile shards" if a.shards > 1 else "")}
t = time.time()
if a.shards > 1:
    img_t = torch.zeros((W, W, 4), device="cuda")
    for r in range(a.shards): scene.render_forward(m, (W, W), spp, seed, tile_shard=(r, a.shards), out=img_t)
    img = img_t.cpu().numpy()
else:
    img = scene.render_forward(m, (W, W), spp, seed).cpu().numpy()
out["gpu_forward_s"] = round(time.time() - t, 3)
p = oracle_params(scene, W, W, spp, seed, mat.shape[:2])
print("gpu forward done; oracle forward ...", flush=True)
t = time.time(); ref = S.render_forward(p, mat); out["oracle_forward_s"] = round(time.time() - t, 1)
print(f"oracle forward {out['oracle_forward_s']} s; fma build ...", flush=True)
flo = Sf.render_forward(p, mat)
print("forward compared; backward ...", flush=True)
out["image"] = {"gpu_vs_oracle": image_diff_stats(img[..., :3], ref[..., :3]), "oracle_fma_vs_ieee": image_diff_stats(flo[..., :3], ref[..., :3])}
cot = np.random.default_rng(1).uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
g = torch.zeros_like(m)
for r in range(a.shards): scene.render_backward(torch.from_numpy(cot).cuda(), g, m, (W, W), spp, seed, tile_shard=(r, a.shards) if a.shards > 1 else None)
g = g.cpu().numpy().astype(np.float64)
pb = oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2])
print("gpu backward done; oracle backward ...", flush=True)
t = time.time(); gref = S.render_backward(pb, cot, mat).astype(np.float64); out["oracle_backward_s"] = round(time.time() - t, 1)
print(f"oracle backward {out['oracle_backward_s']} s; fma build ...", flush=True)
gflo = Sf.render_backward(pb, cot, mat).astype(np.float64)
def gstats(a, b):
    return {"rel_l1": float(np.abs(a - b).sum() / np.abs(b).sum()), "sum_rel": float(abs(a.sum() - b.sum()) / abs(b.sum())),
            "max_abs_over_max": float(np.abs(a - b).max() / np.abs(b).max()), "nnz": int((a != 0).sum()), "nnz_ref": int((b != 0).sum())}
out["gradient"] = {"gpu_vs_oracle": gstats(g, gref), "oracle_fma_vs_ieee": gstats(gflo, gref)}
print(json.dumps(out, indent=1))
json.dump(out, open(a.out, "w"), indent=1)
