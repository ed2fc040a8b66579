#!/usr/bin/env python3
"""would per-origin-primitive candidate masks cut the 9-pair walks of the cbox path kernels?

The camera rays of a work item share a tile, so ONE mask (k_tile_masks) serves the whole wave and the pair loop — wave-uniform,
its operands in SGPRs — skips 6-7 of the 9 pairs.  A bounce or shadow ray leaves a surface point on some primitive P; a pair
of primitives that lies wholly behind P's plane cannot be hit by it, which gives a conservative mask per ORIGIN PRIMITIVE.  But
the pair loop is wave-uniform: a pair is skipped only if EVERY lane's mask excludes it, i.e. the loop runs over the OR of the
64 lanes' masks.  This script measures that OR on the real workload, with the oracle's per-path traces (CPU only):

  * mask[P] = pairs with a corner more than eps in front of P's plane (P's own pair and everything coplanar or behind drop out);
  * the flat loop of k_path is emulated per work item (8x8 tile x 16 samples, lanes = workers, FIFO of parked camera-ray vertices
    in (sample, pixel) order, a lane takes the next parked vertex when its path has ended);
  * per trip: the pairs in the OR of the masks of the lanes that trace a continuation ray / a shadow ray.

Output: average pairs per walk a wave would still test, per-lane average for comparison (what a per-lane skip could reach if
lanes did not share the loop).  Result (profiles/r4_origin_masks.txt): see the bottom lines it prints."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from conftest import CBOX_CAMERA, cbox_material_np, cbox_models
from path_trace import Trace, all_queries
from test_bvh_emulation import world_triangles
from zdr_amd import _native, geometry

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--tiles", type=int, default=48, help="work items sampled over the image")
a = ap.parse_args()

A = geometry.assemble(cbox_models())
tri = world_triangles(A)                                         # (ntris, 3, 3) world-space corners, input order
n = tri.shape[0]
order = np.zeros(n, np.int32); isect = np.zeros((n, 12), np.float32); nq = C.c_uint32(); npar = C.c_uint32()
dummy = np.zeros((8, 16), np.float32)
rc = _native.lib().zdr_debug_build_accel(np.ascontiguousarray(tri.reshape(-1, 9)).ctypes.data, n, _native.ACCEL_BRUTE, dummy.ctypes.data, 8,
                                         C.byref(nq), C.byref(npar), order.ctypes.data, isect.ctypes.data)
assert rc == 0
Q = nq.value                                                     # quads: slots 2q, 2q + 1; single triangles behind them
nprim = Q + (n - 2 * Q)
npairs = (nprim + 1) // 2
prim_tris = [[int(order[2 * q]), int(order[2 * q + 1])] if q < Q else [int(order[q + Q])] for q in range(nprim)]
pair_tris = [sum((prim_tris[p] for p in (2 * k, 2 * k + 1) if p < nprim), []) for k in range(npairs)]
print(f"{n} triangles -> {Q} quads + {n - 2 * Q} triangles = {nprim} primitives, {npairs} pairs")

# mask[t]: pairs that have a corner in front of triangle t's plane
p0, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
ng = np.cross(e1, e2); ng /= np.linalg.norm(ng, axis=1, keepdims=True)
eps = 1e-5 * np.abs(tri).max()
mask = np.zeros(n, np.int64)
for t in range(n):
    for k, ts in enumerate(pair_tris):
        h = ((tri[ts].reshape(-1, 3) - p0[t]) @ ng[t]).max()
        if h > eps:
            mask[t] |= 1 << k
per_tri = np.array([bin(m).count("1") for m in mask])
print("pairs in front of a triangle's plane: min %d, mean %.2f, max %d of %d" % (per_tri.min(), per_tri.mean(), per_tri.max(), npairs))

S = oracle.OracleScene.from_arrays(A)
mat = cbox_material_np()
W, spp = a.res, a.spp
p = oracle.make_params("path", W, W, spp, 1, CBOX_CAMERA, mat.shape[:2], use_tent=True)
inst_begin = A.inst_tri_begin
rng = np.random.default_rng(0)
tiles = [(int(x) * 8, int(y) * 8) for x, y in zip(rng.integers(0, W // 8, a.tiles), rng.integers(0, W // 8, a.tiles))]
tot = {"trips": 0, "cont_walks": 0, "cont_or": 0, "cont_lane": 0.0, "cont_lanes": 0, "shad_walks": 0, "shad_or": 0, "shad_lane": 0.0, "shad_lanes": 0, "full": 0}
for (x0, y0) in tiles:
    q = all_queries(8, 8, spp, x0, y0)                           # (pixel-major, sample-minor)
    tr = Trace(S.path_dump(p, mat, q))
    # FIFO order of the refill: sample index outer, pixel inner
    fifo = sorted(range(q.shape[0]), key=lambda i: (q[i, 2], (q[i, 1] - y0) * 8 + (q[i, 0] - x0)))
    fifo = [i for i in fifo if tr.nvert[i] > 0]                  # paths that shade at least one vertex are parked
    lanes = [None] * 64                                          # (path, vertex index)
    head = 0
    while True:
        for l in range(64):
            if lanes[l] is None and head < len(fifo):
                lanes[l] = (fifo[head], 0); head += 1
        if all(x is None for x in lanes):
            break
        tot["trips"] += 1
        m_cont = m_shad = 0
        c_l = s_l = 0
        for l in range(64):
            if lanes[l] is None:
                continue
            i, k = lanes[l]
            t = int(inst_begin[tr.inst[i, k]] + tr.prim[i, k])  # origin triangle (input order)
            m_shad |= int(mask[t]); s_l += 1; tot["shad_lane"] += per_tri[t]
            went_on = (tr.flags[i, k] >> 1) & 1
            if went_on:
                m_cont |= int(mask[t]); c_l += 1; tot["cont_lane"] += per_tri[t]
            lanes[l] = (i, k + 1) if (went_on and k + 1 < tr.nvert[i]) else None
        if s_l:
            tot["shad_walks"] += 1; tot["shad_or"] += bin(m_shad).count("1"); tot["shad_lanes"] += s_l
        if c_l:
            tot["cont_walks"] += 1; tot["cont_or"] += bin(m_cont).count("1"); tot["cont_lanes"] += c_l
            tot["full"] += bin(m_cont).count("1") == npairs

print(f"{len(tiles)} work items of 8x8 pixels x {spp} samples at {W}x{W}: {tot['trips']} trips, {tot['shad_lanes'] / tot['trips']:.1f} lanes shading per trip")
print(f"continuation walks: the wave's OR keeps {tot['cont_or'] / tot['cont_walks']:.2f} of {npairs} pairs ({tot['full'] / tot['cont_walks']:.1%} of the walks keep all); a lane alone would need {tot['cont_lane'] / tot['cont_lanes']:.2f}")
print(f"shadow walks (front mask for every lane: an upper bound on what could be skipped): the wave's OR keeps {tot['shad_or'] / tot['shad_walks']:.2f} of {npairs} pairs; a lane alone {tot['shad_lane'] / tot['shad_lanes']:.2f}")
