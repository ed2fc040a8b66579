#!/usr/bin/env python3
"""What would an 8-wide node buy the walk of the 1 M-triangle scene?  (VERDICT r3 next #5 ii: "cost on paper, then build the cheaper".)
CPU only: the product's own BVH4 (zdr_debug_build_accel: the very node array the kernels read) is walked with an emulation of the device
loop for the rays of real waves — an 8x8 pixel tile's camera rays, then from their hit points one shadow ray to the light and one cosine
bounce ray each, i.e. what the 64 lanes of a wave trace for their first vertices — and every visit is logged.  An 8-wide tree is then
derived from the same nodes by absorbing node children into their parent, largest surface first, while the result has at most eight
children (the collapse a BVH8 builder would do on this binary tree; absorbed nodes are never visited, their children are tested by the
parent's visit), and the logged visits are re-counted.  Output: visits per ray and per WAVE (the loop runs until its slowest lane is
through), for both widths, and the VALU they cost with the instruction counts of the shipped loop (profiles/r4_bvh_walk_experiments.txt).
    python tools/bvh8_visits.py [--tiles 12] [--n 183]"""
import argparse
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from zdr_amd import _native, procedural
from zdr_amd.scenes import CBOX_CAMERA, cbox_models

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", type=int, default=12)
ap.add_argument("--n", type=int, default=183)
ap.add_argument("--res", type=int, default=1024)
a = ap.parse_args()

A = procedural.tessellated_cbox(cbox_models(), n=a.n)
V = A.verts[:, :3].astype(np.float32)
tri = np.ascontiguousarray(V[A.tris].reshape(-1, 9))            # identity transforms
n = tri.shape[0]
nodes = np.zeros((n, 16), np.float32); order = np.zeros(n, np.int32); isect = np.zeros((n, 12), np.float32)
nn = C.c_uint32(); se = C.c_uint32()
rc = _native.lib().zdr_debug_build_accel(tri.ctypes.data, n, _native.ACCEL_BVH, nodes.ctypes.data, nodes.shape[0], C.byref(nn), C.byref(se), order.ctypes.data, isect.ctypes.data)
assert rc == 0, _native.lib().zdr_last_error()
NN = nn.value
nodes = nodes[:NN]
W = nodes.view(np.uint32)
isect_off = 64 * NN
print(f"{n} triangles, {NN} BVH4 nodes")

# ---- decode once
org = nodes[:, 0:3].astype(np.float64); scl = np.stack([nodes[:, 3], nodes[:, 4], nodes[:, 5]], 1).astype(np.float64)
q = W[:, 6:12]                                                   # lx ly lz hx hy hz words
cw = W[:, 12:16].astype(np.int64)
kid_cnt = cw & 7
kid_node = np.where(kid_cnt == 0, (cw & ~15) // 64, -1)
kid_slot = np.where((kid_cnt >= 1) & (kid_cnt <= 2), ((cw & ~15) - isect_off) // 48, -1)
lo = np.zeros((NN, 4, 3)); hi = np.zeros((NN, 4, 3))
for k in range(4):
    for ax in range(3):
        lo[:, k, ax] = org[:, ax] + scl[:, ax] * ((q[:, ax] >> (8 * k)) & 255)
        hi[:, k, ax] = org[:, ax] + scl[:, ax] * ((q[:, 3 + ax] >> (8 * k)) & 255)
P = isect.astype(np.float64)


def walk(o, d, tmin, tmax, anyhit, log):
    """closest / any hit through the BVH4 in the device's order (nearest child first, the others stacked); logs (node, -1) / (-1, slot)"""
    inv = [1.0 / x if x != 0.0 else math.copysign(1e30, x if x != 0 else 1.0) for x in d]
    best, slot = tmax, -1
    stack = []
    cur = ("n", 0)
    while True:
        if cur[0] == "n":
            i = cur[1]
            log.append(i)
            ents = []
            for k in range(4):
                if kid_cnt[i, k] == 7:
                    continue
                tn, tf = tmin, best
                for ax in range(3):
                    t0 = (lo[i, k, ax] - o[ax]) * inv[ax]; t1 = (hi[i, k, ax] - o[ax]) * inv[ax]
                    if t0 > t1: t0, t1 = t1, t0
                    if t0 > tn: tn = t0
                    if t1 < tf: tf = t1
                if tn <= tf:
                    ents.append((tn, k))
            if ents:
                ents.sort()
                for tn, k in ents[:0:-1]:
                    stack.append(("n", int(kid_node[i, k])) if kid_cnt[i, k] == 0 else ("l", int(kid_slot[i, k]), int(kid_cnt[i, k])))
                tn, k = ents[0]
                cur = ("n", int(kid_node[i, k])) if kid_cnt[i, k] == 0 else ("l", int(kid_slot[i, k]), int(kid_cnt[i, k]))
                continue
        else:
            log.append(-1)
            for s in range(cur[1], cur[1] + cur[2]):
                r = P[s]
                nd = r[0] * d[0] + r[1] * d[1] + r[2] * d[2]
                if nd == 0.0: continue
                t = (r[3] - (r[0] * o[0] + r[1] * o[1] + r[2] * o[2])) / nd
                if not (tmin < t < best): continue
                p = (o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t)
                u = r[4] * p[0] + r[5] * p[1] + r[6] * p[2] + r[7]; v = r[8] * p[0] + r[9] * p[1] + r[10] * p[2] + r[11]
                if u >= 0 and v >= 0 and u + v <= 1:
                    best, slot = t, s
            if anyhit and slot >= 0:
                return slot, best
        if not stack:
            return slot, best
        cur = stack.pop()


# ---- the 8-wide collapse: which BVH4 nodes are absorbed into their parent
def area(i, k):
    e = hi[i, k] - lo[i, k]
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]

absorbed = np.zeros(NN, bool)
kids8 = {}                                                        # surviving node -> number of children of its 8-wide form
todo = [0]
while todo:
    i = todo.pop()
    kids = [(int(kid_node[i, k]), area(i, k)) if kid_cnt[i, k] == 0 else (None, 0.0) for k in range(4) if kid_cnt[i, k] != 7]
    while True:
        cand = [(ar, j) for j, ar in kids if j is not None and len(kids) - 1 + int((kid_cnt[j] != 7).sum()) <= 8]
        if not cand:
            break
        ar, j = max(cand)
        absorbed[j] = True
        kids.remove((j, ar))
        kids += [(int(kid_node[j, k]), area(j, k)) if kid_cnt[j, k] == 0 else (None, 0.0) for k in range(4) if kid_cnt[j, k] != 7]
    kids8[i] = len(kids)
    todo += [j for j, _ in kids if j is not None]
surv = NN - int(absorbed.sum())
print(f"8-wide collapse: {surv} nodes survive ({surv / NN:.2f} of the BVH4's), {np.mean(list(kids8.values())):.2f} of 8 child slots filled")

# ---- the rays of real waves
fov, co, ct, up = CBOX_CAMERA
co, ct, up = np.array(co, float), np.array(ct, float), np.array(up, float)
fwd = (ct - co) / np.linalg.norm(ct - co); right = np.cross(fwd, up); right /= np.linalg.norm(right); upp = np.cross(right, fwd)
tanf = math.tan(0.5 * fov)
light = V[A.tris[A.inst_tri_begin[1]:A.inst_tri_begin[2]]].astype(float)        # (2, 3, 3)
rng = np.random.default_rng(0)
stats = {"cam": [], "shadow": [], "bounce": []}
waves = []                                                                      # per wave: visits of the fused (shadow + bounce) loop per lane, 4-wide / 8-wide
for t in range(a.tiles):
    tx, ty = int(rng.integers(8, a.res // 8 - 8)), int(rng.integers(8, a.res // 8 - 8))
    lanes4, lanes8 = [], []
    for lane in range(64):
        x, y = tx * 8 + (lane & 7) + rng.random(), ty * 8 + (lane >> 3) + rng.random()
        px, py = (2.0 / a.res * x - 1.0) * tanf, (2.0 / a.res * y - 1.0) * tanf
        d = right * px - upp * py + fwd; d /= np.linalg.norm(d)
        log = []
        slot, tt = walk(co, d, 0.0, 1e30, False, log)
        stats["cam"].append(log)
        if slot < 0:
            continue
        p = co + d * tt
        T = tri[order[slot]].reshape(3, 3).astype(float)
        ng = np.cross(T[1] - T[0], T[2] - T[0]); ng /= np.linalg.norm(ng)
        if np.dot(ng, d) > 0: ng = -ng
        p = p + ng * 1e-4
        # shadow ray to a uniform point on the light
        Lt = light[int(rng.integers(0, light.shape[0]))]; u, v = rng.random(2)
        if u + v > 1: u, v = 1 - u, 1 - v
        ql = Lt[0] + u * (Lt[1] - Lt[0]) + v * (Lt[2] - Lt[0])
        ds = ql - p; dist = np.linalg.norm(ds); ds /= dist
        ls = []
        walk(p, ds, 1e-4, 0.9999 * dist, True, ls)
        # cosine bounce
        r1, r2 = rng.random(2); ph = 2 * math.pi * r2; sx, sy, sz = math.sqrt(r1) * math.cos(ph), math.sqrt(r1) * math.sin(ph), math.sqrt(1 - r1)
        tng = np.cross(ng, [1, 0, 0] if abs(ng[0]) < 0.9 else [0, 1, 0]); tng /= np.linalg.norm(tng); bt = np.cross(ng, tng)
        db = tng * sx + bt * sy + ng * sz
        lb = []
        walk(p, db, 0.0, 1e30, False, lb)
        stats["shadow"].append(ls); stats["bounce"].append(lb)
        both = ls + lb
        lanes4.append(len(both))
        lanes8.append(sum(1 for i in both if i < 0 or not absorbed[i]))
    if lanes4:
        waves.append((max(lanes4), np.mean(lanes4), max(lanes8), np.mean(lanes8)))

def summary(name, logs):
    n4 = np.mean([sum(1 for i in l if i >= 0) for l in logs]); lf = np.mean([sum(1 for i in l if i < 0) for l in logs])
    n8 = np.mean([sum(1 for i in l if i >= 0 and not absorbed[i]) for l in logs])
    print(f"{name:7s} rays: {len(logs):5d}   node visits per ray {n4:6.2f} (4-wide) -> {n8:6.2f} (8-wide, x {n8 / n4:.2f})   leaf visits {lf:5.2f}")
    return n4, n8, lf
for k in ("cam", "shadow", "bounce"):
    summary(k, stats[k])
w = np.array(waves)
print(f"{len(waves)} waves (first vertices of an 8x8 tile, shadow + bounce ray per lane, fused loop): trips = visits of the slowest lane")
print(f"  4-wide: mean lane {w[:, 1].mean():.1f} visits, wave {w[:, 0].mean():.1f} trips (lanes busy {w[:, 1].mean() / w[:, 0].mean():.2f} of the trips)")
print(f"  8-wide: mean lane {w[:, 3].mean():.1f} visits, wave {w[:, 2].mean():.1f} trips (lanes busy {w[:, 3].mean() / w[:, 2].mean():.2f}); trips x {w[:, 2].mean() / w[:, 0].mean():.2f}")
# VALU per trip with both bodies issued (shipped loop: node ~100, leaf ~55, control ~10; an 8-wide node: 8 slab tests 112 + hit mask / octant order ~35)
c4 = w[:, 0].mean() * (100 + 55 + 10); c8 = w[:, 2].mean() * (147 + 55 + 10)
print(f"  VALU per wave and vertex, node + leaf + control issued in every trip: 4-wide {c4:.0f}, 8-wide {c8:.0f} (x {c8 / c4:.2f})")
