#!/usr/bin/env python3
"""Would SUBTREE STEALING shorten the walk of the 1 M-triangle scene?  (CPU only; a costing for the next step of DESIGN.md 8 item 1.)
The wave's trip count is set by its slowest lane (67 trips for 40 visits of the mean lane: lanes busy 0.59, tools/bvh8_visits.py), and
every remedy that delays a lane lengthens exactly that walk (profiles/r4_exp_leaf_turns.patch).  The traversal stack of a lane lives in
the wave's LDS, so a lane that is through could take the BOTTOM entry (the farthest subtree still waiting) of the lane with the most
entries, walk it with a copy of that lane's ray and merge the result (min over (t, slot) for a closest-hit ray, OR for an any-hit ray).
This script walks the product's own BVH4 (zdr_debug_build_accel) in lock step for the rays of real waves — per lane a shadow ray, then a
bounce ray, as the fused loop of csrc/accel.h does — with and without stealing and reports trips, visits (a thief prunes with the hit
distance it took along, not with what the owner finds later) and steals.  A steal costs the thief `--steal-trips` trips without a visit.
    python tools/steal_sim.py [--tiles 24] [--n 183] [--steal-trips 1] [--share-best]"""
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
ap.add_argument("--tiles", type=int, default=24)
ap.add_argument("--n", type=int, default=183)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--steal-trips", type=int, default=1)
ap.add_argument("--period", type=int, default=1, help="lanes may steal only in every period-th trip")
ap.add_argument("--lds-entries", type=int, default=0, help="> 0: only the first N stack entries of a lane (the LDS part) can be taken")
ap.add_argument("--min-victim", type=int, default=2, help="a lane is a victim only with at least this many stack entries")
ap.add_argument("--share-best", action="store_true", help="all lanes working for a ray see its best hit distance (an LDS cell per lane)")
a = ap.parse_args()

A = procedural.tessellated_cbox(cbox_models(), n=a.n)
V = A.verts[:, :3].astype(np.float32)
tri = np.ascontiguousarray(V[A.tris].reshape(-1, 9))
n = tri.shape[0]
nodes = np.zeros((n, 16), np.float32); order = np.zeros(n, np.int32); isect = np.zeros((n, 12), np.float32)
nn = C.c_uint32(); se = C.c_uint32()
rc = _native.lib().zdr_debug_build_accel(tri.ctypes.data, n, _native.ACCEL_BVH, nodes.ctypes.data, nodes.shape[0], C.byref(nn), C.byref(se), order.ctypes.data, isect.ctypes.data)
assert rc == 0, _native.lib().zdr_last_error()
NN = nn.value
nodes = nodes[:NN]
Wd = nodes.view(np.uint32)
isect_off = 64 * NN
print(f"{n} triangles, {NN} BVH4 nodes")
org = nodes[:, 0:3].astype(np.float64); scl = np.stack([nodes[:, 3], nodes[:, 4], nodes[:, 5]], 1).astype(np.float64)
q = Wd[:, 6:12]
cw = Wd[:, 12:16].astype(np.int64)
kid_cnt = cw & 7
kid_node = np.where(kid_cnt == 0, (cw & ~15) // 64, -1)
kid_slot = np.where((kid_cnt >= 1) & (kid_cnt <= 2), ((cw & ~15) - isect_off) // 48, -1)
lo = np.zeros((NN, 4, 3)); hi = np.zeros((NN, 4, 3))
for k in range(4):
    for ax in range(3):
        lo[:, k, ax] = org[:, ax] + scl[:, ax] * ((q[:, ax] >> (8 * k)) & 255)
        hi[:, k, ax] = org[:, ax] + scl[:, ax] * ((q[:, 3 + ax] >> (8 * k)) & 255)
P = isect.astype(np.float64)
lo_l, hi_l, kc_l, kn_l, ks_l = lo.tolist(), hi.tolist(), kid_cnt.tolist(), kid_node.tolist(), kid_slot.tolist()
P_l = P.tolist()


class Ray:
    """One ray of a lane; `best` / `slot` are the merged result of everybody who worked for it."""
    def __init__(self, o, d, tmin, tmax, anyhit):
        self.o, self.d, self.tmin, self.anyhit = o, d, tmin, anyhit
        self.inv = [1.0 / x if x != 0.0 else 1e30 for x in d]
        self.best, self.slot, self.done, self.workers = tmax, -1, False, 0


class Task:
    """A lane's share of a ray: its own stack and the hit distance it prunes with."""
    def __init__(self, ray, first, best):
        self.ray, self.cur, self.stack, self.best = ray, first, [], best
        ray.workers += 1


def child_item(i, k):
    return ("n", kn_l[i][k]) if kc_l[i][k] == 0 else ("l", ks_l[i][k], kc_l[i][k])


def step(t, share):
    """one visit; returns False when the task has nothing left"""
    r = t.ray
    if r.done:
        return False
    if share and r.best < t.best:
        t.best = r.best
    o, d, inv = r.o, r.d, r.inv
    if t.cur[0] == "n":
        i = t.cur[1]
        ents = []
        for k in range(4):
            if kc_l[i][k] == 7:
                continue
            tn, tf = r.tmin, t.best
            l3, h3 = lo_l[i][k], hi_l[i][k]
            for ax in range(3):
                t0 = (l3[ax] - o[ax]) * inv[ax]; t1 = (h3[ax] - o[ax]) * inv[ax]
                if t0 > t1: t0, t1 = t1, t0
                if t0 > tn: tn = t0
                if t1 < tf: tf = t1
            if tn <= tf:
                ents.append((tn, k))
        if ents:
            ents.sort()
            for tn, k in ents[:0:-1]:
                t.stack.append(child_item(i, k))
            t.cur = child_item(i, ents[0][1])
            return True
    else:
        for s in range(t.cur[1], t.cur[1] + t.cur[2]):
            rec = P_l[s]
            nd = rec[0] * d[0] + rec[1] * d[1] + rec[2] * d[2]
            if nd == 0.0: continue
            tt = (rec[3] - (rec[0] * o[0] + rec[1] * o[1] + rec[2] * o[2])) / nd
            if not (r.tmin < tt < t.best): continue
            p = (o[0] + d[0] * tt, o[1] + d[1] * tt, o[2] + d[2] * tt)
            u = rec[4] * p[0] + rec[5] * p[1] + rec[6] * p[2] + rec[7]; v = rec[8] * p[0] + rec[9] * p[1] + rec[10] * p[2] + rec[11]
            if u >= 0 and v >= 0 and u + v <= 1:
                t.best = tt
                if tt < r.best or (tt == r.best and s < r.slot): r.best, r.slot = tt, s
        if r.anyhit and r.slot >= 0:
            r.done = True
            return False
    if not t.stack:
        return False
    t.cur = t.stack.pop()
    return True


def run_wave(lanes, steal):
    """lanes: per lane a list of rays in the order it walks them.  Returns (trips, visits, steals, busy lane-trips)."""
    todo = [list(rs) for rs in lanes]
    task = [None] * len(lanes)
    wait = [0] * len(lanes)
    trips = visits = steals = 0
    def next_own(l):
        while todo[l]:
            r = todo[l].pop(0)
            task[l] = Task(r, ("n", 0), r.best)
            return True
        return False
    for l in range(len(lanes)):
        next_own(l)
    while any(t is not None for t in task) or any(w > 0 for w in wait):
        trips += 1
        for l in range(len(lanes)):
            if wait[l] > 0:
                wait[l] -= 1
                continue
            t = task[l]
            if t is None:
                continue
            visits += 1
            if not step(t, a.share_best):
                t.ray.workers -= 1
                task[l] = None
                next_own(l)
        if steal and trips % a.period == 0:
            for l in range(len(lanes)):
                if task[l] is not None or wait[l] > 0 or todo[l]:
                    continue
                victim = max(range(len(lanes)), key=lambda v: len(task[v].stack) if task[v] is not None and not task[v].ray.done else -1)
                tv = task[victim]
                if tv is None or tv.ray.done or len(tv.stack) < a.min_victim:
                    continue
                item = tv.stack.pop(0)                                 # the bottom entry: the farthest subtree still waiting
                task[l] = Task(tv.ray, item, tv.best)
                wait[l] = a.steal_trips
                steals += 1
    return trips, visits, steals


fov, co, ct, up = CBOX_CAMERA
co, ct, up = np.array(co, float), np.array(ct, float), np.array(up, float)
fwd = (ct - co) / np.linalg.norm(ct - co); right = np.cross(fwd, up); right /= np.linalg.norm(right); upp = np.cross(right, fwd)
tanf = math.tan(0.5 * fov)
light = V[A.tris[A.inst_tri_begin[1]:A.inst_tri_begin[2]]].astype(float)
rng = np.random.default_rng(0)


def wave_rays(tx, ty):
    """the shadow and bounce rays of the first vertices of an 8 x 8 pixel tile (as tools/bvh8_visits.py)"""
    out = []
    for lane in range(64):
        x, y = tx * 8 + (lane & 7) + rng.random(), ty * 8 + (lane >> 3) + rng.random()
        px, py = (2.0 / a.res * x - 1.0) * tanf, (2.0 / a.res * y - 1.0) * tanf
        d = right * px - upp * py + fwd; d /= np.linalg.norm(d)
        cam = Ray(co.tolist(), d.tolist(), 0.0, 1e30, False)
        t = Task(cam, ("n", 0), cam.best)
        while step(t, False):
            pass
        if cam.slot < 0:
            out.append(None); continue
        p = co + d * cam.best
        T = tri[order[cam.slot]].reshape(3, 3).astype(float)
        ng = np.cross(T[1] - T[0], T[2] - T[0]); ng /= np.linalg.norm(ng)
        if np.dot(ng, d) > 0: ng = -ng
        p = p + ng * 1e-4
        Lt = light[int(rng.integers(0, light.shape[0]))]; u, v = rng.random(2)
        if u + v > 1: u, v = 1 - u, 1 - v
        ql = Lt[0] + u * (Lt[1] - Lt[0]) + v * (Lt[2] - Lt[0])
        ds = ql - p; dist = np.linalg.norm(ds); ds /= dist
        r1, r2 = rng.random(2); ph = 2 * math.pi * r2; sx, sy, sz = math.sqrt(r1) * math.cos(ph), math.sqrt(r1) * math.sin(ph), math.sqrt(1 - r1)
        tng = np.cross(ng, [1, 0, 0] if abs(ng[0]) < 0.9 else [0, 1, 0]); tng /= np.linalg.norm(tng); bt = np.cross(ng, tng)
        db = tng * sx + bt * sy + ng * sz
        out.append((p.tolist(), ds.tolist(), 0.9999 * dist, db.tolist()))
    return out


tot = {False: [0, 0, 0], True: [0, 0, 0]}
same = 0; rays = 0
for tile in range(a.tiles):
    tx, ty = int(rng.integers(8, a.res // 8 - 8)), int(rng.integers(8, a.res // 8 - 8))
    spec = wave_rays(tx, ty)
    res = {}
    for steal in (False, True):
        lanes = [[] if s is None else [Ray(s[0], s[1], 1e-4, s[2], True), Ray(s[0], s[3], 0.0, 1e30, False)] for s in spec]
        tr, vi, st = run_wave(lanes, steal)
        tot[steal][0] += tr; tot[steal][1] += vi; tot[steal][2] += st
        res[steal] = [(r.slot >= 0) if r.anyhit else (r.slot, r.best) for rs in lanes for r in rs]
    rays += len(res[False]); same += sum(x == y for x, y in zip(res[False], res[True]))
    print(f"tile {tile}: trips {tot[False][0]} -> {tot[True][0]}", flush=True)
w = a.tiles
print(f"{w} waves, {rays} rays; answers identical with and without stealing: {same} of {rays}")
print(f"  as shipped : {tot[False][0] / w:6.1f} trips per wave, {tot[False][1] / w / 64:5.1f} visits per lane (lanes busy {tot[False][1] / (64.0 * tot[False][0]):.2f})")
print(f"  stealing   : {tot[True][0] / w:6.1f} trips per wave, {tot[True][1] / w / 64:5.1f} visits per lane (lanes busy {tot[True][1] / (64.0 * tot[True][0]):.2f}), "
      f"{tot[True][2] / w:.1f} steals per wave; trips x {tot[True][0] / tot[False][0]:.2f}, visits x {tot[True][1] / tot[False][1]:.2f}")
