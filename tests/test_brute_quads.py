"""CPU: which triangles the brute-force accel merges into planar convex quads (zdr_api.cpp, find_quads) — the host
decision behind the quad walk of csrc/accel.h, read back through the host-only zdr_debug_build_accel."""
import ctypes as C

import numpy as np

from zdr_amd import _native


def build(tris):
    tri = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n = tri.shape[0]
    nq, se = C.c_uint32(0), C.c_uint32(0)
    order = np.zeros(n, np.int32); isect = np.zeros((n, 12), np.float32); nodes = np.zeros((1, 16), np.float32)
    rc = _native.lib().zdr_debug_build_accel(tri.ctypes.data, n, _native.ACCEL_BRUTE, nodes.ctypes.data, 1, C.byref(nq), C.byref(se),
                                              order.ctypes.data, isect.ctypes.data)
    assert rc == 0, _native.lib().zdr_last_error()
    global NPAR
    NPAR = se.value                      # parallelograms among the quads (they come first)
    return nq.value, order, isect


def quad(p0, p1, p2, p3):
    """two triangles (p0 p1 p2), (p0 p2 p3) of the quad p0 p1 p2 p3"""
    return [[*p0, *p1, *p2], [*p0, *p2, *p3]]


def test_cornell_box_is_fifteen_quads_and_two_triangles(cbox_arrays):
    A = cbox_arrays
    tris = A.verts[A.tris][:, :, :3]
    nq, order, isect = build(tris)
    assert nq == 15 and NPAR == 11 and sorted(order.tolist()) == list(range(32))
    P = tris.reshape(32, 3, 3).astype(np.float64)
    for q in range(nq):
        a, b = order[2 * q], order[2 * q + 1]
        shared = [p for p in P[a].tolist() if p in P[b].tolist()]
        assert len(shared) == 2                                   # the two triangles of a quad share an edge
        for slot, t in ((2 * q, a), (2 * q + 1, b)):              # whose corners score u + v = 1 in BOTH records: it is their w = 0 edge
            U, V = isect[slot, 4:8].astype(np.float64), isect[slot, 8:12].astype(np.float64)
            for s in shared:
                u, v = U[:3] @ s + U[3], V[:3] @ s + V[3]
                assert abs(u + v - 1.0) < 1e-5 and min(u, v) > -1e-5
            apex = [p for p in P[t].tolist() if p not in shared][0]
            assert abs(U[:3] @ apex + U[3]) < 1e-5 and abs(V[:3] @ apex + V[3]) < 1e-5
    # the point of it: four outer edge functions >= 0 <=> inside one of the two triangles (checked on random points of each quad's plane)
    rng = np.random.default_rng(0)
    for q in range(nq):
        a, b = order[2 * q], order[2 * q + 1]
        pts = np.concatenate([P[a], P[b]]); lo, hi = pts.min(0), pts.max(0)
        n = np.cross(P[a][1] - P[a][0], P[a][2] - P[a][0]); n /= np.linalg.norm(n)
        x = rng.uniform(lo - 0.2, hi + 0.2, (2000, 3)); x -= ((x - P[a][0]) @ n)[:, None] * n
        def bary(slot):
            U, V = isect[slot, 4:8].astype(np.float64), isect[slot, 8:12].astype(np.float64)
            u, v = x @ U[:3] + U[3], x @ V[:3] + V[3]
            return u, v, 1.0 - u - v
        ua, va, wa = bary(2 * q); ub, vb, wb = bary(2 * q + 1)
        in_quad = np.minimum(np.minimum(ua, va), np.minimum(ub, vb)) >= 0
        in_tris = ((ua >= 0) & (va >= 0) & (wa >= 0)) | ((ub >= 0) & (vb >= 0) & (wb >= 0))
        margin = np.minimum(np.abs(np.stack([ua, va, wa, ub, vb, wb])).min(0), 1.0) > 1e-4     # away from the edges
        assert (in_quad == in_tris)[margin].all() and in_quad.any() and (~in_quad).any()


def test_only_planar_convex_pairs_merge():
    flat = quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0))
    assert build(flat)[0] == 1 and NPAR == 1
    assert build(quad((0, 0, 0), (1, 0, 0), (1.2, 1, 0), (0, 1, 0)))[0] == 1 and NPAR == 0      # a trapezium merges, but not as a parallelogram
    bent = quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 1e-3))              # fourth corner out of the plane
    assert build(bent)[0] == 0
    dart = quad((0, 0, 0), (1, 0, 0), (0.2, 0.2, 0), (0, 1, 0))             # reflex corner at the shared edge
    assert build(dart)[0] == 0
    folded = [[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 0, 1, 0, 0, 0.3, 0.3, 0]]  # both on the same side of the shared edge
    assert build(folded)[0] == 0
    twice = [flat[0], flat[0]]                                              # the same triangle twice
    assert build(twice)[0] == 0
    apart = [flat[0], [c + 5 for c in flat[1]]]                             # no shared edge
    assert build(apart)[0] == 0
    strip = quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)) + quad((1, 0, 0), (2, 0, 0), (2, 1, 0), (1, 1, 0)) + [[5, 5, 5, 6, 5, 5, 5, 6, 5]]
    nq, order, _ = build(strip)
    assert nq == 2 and sorted(order[:4].tolist()) == [0, 1, 2, 3] and order[4] == 4      # quads first, the single triangle after them


def test_one_primitive_per_triangle_when_switched_off(monkeypatch):
    monkeypatch.setenv("ZDR_NO_QUADS", "1")
    nq, order, _ = build(quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)))
    assert nq == 0 and order.tolist() == [0, 1]


def emulate_quad_walk(nq, order, isect, rays):
    """float32 NumPy mirror of BruteAccel::closest + brute_resolve (csrc/accel.h): primitives = quads (slots 2q, 2q + 1:
    plane of the first triangle, u and v of both) then single triangles (u, v, w, w); nearest primitive, then the
    triangle of the quad by the sign of the first one's w.  Returns (input triangle or -1, t)."""
    f = np.float32
    n = order.shape[0]
    nprim = n - nq
    o, d, tmin, tmax = rays[:, 0:3], rays[:, 4:7], rays[:, 3], rays[:, 7]
    best_t = tmax.copy(); best_p = np.full(rays.shape[0], -1)
    def edge(rec, p):
        return (p @ rec[:3] + rec[3]).astype(f)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for q in range(nprim):
            a = 2 * q if q < nq else q + nq
            N, U, V = isect[a, 0:4], isect[a, 4:8], isect[a, 8:12]
            t = ((N[3] - (o @ N[:3]).astype(f)) / (d @ N[:3]).astype(f)).astype(f)
            p = (o + d * t[:, None]).astype(f)
            u, v = edge(U, p), edge(V, p)
            if q < 2 * (NPAR // 2):                      # pairs of parallelograms: the second triangle's outer edges are 1 - u, 1 - v
                e3, e4 = (f(1.0) - u).astype(f), (f(1.0) - v).astype(f)
            elif q < nq:
                e3, e4 = edge(isect[a + 1, 4:8], p), edge(isect[a + 1, 8:12], p)
            else:
                e3 = e4 = (f(1.0) - (u + v)).astype(f)
            c = np.minimum(np.minimum(u, v), np.minimum(e3, e4))
            ok = (t > tmin) & (t < best_t) & (c >= 0)
            best_t = np.where(ok, t, best_t); best_p = np.where(ok, q, best_p)
    tri = np.full(rays.shape[0], -1)
    hit = best_p >= 0
    slot = np.where(best_p < nq, 2 * best_p, best_p + nq)
    p = (o + d * best_t[:, None]).astype(f)
    for i in np.nonzero(hit)[0]:
        s = slot[i]
        if best_p[i] < nq:
            u, v = p[i] @ isect[s, 4:7] + isect[s, 7], p[i] @ isect[s, 8:11] + isect[s, 11]
            if 1.0 - (u + v) < 0: s += 1
        tri[i] = order[s]
    return tri, best_t


def test_quad_walk_emulation_matches_the_oracle(cbox_arrays, cbox_oracle):
    """The walk over quads finds the triangles the oracle's per-triangle brute force finds (same tolerance as the GPU test)."""
    from gpu_util import random_rays
    A = cbox_arrays
    nq, order, isect = build(A.verts[A.tris][:, :, :3])
    rays = random_rays(20000, (-3, 0, -5.5), (2.5, 5.2, 6), seed=9)
    tri, t = emulate_quad_walk(nq, order, isect, rays)
    rip, rbt = cbox_oracle.trace_closest(rays)
    ref = np.where(rip[:, 0] >= 0, A.inst_tri_begin[np.maximum(rip[:, 0], 0)] + rip[:, 1], -1)
    assert (tri != ref).mean() < 2e-4
    both = (tri >= 0) & (tri == ref)
    terr = np.abs(t[both] - rbt[both, 2]) / (1e-5 * np.abs(rbt[both, 2]) + 5e-6)
    assert both.mean() > 0.3 and (terr > 1).mean() < 2e-4 and terr.max() < 50


# ---- the shadow walk's pair mask: which triangles can never lie between a surface point and a point of a light

def never_occluders(tris, is_light):
    tri = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    light = np.ascontiguousarray(is_light, np.uint8)
    out = np.zeros(tri.shape[0], np.uint8)
    rc = _native.lib().zdr_debug_never_occluders(tri.ctypes.data, tri.shape[0], light.ctypes.data, out.ctypes.data)
    assert rc == 0, _native.lib().zdr_last_error()
    return out.astype(bool)


def test_the_walls_of_the_cornell_box_never_occlude_a_shadow_segment(cbox_arrays):
    """Back wall, floor, left and right wall (8 triangles) support the scene from outside and the ceiling light keeps > 2 units from
    them: no segment from a surface point to a point of the light can meet them inside (1e-4, 0.9999 dist).  The CEILING does not
    qualify: the light hangs 7 mm below it, too close to rule out a hit by rounding alone — it stays in the walk, like the box
    faces and the light itself.  3 of the 9 pairs of the pair walk drop out of every shadow walk."""
    import oracle
    from zdr_amd import geometry
    A = cbox_arrays
    tris = A.verts[A.tris][:, :, :3].astype(np.float32)           # identity transforms: object space = world space
    is_light = np.zeros(32, bool); is_light[A.inst_tri_begin[1]:A.inst_tri_begin[2]] = True
    never = never_occluders(tris, is_light)
    assert never.sum() == 8 and not never[is_light].any()
    ylo = tris[..., 1].min(axis=1); yhi = tris[..., 1].max(axis=1)
    ceiling = (ylo > 5.3) & ~is_light
    assert ceiling.sum() == 2 and not never[ceiling].any()
    # the claim itself, by brute force: 200,000 shadow segments from points on EVERY triangle of the scene to points on the light,
    # traced by the oracle against a scene made of the flagged triangles alone: none is occluded
    rng = np.random.default_rng(0)
    n = 200000
    def points(idx):
        u = rng.random((n, 2)).astype(np.float32); f = u.sum(1) > 1; u[f] = 1 - u[f]
        T = tris[idx]
        return T[:, 0] + u[:, :1] * (T[:, 1] - T[:, 0]) + u[:, 1:] * (T[:, 2] - T[:, 0])
    p = points(rng.integers(0, 32, n)); q = points(rng.choice(np.nonzero(is_light)[0], n))
    # a fifth of the origins exactly ON an edge or corner of their triangle (where a wall meets the floor: the closest a ray gets)
    p[: n // 5] = tris[rng.integers(0, 32, n // 5), rng.integers(0, 3, n // 5)]
    d = q - p; dist = np.linalg.norm(d, axis=1).astype(np.float32)
    ok = dist > 1e-3
    rays = np.zeros((n, 8), np.float32); rays[:, :3] = p; rays[:, 3] = 1e-4; rays[:, 4:7] = d / np.maximum(dist, 1e-30)[:, None]; rays[:, 7] = np.float32(0.9999) * dist
    nv = tris[never].reshape(-1, 3)
    verts8 = np.zeros((nv.shape[0], 8), np.float32); verts8[:, :3] = nv; verts8[:, 7] = 1
    S = oracle.OracleScene.from_arrays(geometry.from_arrays(verts8, np.arange(nv.shape[0], dtype=np.int32).reshape(-1, 3)))
    occ = S.trace_any(rays[ok])
    assert occ.sum() == 0, int(occ.sum())


def test_never_occluder_classification_is_conservative():
    """No lights: nothing is ruled out.  A wall with the light ON its far side is an occluder; a light too close to a supporting
    plane (the bound on a rounding-induced hit exceeds half of tmin) keeps that plane in the walk; a plane that cuts the scene
    is never ruled out."""
    floor = quad((0, 0, 0), (4, 0, 0), (4, 0, 4), (0, 0, 4))
    lamp = quad((1.5, 3, 1.5), (2.5, 3, 1.5), (2.5, 3, 2.5), (1.5, 3, 2.5))
    screen = quad((0, 1, 2), (4, 1, 2), (4, 2, 2), (0, 2, 2))    # a vertical panel in the middle of the floor: geometry on both sides
    low_lamp = quad((1.5, 0.001, 1.5), (2.5, 0.001, 1.5), (2.5, 0.001, 2.5), (1.5, 0.001, 2.5))
    t = np.array(floor + lamp + screen, np.float32)
    light = np.array([0, 0, 1, 1, 0, 0], bool)
    assert never_occluders(t, light).tolist() == [True, True, False, False, False, False]
    assert not never_occluders(t, np.zeros(6, bool)).any()
    t2 = np.array(floor + low_lamp, np.float32)
    assert not never_occluders(t2, np.array([0, 0, 1, 1], bool)).any()      # 1 mm above the floor: too close to call
    below = quad((1.5, -3, 1.5), (2.5, -3, 1.5), (2.5, -3, 2.5), (1.5, -3, 2.5))
    t3 = np.array(floor + lamp + below, np.float32)               # lights on BOTH sides of the floor
    assert not never_occluders(t3, np.array([0, 0, 1, 1, 1, 1], bool))[:2].any()
