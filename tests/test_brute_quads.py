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
