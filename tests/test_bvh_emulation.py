"""CPU emulation of the DEVICE traversal (csrc/accel.h, BvhAccel::traverse) on the node / triangle
arrays the host builder produces (zdr_debug_build_accel, no GPU involved), checked against the
oracle's brute force.  This is the executable spec of the traversal: change accel.h and this file
together.  It exists because a traversal bug on the GPU is a hang, not a failed assert."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import cbox_models
from zdr_amd import _native, geometry



def world_triangles(A):
    out = np.zeros((A.tris.shape[0], 3, 3), np.float32)
    for i in range(A.ninst):
        M = A.inst_xform[i].reshape(4, 4)
        for t in range(A.inst_tri_begin[i], A.inst_tri_begin[i + 1]):
            for k in range(3):
                v = A.verts[A.tris[t, k], :3]
                out[t, k] = [M[r, 0] * v[0] + M[r, 1] * v[1] + M[r, 2] * v[2] + M[r, 3] for r in range(3)]
    return out


def build(A, accel):
    tri = np.ascontiguousarray(world_triangles(A).reshape(-1, 9))
    n = tri.shape[0]
    nodes = np.zeros((max(n, 8), 16), np.float32)      # BVH4 node = 64 bytes (csrc/scene.h)
    order = np.zeros(n, np.int32); isect = np.zeros((n, 12), np.float32); nn = C.c_uint32(); se = C.c_uint32()
    rc = _native.lib().zdr_debug_build_accel(tri.ctypes.data, n, accel, nodes.ctypes.data, nodes.shape[0], C.byref(nn), C.byref(se), order.ctypes.data, isect.ctypes.data)
    assert rc == 0, _native.lib().zdr_last_error()
    global STACK
    STACK = se.value                          # what the kernels allocate in LDS for this tree
    assert 8 <= STACK <= 64 and STACK % 4 == 0
    return nodes[:nn.value], order, isect


def tri_test(q, o, d, tmin, tmax):
    f = np.float32
    nd = f(q[0] * d[0] + q[1] * d[1] + q[2] * d[2])
    tn = f(q[3] - f(q[0] * o[0] + q[1] * o[1] + q[2] * o[2]))
    with np.errstate(divide="ignore", invalid="ignore"):
        t = f(tn / nd)
    p = o + d * t
    u = f(q[4] * p[0] + q[5] * p[1] + q[6] * p[2] + q[7]); v = f(q[8] * p[0] + q[9] * p[1] + q[10] * p[2] + q[11])
    return bool(t > tmin and t < tmax and u >= 0 and v >= 0 and u + v <= 1), t


def qbox_entry(k, q, A, B, tmin, tmax, inv):
    """mirror of qbox_entry<K>: q = (lxq, lyq, lzq, hxq, hyq, hzq) words, byte k belongs to child k; t = q A + B in float32.
    Per axis the ray enters through the low plane when its direction is positive, else through the high plane."""
    f = np.float32
    ql = np.array([(int(q[a]) >> (8 * k)) & 255 for a in range(3)], np.float32)
    qh = np.array([(int(q[3 + a]) >> (8 * k)) & 255 for a in range(3)], np.float32)
    neg = inv < 0
    qn, qf = np.where(neg, qh, ql), np.where(neg, ql, qh)
    with np.errstate(invalid="ignore", over="ignore"):
        # fmaf: one rounding; float64 product + sum rounded once is exact enough to mirror it (24 x 8 bit product is exact)
        tnear = (qn.astype(np.float64) * A.astype(np.float64) + B.astype(np.float64)).astype(f)
        tfar = (qf.astype(np.float64) * A.astype(np.float64) + B.astype(np.float64)).astype(f)
    tn = np.fmax(np.fmax.reduce(tnear), tmin)            # fmin/fmax drop NaNs like v_min_f32 / v_max_f32
    tf = np.fmin(np.fmin.reduce(tfar), tmax)
    return tn if tn <= tf else 3.0e38


def decode_node(n):
    """{origin.xyz, scale.x} {scale.y, scale.z, qlo.x, qlo.y} {qlo.z, qhi.x, qhi.y, qhi.z} {child words}"""
    w = n.view(np.uint32)
    origin = n[0:3].copy(); scale = np.array([n[3], n[4], n[5]], np.float32)
    q = [w[6], w[7], w[8], w[9], w[10], w[11]]
    return origin, scale, q, [int(x) for x in w[12:16]]


def traverse(nodes, isect, o, d, tmin, tmax, any_hit):
    """line-by-line mirror of BvhAccel::traverse<ANY>"""
    ntris, nnodes = isect.shape[0], nodes.shape[0]
    best_t, slot = np.float32(tmax), -1
    with np.errstate(divide="ignore"):
        inv = np.float32(1.0) / d
    stack = []
    # a child word is the byte offset of what it names — nodes (64 bytes each) first, the plane records (48 bytes per slot) behind them — | its count
    isect_off = 64 * nnodes
    def where(word):
        off, c = word & ~15, word & 7
        return (off // 64 if c == 0 else (off - isect_off) // 48), c
    nid, cnt = 0, (ntris if nnodes == 0 else 0)
    budget = 2 * (nnodes + ntris) + 8
    steps = 0
    while True:
        budget -= 1
        if budget < 0:
            raise AssertionError("watchdog fired: the walk visited something twice")
        steps += 1
        descended = False
        if cnt == 0:
            origin, scale, q, p = decode_node(nodes[nid])
            with np.errstate(invalid="ignore", over="ignore"):
                A = (scale * inv).astype(np.float32); B = ((origin - o) * inv).astype(np.float32)
            e = [qbox_entry(c, q, A, B, tmin, best_t, inv) for c in range(4)]      # unused slots carry an inverted box: no test for them
            if np.isfinite(o).all() and np.isfinite(d).all():              # (a ray of NaNs / infinities decides no comparison: it may "enter" one, and its walk ends there)
                assert all(e[c] > 2.0e38 for c in range(4) if (p[c] & 7) == 7), (o, d)
            em = min(e)
            if em < 2.0e38:
                nxt, taken = -1, [False] * 4
                for c in range(4):
                    if nxt < 0 and e[c] == em:
                        nxt, taken[c] = p[c], True
                assert len(stack) + 4 <= STACK        # the device stores four slots unconditionally
                for c in range(4):
                    if not taken[c] and e[c] < 2.0e38:
                        stack.append(p[c])
                if nxt >= 0:
                    nid, cnt = where(nxt)
                    descended = True
        else:
            for s in range(nid, nid + cnt):
                ok, t = tri_test(isect[s], o, d, tmin, best_t)
                if ok:
                    best_t, slot = t, s
            if any_hit and slot >= 0:
                return slot, best_t, steps
        if descended:
            continue
        if not stack:
            break
        e = stack.pop()
        nid, cnt = where(e)
        if cnt == 7:                                   # only a ray of NaNs gets into an unused slot: its walk ends, it hits nothing
            break
    return slot, best_t, steps


def rays_for(lo, hi, n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def check_scene(A, lo, hi, nrays, seed):
    nodes, order, isect = build(A, _native.ACCEL_BVH)
    # structure: leaves tile the slots, order is a permutation
    assert sorted(order.tolist()) == list(range(A.tris.shape[0]))
    S = oracle.OracleScene.from_arrays(A)
    o, d = rays_for(lo, hi, nrays, seed)
    rays = np.zeros((nrays, 8), np.float32); rays[:, :3] = o; rays[:, 4:7] = d; rays[:, 7] = 1e30
    rip, rbt = S.trace_closest(rays)
    tri_inst = np.repeat(np.arange(A.ninst), np.diff(A.inst_tri_begin))
    worst = 0
    for i in range(nrays):
        slot, t, steps = traverse(nodes, isect, o[i], d[i], np.float32(0), np.float32(1e30), False)
        worst = max(worst, steps)
        if rip[i, 0] < 0:
            assert slot < 0, i
        else:
            assert slot >= 0, i
            tri = order[slot]
            same = tri_inst[tri] == rip[i, 0] and tri - A.inst_tri_begin[tri_inst[tri]] == rip[i, 1]
            assert same or abs(t - rbt[i, 2]) < 1e-5 * (1 + abs(rbt[i, 2])), (i, tri, rip[i])   # ties on shared edges
            assert abs(t - rbt[i, 2]) <= 1e-5 * abs(rbt[i, 2]) + 5e-6
        # any-hit with a bounded ray agrees with the closest hit
        tm = np.float32(0.7 * rbt[i, 2]) if rip[i, 0] >= 0 else np.float32(5.0)
        s_any, _, _ = traverse(nodes, isect, o[i], d[i], np.float32(1e-4), tm, True)
        occ = S.trace_any(np.concatenate([o[i], [1e-4], d[i], [tm]]).astype(np.float32)[None])[0]
        assert (s_any >= 0) == bool(occ), i
    return nodes.shape[0], worst


def test_cbox_forced_bvh():
    A = geometry.assemble(cbox_models())
    nn, worst = check_scene(A, (-3, 0, -5.5), (2.5, 5.2, 6), 300, 1)
    assert 1 <= nn <= 32 and worst <= 2 * (nn + 32)


def test_terrain_bvh():
    from gpu_util import terrain_arrays
    A = terrain_arrays(n=14)                     # 394 triangles
    nn, worst = check_scene(A, (-3, -0.5, -3), (3, 3.5, 3), 250, 2)
    assert nn > 20


def test_nan_and_axis_aligned_rays_terminate():
    from gpu_util import terrain_arrays
    A = terrain_arrays(n=10)
    nodes, order, isect = build(A, _native.ACCEL_BVH)
    for o, d in [((np.nan, 0, 0), (0, -1, 0)), ((0, 2, 0), (0, -1, 0)), ((0, 2, 0), (np.nan, np.nan, np.nan)),
                 ((0.3, 2, 0.1), (1, 0, 0)), ((3, 0.0, 3), (-1, 0, 0)), ((np.inf, 0, 0), (1, 0, 0))]:
        slot, t, steps = traverse(nodes, isect, np.array(o, np.float32), np.array(d, np.float32), np.float32(0), np.float32(1e30), False)
        assert steps <= 2 * (nodes.shape[0] + isect.shape[0]) + 8


def test_tiny_scene_is_a_single_leaf():
    v = np.array([[0, 0, 0, 0, 0, 0, 1, 0], [1, 0, 0, 1, 0, 0, 1, 0], [0, 0, 1, 0, 1, 0, 1, 0]], np.float32)
    A = geometry.from_arrays(v, np.array([[0, 2, 1]], np.int32))
    nodes, order, isect = build(A, _native.ACCEL_BVH)
    assert nodes.shape[0] == 0
    slot, t, _ = traverse(nodes, isect, np.array([0.2, 1, 0.2], np.float32), np.array([0, -1, 0], np.float32), np.float32(0), np.float32(1e30), False)
    assert slot == 0 and abs(t - 1) < 1e-6


def test_the_top_of_the_tree_is_numbered_breadth_first():
    """zdr_api.cpp numbers the first ZDR_BVH_BFS_NODES (341) nodes breadth-first — root, its children, their children … — and
    everything below depth-first: 'node id < K' is the top of the tree for any K up to there (what a kernel would keep in
    LDS, accel.h), and a parent always precedes its children."""
    from gpu_util import terrain_arrays
    A = terrain_arrays(n=64)                                     # 8,194 triangles: ~1,700 nodes
    nodes, order, isect = build(A, 2)
    assert nodes.shape[0] > 341
    level, frontier, seen = 0, [0], 0
    expect = 0
    while frontier and expect < 341:
        nxt = []
        for nid in frontier:
            if expect < 341:
                assert nid == expect, (level, nid, expect)       # breadth-first: ids come level by level, left to right
            expect += 1
            child = nodes[nid].view(np.uint32)[12:16]
            for cw in child:
                if (cw & 7) == 0:                                # a node: the word is its byte offset, 64 bytes per node
                    assert (cw >> 6) > nid
                    nxt.append(int(cw >> 6))
        frontier, level = nxt, level + 1
    assert level >= 4
