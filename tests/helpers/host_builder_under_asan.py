"""Run by tests/test_host_sanitizers.py under LD_PRELOAD=libclang_rt.asan: the HOST side of libzdr_hip.so that needs no GPU —
the BVH builder, the quad merge, the plane records (zdr_debug_build_accel) and the argument checks of the entry points."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C
import numpy as np
from zdr_amd import _native, build as hip_build
_native.LIB_PATH = sys.argv[1]
hip_build.stale = lambda: False
from conftest import cbox_models
from gpu_util import multi_light_arrays, terrain_arrays
from zdr_amd import geometry, procedural
import test_bvh_emulation as E
import test_brute_quads as Q
L = _native.lib()
A = geometry.assemble(cbox_models())
big = procedural.tessellated_cbox(cbox_models(), n=24)
for arrays in (A, multi_light_arrays(), terrain_arrays(n=24), big):
    nodes, order, isect = E.build(arrays, _native.ACCEL_BVH)
    print("bvh", arrays.tris.shape[0], len(nodes), int(order.sum()))
    nq, order, isect = Q.build(E.world_triangles(arrays))
    print("brute", arrays.tris.shape[0], nq, Q.NPAR, int(order.sum()))
print("quads", Q.build(A.verts[A.tris][:, :, :3])[0])
Q.test_only_planar_convex_pairs_merge()
# degenerate input: zero-area and NaN triangles must not crash the builders
bad = np.zeros((6, 9), np.float32); bad[1] = np.nan; bad[2, :3] = 1; bad[3] = [0, 0, 0, 1, 0, 0, 0, 1, 0]; bad[4] = [0, 0, 0, 0, 1, 0, 1, 0, 0]; bad[5] = bad[3]
for accel in (_native.ACCEL_BRUTE, _native.ACCEL_BVH):
    nq, se = C.c_uint32(0), C.c_uint32(0)
    order = np.zeros(6, np.int32); isect = np.zeros((6, 12), np.float32); nodes = np.zeros((16, 16), np.float32)
    rc = L.zdr_debug_build_accel(bad.ctypes.data, 6, accel, nodes.ctypes.data, 16, C.byref(nq), C.byref(se), order.ctypes.data, isect.ctypes.data)
    print("degenerate", accel, rc, sorted(order.tolist()))
# the shadow walk's classification (zdr_debug_never_occluders): real scenes, every / no triangle a light, degenerate and NaN triangles
for arrays in (A, multi_light_arrays()):
    tri = E.world_triangles(arrays)
    lights = np.zeros(tri.shape[0], bool)
    for i in range(1, arrays.ninst): lights[arrays.inst_tri_begin[i]:arrays.inst_tri_begin[i + 1]] = True
    for flags in (lights, np.ones_like(lights), np.zeros_like(lights)):
        print("never", tri.shape[0], int(flags.sum()), int(Q.never_occluders(tri, flags).sum()))
print("never degenerate", Q.never_occluders(bad.reshape(6, 3, 3), np.array([0, 0, 0, 1, 0, 0], bool)).tolist())
print("never null", L.zdr_debug_never_occluders(None, 0, None, None))
# argument checks that return before any HIP call
h = C.c_void_p()
print("create(null)", L.zdr_scene_create(None, 0, None, 0, None, None, None, 0, 0, 0, C.byref(h)), L.zdr_last_error().decode()[:40])
v = np.zeros((3, 8), np.float32); t = np.array([[0, 1, 2]], np.int32); e = np.zeros((1, 3), np.float32)
for begin, tri, what in (([1, 1], t, "begin"), ([0, 2], t, "span"), ([0, 1], np.array([[0, 1, 7]], np.int32), "index"), ([0, 1], np.array([[0, -1, 2]], np.int32), "negative")):
    b = np.array(begin, np.int32)
    rc = L.zdr_scene_create(v.ctypes.data, 3, tri.ctypes.data, 1, b.ctypes.data, None, e.ctypes.data, 1, 0, 0, C.byref(h))
    print("create", what, rc, L.zdr_last_error().decode()[:50]); assert rc != 0
print("destroy(null)", L.zdr_scene_destroy(None), "check(null)", L.zdr_scene_check(None, None))
print("ok")
