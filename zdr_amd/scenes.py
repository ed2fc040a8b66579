"""The workloads BASELINE.json names, as Scene objects: Cornell-box geometry, camera and the two materials of
SURVEY §8d (fd_validate.py:21-33, example.py:13-23), plus the 1 M-triangle scene of configs[4].

Used by bench.py, __graft_entry__.smoke(), tools/ and the tests.  Nothing here imports the CPU oracle: the benchmark's
timed region and everything it loads before the ``cpu_baseline`` leg is the product alone.
"""
from __future__ import annotations

import os

import numpy as np

from . import geometry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Data fixtures taken from the reference's assets/ (OBJ + PNG, no code); override with ZDR_ASSETS.
ASSETS = os.environ.get("ZDR_ASSETS", os.path.join(ROOT, "tests", "golden", "assets"))

CBOX_CAMERA = (50 / 180 * 3.1415926, (-0.2, 2.6, 6.0), (-0.2, 2.6, -2.5), (0.0, 1.0, 0.0))  # fd_validate.py:28-33

# name: (integrator, resolution, spp, scene, BASELINE.json configs index)
CONFIGS = {
    "c1": ("collocated", 256, 1, "cbox", 0),
    "c2": ("direct", 512, 64, "cbox", 1),
    "c3": ("path", 512, 256, "cbox", 2),
    "c4": ("path", 1024, 1024, "cbox", 3),
    "c5": ("path", 1024, 256, "tess1m", 4),
}
TESS1M_N = 183                 # 30 n^2 + 2 = 1,004,672 triangles


def cbox_models(emission=20.0):
    """fd_validate.py:21-24: instance 0 = cboxuv.obj (textured), instance 1 = cbox-light.obj (emitter)."""
    return [(os.path.join(ASSETS, "cboxuv.obj"), None, 0.0), (os.path.join(ASSETS, "cbox-light.obj"), None, emission)]


def cbox_material_np() -> np.ndarray:
    """Material A of SURVEY §8d: ((cboxd RGB, cboxr R) / 255) ** 2.2, (1024, 1024, 4) float32 (example.py:13-18)."""
    from PIL import Image
    d = np.asarray(Image.open(os.path.join(ASSETS, "cboxd.png")))[..., :3]
    r = np.asarray(Image.open(os.path.join(ASSETS, "cboxr.png")))[..., :1]
    return np.ascontiguousarray((np.concatenate([d, r], -1).astype(np.float32) / np.float32(255.0)) ** np.float32(2.2))


def fd_material_np(res=1024, seed=0) -> np.ndarray:
    """Material B of SURVEY §8d: diffuse U(0.2, 0.8), roughness U(0.3, 0.9); interior values so FD is legal
    (fd_validate.py:93-94)."""
    rng = np.random.default_rng(seed)
    m = np.empty((res, res, 4), np.float32)
    m[..., :3] = rng.uniform(0.2, 0.8, (res, res, 3))
    m[..., 3] = rng.uniform(0.3, 0.9, (res, res))
    return m


def cbox_camera():
    from .mathtypes import Camera, float3
    return Camera(fov=CBOX_CAMERA[0], origin=float3(*CBOX_CAMERA[1]), target=float3(*CBOX_CAMERA[2]), up=float3(*CBOX_CAMERA[3]))


def make_scene(integrator, accel="auto", models=None, arrays=None, **kw):
    """A Scene of the Cornell box (or of ``models`` / prebuilt ``arrays``) seen through the fd_validate camera."""
    from .render import Scene
    s = Scene(arrays if arrays is not None else (models or cbox_models()), integrator=integrator, accel=accel, **kw)
    s.camera = cbox_camera()
    return s


def tess1m_arrays(n=TESS1M_N):
    """BASELINE configs[4]: instance 0 of the Cornell box tessellated and displaced (seed 0) to 30 n^2 triangles."""
    from . import procedural
    return procedural.tessellated_cbox(cbox_models(), n=n)


def config_scene(name):
    """(scene, resolution, spp) of one BASELINE config."""
    integrator, W, spp, kind, _ = CONFIGS[name]
    scene = make_scene(integrator, arrays=tess1m_arrays()) if kind == "tess1m" else make_scene(integrator)
    return scene, W, spp


# ---------------------------------------------------------------- procedural test / measurement scenes
def terrain_arrays(n=64, seed=0, light=True):
    """Procedural BVH-stress scene: an n x n displaced height-field (2 n^2 triangles, smooth normals,
    UV atlas = the unit square) lit by a quad light above it; instance 0 textured, instance 1 light."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-3.0, 3.0, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(g, g, indexing="xy")
    Y = (0.35 * np.sin(1.7 * X) * np.cos(1.3 * Z) + 0.05 * rng.standard_normal(X.shape)).astype(np.float32)
    P = np.stack([X, Y, Z], -1).reshape(-1, 3)
    U = np.stack([(X + 3) / 6, (Z + 3) / 6], -1).reshape(-1, 2).astype(np.float32)
    idx = lambda i, j: i * (n + 1) + j
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = idx(i, j), idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)
            tris += [(a, c, b), (a, d, c)]          # wound so that normals point +y
    tris = np.asarray(tris, np.int32)
    verts = np.zeros((P.shape[0], 8), np.float32)
    verts[:, 0:3] = P; verts[:, 3:5] = U; verts[:, 5:8] = np.nan
    geometry.recompute_normal(verts, tris)
    lv = np.array([[-1, 4, -1, 0, 0, 0, -1, 0], [1, 4, -1, 0, 0, 0, -1, 0], [1, 4, 1, 0, 0, 0, -1, 0], [-1, 4, 1, 0, 0, 0, -1, 0]], np.float32)
    lt = np.array([[0, 1, 2], [0, 2, 3]], np.int32) + verts.shape[0]
    V = np.concatenate([verts, lv]); T = np.concatenate([tris, lt])
    em = np.array([[0, 0, 0], [30, 30, 30]], np.float32) if light else np.zeros((2, 3), np.float32)
    return geometry.from_arrays(V, T, [0, tris.shape[0], T.shape[0]], None, em)


TERRAIN_CAMERA_TUPLE = (0.9, (0.5, 3.0, 6.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))


def terrain_camera():
    from .mathtypes import Camera, float3
    f, o, tg, up = TERRAIN_CAMERA_TUPLE
    return Camera(fov=f, origin=float3(*o), target=float3(*tg), up=float3(*up))


def random_rays(n, lo, hi, seed=0, tmax=1e30):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = o; r[:, 3] = 0.0; r[:, 4:7] = d; r[:, 7] = tmax
    return r


def panel_mesh(nx, ny):
    """A 2 x 2 panel in the xz plane facing +y (like quad.obj), cut into nx x ny cells of two triangles each:
    2 nx ny triangles, per-vertex normal (0, 1, 0), uv = the unit square."""
    gx = np.linspace(-1.0, 1.0, nx + 1, dtype=np.float32); gz = np.linspace(-1.0, 1.0, ny + 1, dtype=np.float32)
    verts = np.zeros(((nx + 1) * (ny + 1), 8), np.float32)
    k = 0
    for j in range(ny + 1):
        for i in range(nx + 1):
            verts[k] = (gx[i], 0.0, gz[j], (gx[i] + 1) / 2, (gz[j] + 1) / 2, 0.0, 1.0, 0.0); k += 1
    tris = []
    for j in range(ny):
        for i in range(nx):
            a, b, c, d = j * (nx + 1) + i, j * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i
            tris += [(a, c, b), (a, d, c)]          # wound so that cross(p1 - p0, p2 - p0) points +y
    return verts, np.asarray(tris, np.int32)


def multi_light_arrays(emissions=((0, 0, 0), (20, 20, 20), (6, 2, 1), (0, 0, 0), (1, 3, 8))):
    """Light-stage style scene (test_lightstage.py:24-62): the Cornell box (instance 0, textured) with FOUR more
    instances of DIFFERENT triangle counts — the ceiling light (2 triangles), a warm panel on the left wall (8),
    a blocker slab in mid-air (2, never emits: the light list must skip it) and a cool panel on the back wall (6),
    each with its own transform.  `emissions` has one rgb per instance."""
    base = geometry.assemble(cbox_models())
    V = [base.verts]; T = [base.tris]; begin = list(base.inst_tri_begin); X = [base.inst_xform[0], base.inst_xform[1]]
    def add(verts, tris, xform):
        nv = sum(v.shape[0] for v in V)
        V.append(verts); T.append(tris + nv); begin.append(begin[-1] + tris.shape[0]); X.append(np.asarray(xform, np.float32).reshape(16))
    # left wall x = -3.0: panel normal +y -> +x (rotate about z by -90 degrees), scaled 0.5 x 1.0
    add(*panel_mesh(2, 2), [[0, 1, 0, -2.9], [-1.0, 0, 0, 2.5], [0, 0, 0.5, -3.0], [0, 0, 0, 1]])
    # blocker: a slab facing down, hanging under the ceiling light
    add(*panel_mesh(1, 1), [[0.6, 0, 0, -0.2], [0, -1, 0, 4.2], [0, 0, -0.6, -3.0], [0, 0, 0, 1]])
    # back wall z = -5.8: panel normal +y -> +z, 3 x 1 cells
    add(*panel_mesh(3, 1), [[0.9, 0, 0, 0.4], [0, 0, -0.4, 1.6], [0, 1, 0, -5.7], [0, 0, 0, 1]])
    return geometry.from_arrays(np.concatenate(V), np.concatenate(T), begin, np.stack(X), np.asarray(emissions, np.float32))
