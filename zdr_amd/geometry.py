"""Scene assembly on the host: models -> flat arrays handed to the C-ABI.

Follows Scene.load_geometry (/root/reference/render.py:73-128): one mesh instance per
``(obj_file, transform, emission)`` model, geometry cached per file name, an instance is a
light iff any emission component is > 0, normals recomputed when the OBJ carries none
(recompute_normal.py:5-51).  Arrays stay in OBJECT space; the 4x4 transforms travel beside
them (row-major) and the native side flattens them at BVH build time.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .load_obj import concat_triangles, read_obj
from .mathtypes import as_row_major_4x4, float3

MAX_INSTANCES = 10000  # render.py:114-115


@dataclass
class SceneArrays:
    verts: np.ndarray           # (nverts, 8) float32: v[3], vt[2], vn[3]   (vertex.py:4)
    tris: np.ndarray            # (ntris, 3) int32, indices into verts
    inst_tri_begin: np.ndarray  # (ninst + 1,) int32
    inst_xform: np.ndarray      # (ninst, 16) float32 row-major
    inst_emission: np.ndarray   # (ninst, 3) float32

    @property
    def ninst(self) -> int:
        return int(self.inst_emission.shape[0])


def normalize_emission(e) -> np.ndarray:
    """None -> 0, number -> grey, float3/sequence -> rgb (render.py:85-88,142)."""
    if e is None:
        return np.zeros(3, np.float32)
    if isinstance(e, (int, float)):
        return np.full(3, float(e), np.float32)
    return np.asarray(tuple(float3(e)), np.float32)


def recompute_normal(verts: np.ndarray, tris: np.ndarray) -> None:
    """Area-weighted vertex normals: sum of un-normalised face normals cross(e1, e2) at the
    three corners, then normalise (recompute_normal.py:5-39). In place; float32 sums."""
    p = verts[:, 0:3]
    e1 = p[tris[:, 1]] - p[tris[:, 0]]
    e2 = p[tris[:, 2]] - p[tris[:, 0]]
    fn = np.cross(e1, e2).astype(np.float32)
    acc = np.zeros((verts.shape[0], 3), np.float32)
    for k in range(3):
        np.add.at(acc, tris[:, k], fn)
    ln = np.sqrt((acc * acc).sum(axis=1, keepdims=True, dtype=np.float32))
    verts[:, 5:8] = acc / ln


def load_mesh(obj_file: str):
    vertices, faces = read_obj(obj_file)
    verts = np.array([(*v, *t, *n) for v, t, n in vertices], dtype=np.float32).reshape(-1, 8)
    tris = np.asarray(concat_triangles(faces), dtype=np.int32).reshape(-1, 3)
    if len(vertices) and np.isnan(verts[:, 5:8]).any():  # App. B-14: test all vertices, not only [0]
        recompute_normal(verts, tris)
    return verts, tris


def assemble(models) -> SceneArrays:
    if len(models) == 0:
        raise ValueError("a scene needs at least one model")
    if len(models) > MAX_INSTANCES:
        raise RuntimeError("exceeding maximum number of mesh instances")
    cache = {}
    vparts, tparts, begin, xf, em = [], [], [0], [], []
    vbase = 0
    for obj_file, transform, emission in models:
        if obj_file not in cache:
            cache[obj_file] = load_mesh(obj_file)
        v, t = cache[obj_file]
        # an instance references its own copy of the vertices so that triangle indices stay
        # global; shared meshes cost memory only in this cold path
        vparts.append(v)
        tparts.append(t + vbase)
        vbase += v.shape[0]
        begin.append(begin[-1] + t.shape[0])
        xf.append(as_row_major_4x4(transform).reshape(16))
        em.append(normalize_emission(emission))
    return SceneArrays(
        verts=np.ascontiguousarray(np.concatenate(vparts, axis=0), np.float32),
        tris=np.ascontiguousarray(np.concatenate(tparts, axis=0), np.int32),
        inst_tri_begin=np.asarray(begin, np.int32),
        inst_xform=np.ascontiguousarray(np.stack(xf), np.float32),
        inst_emission=np.ascontiguousarray(np.stack(em), np.float32),
    )


def from_arrays(verts, tris, inst_tri_begin=None, inst_xform=None, inst_emission=None) -> SceneArrays:
    """Build SceneArrays straight from arrays (procedural scenes, tests)."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 8)
    tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
    if inst_tri_begin is None:
        inst_tri_begin = [0, tris.shape[0]]
    inst_tri_begin = np.asarray(inst_tri_begin, np.int32)
    ninst = inst_tri_begin.shape[0] - 1
    if inst_xform is None:
        inst_xform = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (ninst, 1))
    if inst_emission is None:
        inst_emission = np.zeros((ninst, 3), np.float32)
    return SceneArrays(verts, tris, inst_tri_begin,
                       np.ascontiguousarray(inst_xform, np.float32).reshape(ninst, 16),
                       np.ascontiguousarray(inst_emission, np.float32).reshape(ninst, 3))
