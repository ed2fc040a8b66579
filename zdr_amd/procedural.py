"""Procedural scenes.

``tessellated_cbox`` is BASELINE.json configs[4] ("1M-tri synthetic OBJ (Sponza-scale)"):
every triangle of the textured cbox mesh is subdivided into n^2 triangles on a barycentric grid and
the new vertices are displaced along the interpolated normal by a smooth, seeded function of the
world position (identical on shared edges, so the surface stays closed).  UVs are interpolated, so
the same 1024^2 material textures apply; the light instance is kept as it is.
"""
from __future__ import annotations

import numpy as np

from .geometry import SceneArrays, assemble


def _displacement(p: np.ndarray, amplitude: float, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    k = rng.uniform(3.0, 9.0, (4, 3))
    ph = rng.uniform(0.0, 2 * np.pi, 4)
    d = np.zeros(p.shape[0], np.float64)
    for i in range(4):
        d += np.sin(p.astype(np.float64) @ k[i] + ph[i])
    return (amplitude * 0.25 * d).astype(np.float32)


def tessellate(verts: np.ndarray, tris: np.ndarray, n: int, amplitude: float, seed: int):
    """n^2 sub-triangles per input triangle; returns (verts8, tris) with per-triangle private vertices."""
    i, j = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    keep = (i + j) <= n
    gi, gj = i[keep], j[keep]                                # grid points of one triangle
    idx = -np.ones((n + 1, n + 1), np.int64)
    idx[gi, gj] = np.arange(gi.size)
    b1 = (gi / n).astype(np.float32); b2 = (gj / n).astype(np.float32); b0 = (1 - b1 - b2).astype(np.float32)
    up = [(idx[a, b], idx[a + 1, b], idx[a, b + 1]) for a in range(n) for b in range(n - a)]
    dn = [(idx[a + 1, b], idx[a + 1, b + 1], idx[a, b + 1]) for a in range(n) for b in range(n - a - 1)]
    local = np.asarray(up + dn, np.int64)                    # n^2 triangles
    V = verts[tris]                                          # (T, 3, 8)
    P = b0[None, :, None] * V[:, None, 0, :] + b1[None, :, None] * V[:, None, 1, :] + b2[None, :, None] * V[:, None, 2, :]   # (T, G, 8)
    P = P.reshape(-1, 8).astype(np.float32)
    nrm = P[:, 5:8] / np.linalg.norm(P[:, 5:8], axis=1, keepdims=True)
    P[:, 0:3] += nrm * _displacement(P[:, 0:3], amplitude, seed)[:, None]
    P[:, 5:8] = nrm
    T = tris.shape[0]
    out_tris = (local[None, :, :] + (np.arange(T) * gi.size)[:, None, None]).reshape(-1, 3)
    return P, out_tris.astype(np.int32)


def tessellated_cbox(models, n: int = 183, amplitude: float = 0.01, seed: int = 0) -> SceneArrays:
    """models: the usual [(obj, transform, emission), ...]; instance 0 is tessellated (30 n^2 triangles
    for cboxuv.obj; n = 183 gives 1,004,670), the other instances are kept."""
    base = assemble(models)
    b = base.inst_tri_begin
    v0, t0 = tessellate(base.verts, base.tris[b[0]:b[1]], n, amplitude, seed)
    rest = base.tris[b[1]:]
    used = np.unique(rest)
    remap = -np.ones(base.verts.shape[0], np.int64); remap[used] = np.arange(used.size) + v0.shape[0]
    verts = np.concatenate([v0, base.verts[used]]).astype(np.float32)
    tris = np.concatenate([t0, remap[rest].astype(np.int32)]).astype(np.int32)
    begin = np.concatenate([[0], b[1:] - b[1] + t0.shape[0]]).astype(np.int32)
    return SceneArrays(np.ascontiguousarray(verts), np.ascontiguousarray(tris), begin, base.inst_xform.copy(), base.inst_emission.copy())
