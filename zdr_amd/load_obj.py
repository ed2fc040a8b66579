"""Wavefront OBJ ingest (host side, cold path).

Same contract as /root/reference/load_obj.py:1-68: ``read_obj`` returns the de-duplicated
``(position, texcoord, normal)`` vertex tuples in first-use order plus faces re-indexed into
that list; a missing ``vt`` becomes ``(0, 0)``, a missing ``vn`` becomes NaNs (which triggers
normal recomputation, render.py:101-103); ``concat_triangles`` fan-triangulates.
"""
from __future__ import annotations

_NAN3 = (float("nan"),) * 3


def _corner(token: str):
    """'p', 'p/t', 'p//n' or 'p/t/n' -> zero-based (p, t|None, n|None)."""
    f = token.split("/")
    p = int(f[0]) - 1
    t = int(f[1]) - 1 if len(f) > 1 and f[1] else None
    n = int(f[2]) - 1 if len(f) > 2 and f[2] else None
    return p, t, n


def read_obj(file_path):
    pos, tex, nrm = [], [], []
    vertices, index_of, faces = [], {}, []
    with open(file_path, "r") as fh:
        for raw in fh:
            tok = raw.split()
            if not tok:
                continue
            kind = tok[0]
            if kind == "v":
                pos.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif kind == "vt":
                tex.append((float(tok[1]), float(tok[2])))
            elif kind == "vn":
                nrm.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif kind == "f":
                face = []
                for c in tok[1:]:
                    p, t, n = _corner(c)
                    key = (pos[p], tex[t] if t is not None else (0.0, 0.0), nrm[n] if n is not None else _NAN3)
                    # A corner without a normal is never merged: the reference builds a fresh
                    # NaN tuple per corner (load_obj.py:49) and NaN != NaN, so such corners
                    # always miss its dict.  Normal recomputation then yields flat face normals.
                    idx = index_of.get(key) if n is not None else None
                    if idx is None:
                        idx = len(vertices)
                        if n is not None:
                            index_of[key] = idx
                        vertices.append(key)
                    face.append(idx)
                faces.append(face)
    return vertices, faces


def concat_triangles(faces):
    tris = []
    for f in faces:
        for i in range(2, len(f)):
            tris.extend((f[0], f[i - 1], f[i]))
    return tris
