"""Small value types of the public API.

The reference exports LuisaCompute's ``float3`` and a ``Camera`` struct
(/root/reference/render.py:10,28; ``from zdr import Scene, Camera, float3``,
__init__.py:1).  These are host-side stand-ins with the same constructor
forms and field names; they carry no device state.
"""
from __future__ import annotations

import numpy as np


class float3:
    """``float3(v)`` broadcasts, ``float3(x, y, z)`` sets components (luisa.float3)."""

    __slots__ = ("x", "y", "z")

    def __init__(self, *args):
        if len(args) == 1:
            a = args[0]
            if isinstance(a, float3):
                self.x, self.y, self.z = a.x, a.y, a.z
            elif isinstance(a, (int, float)):
                self.x = self.y = self.z = float(a)
            else:
                self.x, self.y, self.z = (float(c) for c in a)
        elif len(args) == 3:
            self.x, self.y, self.z = (float(c) for c in args)
        elif len(args) == 0:
            self.x = self.y = self.z = 0.0
        else:
            raise TypeError("float3 takes 0, 1 or 3 arguments")

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __len__(self):
        return 3

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __eq__(self, o):
        return isinstance(o, float3) and tuple(self) == tuple(o)

    def __repr__(self):
        return f"float3({self.x}, {self.y}, {self.z})"


class float4x4:
    """4x4 transform.  ``float4x4(s)`` is ``s * identity``; ``float4x4(*sixteen)`` takes the
    entries in COLUMN-major order like ``luisa.float4x4`` (test_lightstage.py:44 passes
    ``m.transpose().flatten()``).  ``float4x4.from_rows(m)`` takes a row-major 4x4 array."""

    __slots__ = ("m",)

    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], (int, float)):
            self.m = np.eye(4, dtype=np.float64) * float(args[0])
        elif len(args) == 16:
            self.m = np.asarray(args, dtype=np.float64).reshape(4, 4).T.copy()
        elif len(args) == 1 and isinstance(args[0], float4x4):
            self.m = args[0].m.copy()
        else:
            raise TypeError("float4x4 takes a scalar or 16 column-major entries")

    @staticmethod
    def from_rows(rows) -> "float4x4":
        r = float4x4(1.0)
        r.m = np.asarray(rows, dtype=np.float64).reshape(4, 4).copy()
        return r

    def rows(self) -> np.ndarray:
        return self.m


def as_row_major_4x4(t) -> np.ndarray:
    """None -> identity; float4x4 -> its matrix; array-like 4x4 -> taken as row-major."""
    if t is None:
        return np.eye(4, dtype=np.float32)
    if isinstance(t, float4x4):
        return t.m.astype(np.float32)
    a = np.asarray(t, dtype=np.float64)
    if a.shape != (4, 4):
        raise ValueError("transform must be None, a float4x4 or a 4x4 array")
    return a.astype(np.float32)


class Camera:
    """Pinhole camera: ``fov`` is the full horizontal angle in radians (render.py:28,
    camera.py:5-17).  Field order/names as in the reference struct."""

    __slots__ = ("fov", "origin", "target", "up")

    def __init__(self, fov, origin, target, up):
        self.fov = float(fov)
        self.origin = float3(origin)
        self.target = float3(target)
        self.up = float3(up)

    def copy(self) -> "Camera":
        return Camera(self.fov, self.origin, self.target, self.up)

    def __repr__(self):
        return f"Camera(fov={self.fov}, origin={self.origin}, target={self.target}, up={self.up})"
