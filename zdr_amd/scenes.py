"""The workloads BASELINE.json names, as Scene objects: Cornell-box geometry, camera and the two materials of
SURVEY §8d (fd_validate.py:21-33, example.py:13-23), plus the 1 M-triangle scene of configs[4].

Used by bench.py, __graft_entry__.smoke(), tools/ and the tests.  Nothing here imports the CPU oracle: the benchmark's
timed region and everything it loads before the ``cpu_baseline`` leg is the product alone.
"""
from __future__ import annotations

import os

import numpy as np

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
