import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ASSETS = os.path.join(ROOT, "tests", "golden", "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")

CBOX_CAMERA = (50 / 180 * 3.1415926, (-0.2, 2.6, 6.0), (-0.2, 2.6, -2.5), (0.0, 1.0, 0.0))  # fd_validate.py:28-33


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def cbox_models(emission=20.0):
    return [(os.path.join(ASSETS, "cboxuv.obj"), None, 0.0), (os.path.join(ASSETS, "cbox-light.obj"), None, emission)]


def cbox_material_np():
    """Material A of SURVEY §8d: ((cboxd RGB, cboxr R)/255) ** 2.2 (example.py:13-18)."""
    from PIL import Image
    d = np.asarray(Image.open(os.path.join(ASSETS, "cboxd.png")))[..., :3]
    r = np.asarray(Image.open(os.path.join(ASSETS, "cboxr.png")))[..., :1]
    return np.ascontiguousarray((np.concatenate([d, r], -1).astype(np.float32) / np.float32(255.0)) ** np.float32(2.2))


def fd_material_np(res=1024, seed=0):
    """Material B of SURVEY §8d: diffuse U(0.2,0.8), roughness U(0.3,0.9); interior so FD is legal."""
    rng = np.random.default_rng(seed)
    m = np.empty((res, res, 4), np.float32)
    m[..., :3] = rng.uniform(0.2, 0.8, (res, res, 3))
    m[..., 3] = rng.uniform(0.3, 0.9, (res, res))
    return m


@pytest.fixture(scope="session")
def cbox_arrays():
    from zdr_amd import geometry
    return geometry.assemble(cbox_models())


@pytest.fixture(scope="session")
def cbox_oracle(cbox_arrays):
    import oracle
    return oracle.OracleScene.from_arrays(cbox_arrays)


@pytest.fixture(scope="session")
def cbox_material():
    return cbox_material_np()


@pytest.fixture(scope="session")
def fd_material():
    return fd_material_np(256, 0)
