import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from zdr_amd.scenes import ASSETS, CBOX_CAMERA, cbox_material_np, cbox_models, fd_material_np  # noqa: E402,F401  (the workloads live in the product; re-exported for the tests)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


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
