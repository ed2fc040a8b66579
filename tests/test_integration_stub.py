"""INTEGRATION.md shows the ctypes stub a maintainer of the reference would drop into its render.py.  These tests run THAT
text: the struct it declares must be the library's (CPU), and the class it defines must render the Cornell box through
libzdr_hip.so to the very image and gradient zdr_amd.Scene produces (GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, cbox_models
from zdr_amd import _native
from zdr_amd.load_obj import concat_triangles, read_obj


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    assert 'C.CDLL("libzdr_hip.so")' in code
    _native.lib()                                               # builds the library if it is stale
    code = code.replace('C.CDLL("libzdr_hip.so")', f'C.CDLL({_native.LIB_PATH!r})')
    ns = {"read_obj": read_obj, "concat_triangles": concat_triangles}   # the reference's own load_obj.py provides these
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    return ns


def test_the_stub_declares_the_librarys_struct():
    ns = _stub_namespace()
    P, Q = ns["_Params"], _native.RenderParams
    assert C.sizeof(P) == C.sizeof(Q)
    assert [(n, getattr(P, n).offset, getattr(P, n).size) for n, _ in P._fields_] == [(n, getattr(Q, n).offset, getattr(Q, n).size) for n, _ in Q._fields_]
    assert C.sizeof(ns["_Cam"]) == C.sizeof(_native.CameraPOD)
    assert ns["_INTEGRATOR"] == {k: v for k, v in _native.INTEGRATORS.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_the_stub_renders_what_the_package_renders(integrator):
    import torch
    from zdr_amd.scenes import cbox_camera, cbox_material_np, make_scene
    ns = _stub_namespace()
    stub = object.__new__(ns["Scene"])                          # the reference's Scene.__init__ needs luisa: only the replaced methods are run
    stub.camera, stub.integrator_name, stub.use_tent_filter = cbox_camera(), integrator, True
    stub.load_geometry(cbox_models())
    assert stub.inst_count == 2
    m = torch.from_numpy(cbox_material_np()).cuda()
    W, H, spp, seed = 96, 64, 16, 5
    img = stub.render_forward(m, (W, H), spp, seed)
    g = torch.zeros_like(m)
    cot = torch.rand((H, W, 4), device="cuda")
    stub.render_backward(cot, g, m, (W, H), spp, seed)
    torch.cuda.synchronize()
    scene = make_scene(integrator)
    ref = scene.render_forward(m, (W, H), spp, seed)
    gref = torch.zeros_like(m)
    scene.render_backward(cot, gref, m, (W, H), spp, seed)
    assert torch.equal(img, ref) and img[..., :3].mean() > 0.05                 # same library, same parameters: bit for bit
    torch.testing.assert_close(g, gref, rtol=1e-5, atol=1e-6 * float(gref.abs().max()))   # float atomics: arrival order
    ns["_L"].zdr_scene_destroy(stub._h)
