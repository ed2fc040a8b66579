"""The C-ABI from a C caller: tests/helpers/c_client.c (C99, include/zdr.h, hipMalloc'ed buffers, no Python, no torch in its process)
renders the Cornell box forward and backward through libzdr_hip.so; its output must be what zdr_amd.Scene produces through ctypes —
the image bit for bit, the gradient up to the arrival order of the float atomics."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, cbox_models
from zdr_amd import _native, geometry

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_client(tmp_path):
    _native.lib()                                               # builds libzdr_hip.so if it is stale
    libdir = os.path.dirname(_native.LIB_PATH)
    exe = str(tmp_path / "c_client")
    # (-D__HIP_PLATFORM_AMD__: what hipcc defines by itself; hip_runtime_api.h wants it from a bare gcc)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-D__HIP_PLATFORM_AMD__", "-isystem", f"{ROCM}/include", f"-I{ROOT}/include",
                    os.path.join(ROOT, "tests", "helpers", "c_client.c"), "-o", exe, f"-L{libdir}", "-lzdr_hip", f"-L{ROCM}/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{ROCM}/lib"], check=True, capture_output=True, text=True)
    return exe


def test_the_c_client_compiles_and_links_against_the_header(tmp_path):
    exe = build_client(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)   # no arguments: usage, before any HIP call
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_a_c_caller_gets_what_the_python_host_gets(integrator, tmp_path):
    import torch
    from zdr_amd.scenes import cbox_camera, cbox_material_np, make_scene
    exe = build_client(tmp_path)
    A = geometry.assemble(cbox_models())
    cam = cbox_camera()
    mat = cbox_material_np()
    W, H, spp, seed = 96, 64, 16, 11
    cot = np.random.default_rng(3).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)
    scene = make_scene(integrator)
    head = np.array([A.verts.shape[0], A.tris.shape[0], A.ninst, W, H, spp, seed, mat.shape[0], mat.shape[1], _native.INTEGRATORS[integrator],
                     int(scene.use_tent_filter), scene.max_depth, scene.rr_depth, _native.ACCELS["auto"], 0, 0], np.int32)
    camera = np.array([cam.fov, *cam.origin, *cam.target, *cam.up], np.float32)
    with open(tmp_path / "scene.bin", "wb") as f:
        for a in (head, camera, A.verts, A.tris, A.inst_tri_begin, A.inst_xform, A.inst_emission, mat, cot):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([exe, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok") and "refused" in r.stdout
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    img = out[:H * W * 4].reshape(H, W, 4); grad = out[H * W * 4:].reshape(mat.shape)
    m = torch.from_numpy(mat).cuda()
    ref = scene.render_forward(m, (W, H), spp, seed)
    gref = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), gref, m, (W, H), spp, seed)
    scene.check()
    ref, gref = ref.cpu().numpy(), gref.cpu().numpy()
    assert ref[..., :3].mean() > 0.05 and np.abs(gref).sum() > 0
    assert np.array_equal(img, ref)                                                       # same library, same parameters: bit for bit
    np.testing.assert_allclose(grad, gref, rtol=1e-5, atol=1e-6 * float(np.abs(gref).max()))   # float atomics: arrival order
