"""-m gpu: the light-stage scene as /root/reference/test_lightstage.py builds it (:24-62) — an object at the origin inside a sphere of
small quad lights, each placed by the script's own `rotate_mat` (yaw @ pitch @ translate, handed over column-major through
float4x4(*m.transpose().flatten())), camera sphere_camera1, integrator 'direct' — with sphere.obj (reference-pinned geometry,
tests/golden/obj_fixtures.npz; the script's bunnyuv.obj is not among the reference's files) as the object.  HIP against the oracle,
forward and backward, for the script's integrator and for `path`, and light switching through update_lights."""
from math import acos, cos, pi, sin

import numpy as np
import pytest
import torch

import oracle
from conftest import ASSETS, fd_material_np
from gpu_util import Flips, assert_grad_parity, assert_image_parity, oracle_params
from zdr_amd import Camera, Scene, float3, float4x4, geometry

pytestmark = pytest.mark.gpu


def rotate_mat(theta, phi, offset):                              # test_lightstage.py:24-45, verbatim in structure
    pitch = np.array([[cos(theta), -sin(theta), 0, 0], [sin(theta), cos(theta), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    yaw = np.array([[cos(phi), 0, -sin(phi), 0], [0, 1, 0, 0], [sin(phi), 0, cos(phi), 0], [0, 0, 0, 1]])
    translate = np.array([[1, 0, 0, offset[0]], [0, 1, 0, offset[1]], [0, 0, 1, offset[2]], [0, 0, 0, 1]])
    m = yaw @ pitch @ translate
    return float4x4(*m.transpose().flatten())


NLIGHT = 30
LIGHTS = (29, 22, 16, 9)                                         # the script keeps i == 29; three more of its 30 positions


def models():
    out = [(f"{ASSETS}/sphere.obj", rotate_mat(0, -0.4, (0, 0, 0)), None)]
    for i in LIGHTS:
        out.append((f"{ASSETS}/quad.obj", rotate_mat(acos((i + 0.5) / NLIGHT * 2 - 1), pi * 2 * 0.618 * (i + 1), (0, 0, 0)), 50))
    return out


CAMERA = Camera(fov=50 / 180 * 3.1415926, origin=float3(0, 0.5, 2), target=float3(0, 0, 0), up=float3(0.0, 1.0, 0.0))   # sphere_camera1


@pytest.fixture(scope="module")
def stage():
    A = geometry.assemble(models())
    return A, oracle.OracleScene.from_arrays(A), oracle.OracleScene.from_arrays(A, variant="fma")


@pytest.mark.parametrize("integrator", ["direct", "path"])
def test_light_stage_matches_the_oracle(integrator, stage):
    A, S, Sf = stage
    scene = Scene(models(), integrator=integrator)
    scene.camera = CAMERA
    assert scene.info()["accel"] == "bvh" and scene.info()["ntris"] == 960 + 2 * len(LIGHTS) and scene.light_count == len(LIGHTS)
    mat = fd_material_np(256, 3); mat[..., 3] = 0.6 + 0.4 * mat[..., 3]      # roughness 0.72 - 0.96: bars without the glossy ruler
    W, spp, seed = 96, 16, 8
    m = torch.from_numpy(mat).cuda().requires_grad_()
    cot = np.random.default_rng(5).uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
    img = scene.render(m, res=(W, W), spp=spp, seed=seed)
    (img * torch.from_numpy(cot).cuda()).sum().backward()
    scene.check()
    ref = S.render_forward(oracle_params(scene, W, W, spp, seed, mat.shape[:2]), mat)
    gref = S.render_backward(oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]), cot, mat)
    assert ref[..., :3].mean() > 0.01 and np.abs(gref).sum() > 0           # the lights reach the object
    path = integrator == "path"
    ff = Flips(scene, S, Sf, mat, (W, W), spp, seed, what="light stage forward") if path else None
    fb = Flips(scene, S, Sf, mat, (W, W), spp, seed + 1, cot=cot, what="light stage backward") if path else None
    p, pb = oracle_params(scene, W, W, spp, seed, mat.shape[:2]), oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2])
    assert_image_parity(img.detach().cpu().numpy()[..., :3], ref[..., :3], f"light stage {integrator} forward", flips=ff, floor=Sf.render_forward(p, mat)[..., :3])
    assert_grad_parity(m.grad.cpu().numpy(), gref, f"light stage {integrator} backward", flips=fb, floor=Sf.render_backward(pb, cot, mat))


def test_light_stage_switching(stage):
    """One light at a time, as a light stage is used (test_lightstage.py:66 keeps an update_lights line for it)."""
    A, S, Sf = stage
    scene = Scene(models(), integrator="direct")
    scene.camera = CAMERA
    mat = fd_material_np(128, 4); mat[..., 3] = 0.6 + 0.4 * mat[..., 3]
    m = torch.from_numpy(mat).cuda()
    W, spp = 64, 16
    means = []
    for k in range(len(LIGHTS)):
        em = [None] + [200 if j == k else 0 for j in range(len(LIGHTS))]
        scene.update_lights(em)
        e = np.zeros((1 + len(LIGHTS), 3), np.float32); e[1 + k] = 200
        S.set_emissions(e); Sf.set_emissions(e)
        img = scene.render_forward(m, (W, W), spp, 30 + k).cpu().numpy()
        p = oracle_params(scene, W, W, spp, 30 + k, mat.shape[:2])
        ref = S.render_forward(p, mat)
        # `direct` has no per-path dump: the samples that graze the sphere's silhouette (hit on one side, miss on the other) stay in,
        # and the oracle's FMA build, which has them too, is the ruler
        assert_image_parity(img[..., :3], ref[..., :3], f"light stage, light {LIGHTS[k]} alone", floor=Sf.render_forward(p, mat)[..., :3])
        means.append(float(ref[..., :3].mean()))
    S.set_emissions(A.inst_emission); Sf.set_emissions(A.inst_emission)
    assert max(means) > 0.01 and len({round(x, 5) for x in means}) >= 3      # the lights differ (one of the four sits behind the object as the camera sees it)
