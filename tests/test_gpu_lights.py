"""-m gpu: scenes with SEVERAL mesh lights of different triangle counts (light.py:33-48: light -> instance ->
triangle range; sample_light_pdf's n * T, light.py:105-110) and Scene.update_lights switching them
(render.py:130-148, test_lightstage.py:24-62), HIP path against the oracle."""
import numpy as np
import pytest
import torch

import oracle
from conftest import cbox_material_np, fd_material_np
from gpu_util import Flips, assert_grad_parity, assert_image_parity, make_scene, multi_light_arrays, oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stage():
    A = multi_light_arrays()
    return A, oracle.OracleScene.from_arrays(A), oracle.OracleScene.from_arrays(A, variant="fma")


@pytest.mark.parametrize("integrator", ["direct", "path"])
@pytest.mark.parametrize("accel", ["brute", "bvh"])
def test_three_lights_forward_and_backward(integrator, accel, stage):
    A, S, Sf = stage
    S.set_emissions(A.inst_emission); Sf.set_emissions(A.inst_emission)
    scene = make_scene(integrator, arrays=A, accel=accel)
    assert scene.info()["light_count"] == 3 and scene.light_count == 3
    mat = cbox_material_np()                                     # rough material: tight bounds, no floor
    W, spp, seed = 96, 16, 11
    cot = np.random.default_rng(4).uniform(0.5, 1.5, (W, W, 4)).astype(np.float32)
    m = torch.from_numpy(mat).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=seed)
    ref = S.render_forward(oracle_params(scene, W, W, spp, seed, mat.shape[:2]), mat)
    assert ref[..., :3].mean() > 0.1
    assert_image_parity(img.detach().cpu().numpy()[..., :3], ref[..., :3], f"three lights {integrator}/{accel} forward")
    (img * torch.from_numpy(cot).cuda()).sum().backward()
    gref = S.render_backward(oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]), cot, mat)
    assert_grad_parity(m.grad.cpu().numpy(), gref, f"three lights {integrator}/{accel} backward")


def test_three_lights_glossy_material(stage):
    A, S, Sf = stage
    S.set_emissions(A.inst_emission); Sf.set_emissions(A.inst_emission)
    scene = make_scene("path", arrays=A)
    mat = fd_material_np(256, 0)
    W, spp, seed = 64, 16, 3
    m = torch.from_numpy(mat).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=seed)
    p = oracle_params(scene, W, W, spp, seed, mat.shape[:2])
    assert_image_parity(img.detach().cpu().numpy()[..., :3], S.render_forward(p, mat)[..., :3], "three lights glossy forward",
                        floor=Sf.render_forward(p, mat)[..., :3], flips=Flips(scene, S, Sf, mat, (W, W), spp, seed, what="three lights glossy forward"))
    img.sum().backward()
    pb = oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]); ones = np.ones((W, W, 4), np.float32)
    assert_grad_parity(m.grad.cpu().numpy(), S.render_backward(pb, ones, mat), "three lights glossy backward",
                       floor=Sf.render_backward(pb, ones, mat), flips=Flips(scene, S, Sf, mat, (W, W), spp, seed + 1, cot=ones, what="three lights glossy backward"))


@pytest.mark.parametrize("integrator", ["direct", "path"])
def test_update_lights_against_the_oracle(integrator, stage):
    """Light-stage switching: every step changes which instances emit (and so light_count, the light list and the
    flat light table) and is compared with the oracle's set_emissions — forward and backward."""
    A, S, Sf = stage
    scene = make_scene(integrator, arrays=A)
    mat = cbox_material_np()
    m = torch.from_numpy(mat).cuda()
    W, spp = 64, 16
    # the paths that take another branch than the oracle's are MEASURED per step (path integrator: the dump exists for it alone)
    flips = (lambda seed, cot, what: Flips(scene, S, Sf, mat, (W, W), spp, seed, cot=cot, what=what)) if integrator == "path" else (lambda *a: None)
    ones = np.ones((W, W, 4), np.float32)
    steps = [
        [None, 20.0, None, None, None],                          # the ceiling light alone: light_count 1 (light0_T short cut)
        [None, None, (6.0, 2.0, 1.0), None, None],               # only the 8-triangle panel: light 0 is another instance
        [None, None, (6.0, 2.0, 1.0), None, (1.0, 3.0, 8.0)],    # two lights, neither is instance 1
        [None, 20.0, (6.0, 2.0, 1.0), 5.0, (1.0, 3.0, 8.0)],     # the blocker becomes the fourth light
        [None, 0.0, 0.0, 0.0, 0.0],                              # dark
        [None, 20.0, None, None, (1.0, 3.0, 8.0)],
    ]
    for k, em in enumerate(steps):
        scene.update_lights(em)
        e = np.stack([np.zeros(3) if x is None else np.broadcast_to(np.asarray(x, np.float32), (3,)) for x in em]).astype(np.float32)
        S.set_emissions(e); Sf.set_emissions(e)
        n = int((e > 0).any(axis=1).sum())
        assert scene.info()["light_count"] == n == scene.light_count
        img = scene.render_forward(m, (W, W), spp, 20 + k).cpu().numpy()
        ref = S.render_forward(oracle_params(scene, W, W, spp, 20 + k, mat.shape[:2]), mat)
        if n == 0:
            assert img[..., :3].max() == 0.0 and ref[..., :3].max() == 0.0
            continue
        assert_image_parity(img[..., :3], ref[..., :3], f"update_lights step {k} {integrator} forward", flips=flips(20 + k, None, f"update_lights step {k} forward"))
        g = torch.zeros_like(m)
        scene.render_backward(torch.from_numpy(ones).cuda(), g, m, (W, W), spp, 20 + k)
        gref = S.render_backward(oracle_params(scene, W, W, spp, 21 + k, mat.shape[:2]), ones, mat)
        assert_grad_parity(g.cpu().numpy(), gref, f"update_lights step {k} {integrator} backward", flips=flips(21 + k, ones, f"update_lights step {k} backward"))
    S.set_emissions(A.inst_emission); Sf.set_emissions(A.inst_emission)


def test_backward_replays_the_emission_snapshot_of_its_forward(stage):
    """render.py:216-222: backward re-uploads the emissions the forward saw (the scene stays at the snapshot)."""
    A, S, Sf = stage
    scene = make_scene("path", arrays=A)
    mat = cbox_material_np()
    m = torch.from_numpy(mat).cuda().requires_grad_()
    first = [None, 20.0, (6.0, 2.0, 1.0), None, None]
    scene.update_lights(first)
    img = scene.render(m, res=(48, 48), spp=16, seed=2)
    scene.update_lights([None, None, None, None, 9.0])           # changed between forward and backward
    img.sum().backward()
    e = np.zeros((5, 3), np.float32); e[1] = 20.0; e[2] = (6.0, 2.0, 1.0)
    S.set_emissions(e); Sf.set_emissions(e)
    ones = np.ones((48, 48, 4), np.float32)
    gref = S.render_backward(oracle_params(scene, 48, 48, 16, 3, mat.shape[:2]), ones, mat)
    assert scene.emissions is first                              # the scene was left at the forward's snapshot: the dump below sees the same lights
    assert_grad_parity(m.grad.cpu().numpy(), gref, "backward under the forward's emission snapshot",
                       flips=Flips(scene, S, Sf, mat, (48, 48), 16, 3, cot=ones, what="emission snapshot backward"))
    S.set_emissions(A.inst_emission); Sf.set_emissions(A.inst_emission)
