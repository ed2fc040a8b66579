"""Environment lighting (/root/reference/envmap.py): host tables, oracle estimators (CPU) and
HIP-vs-oracle parity (GPU)."""
import numpy as np
import pytest

import oracle
from conftest import CBOX_CAMERA, cbox_models, fd_material_np
from test_oracle_brdf import brdf64
from test_oracle_render import quad_scene
from zdr_amd import envmap, geometry


def sun_sky(seed=0):
    rng = np.random.default_rng(seed)
    img = rng.uniform(0.05, 0.6, (32, 64, 3)).astype(np.float32)
    img[5:8, 40:44] = (300.0, 260.0, 200.0)
    return img


@pytest.fixture(scope="module")
def sky_tables():
    I = envmap.prepare_image(sun_sky())
    return (I,) + envmap.build_tables(I)


def test_tables_are_a_normalised_density_and_the_alias_method_reproduces_it(sky_tables):
    I, prob, alias, pdf = sky_tables
    W, H = envmap.SAMPLE_MAP_W, envmap.SAMPLE_MAP_H
    assert I.shape == (64, 64, 4)                                   # 2:1 image made square by repeating rows (envmap.py:124-125)
    assert prob.shape == (H + H * W,) and alias.shape == prob.shape and pdf.shape == (H * W,)
    assert abs(pdf.mean() - 1.0) < 1e-5 and (pdf >= 0).all()
    assert (alias[:H] >= 0).all() and (alias[:H] < H).all() and (alias[H:] >= 0).all() and (alias[H:] < W).all()
    # exact probabilities implied by the alias tables == pdf
    q = np.zeros(H)
    for i in range(H):
        q[i] += min(prob[i], 1.0) / H; q[alias[i]] += max(0.0, 1.0 - prob[i]) / H
    np.testing.assert_allclose(q, pdf.reshape(H, W).mean(axis=1) / H, rtol=1e-4, atol=1e-9)
    y = int(np.argmax(q))
    qx = np.zeros(W); off = H + y * W
    for i in range(W):
        qx[i] += min(prob[off + i], 1.0) / W; qx[alias[off + i]] += max(0.0, 1.0 - prob[off + i]) / W
    row = pdf.reshape(H, W)[y]
    np.testing.assert_allclose(qx, row / row.sum(), rtol=1e-4, atol=1e-9)


def test_prepare_image_rejects_other_aspect_ratios():
    with pytest.raises(RuntimeError, match="1:2 or 1:1"):
        envmap.prepare_image(np.zeros((10, 30, 3), np.float32))


def test_constant_environment_gives_the_hemispherical_albedo():
    """furnace test on a single quad: radiance = L_env * integral of f cos; direct == path(max_depth 2) sample for sample"""
    S = oracle.OracleScene.from_arrays(quad_scene())
    cam = (0.5, (0.0, 2.0, 0.001), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    mat = np.zeros((4, 4, 4), np.float32); mat[..., :3] = (0.3, 0.5, 0.7); mat[..., 3] = 0.6
    I = envmap.prepare_image(np.full((32, 64, 3), 2.0, np.float32))
    S.set_envmap(I, *envmap.build_tables(I))
    W = 8
    a = S.render_forward(oracle.make_params("direct", W, W, 2048, 0, cam, (4, 4)), mat)
    b = S.render_forward(oracle.make_params("path", W, W, 2048, 0, cam, (4, 4), max_depth=2), mat)
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)       # same samples, same terms; only the association differs
    wo = np.array([0.0, 0.0, 1.0]); n = 300; acc = np.zeros(3)
    for c in (np.arange(n) + 0.5) / n:
        st = np.sqrt(1 - c * c)
        for ph in (np.arange(48) + 0.5) / 48 * 2 * np.pi:
            acc += brdf64(wo, np.array([st * np.cos(ph), st * np.sin(ph), c]), np.array([0.3, 0.5, 0.7]), 0.6)
    albedo = acc * (1.0 / n) * (2 * np.pi / 48)
    np.testing.assert_allclose(a[W // 2, W // 2, :3], 2.0 * albedo, rtol=1e-2)


def test_light_sampling_and_bsdf_sampling_agree_under_a_sun(sky_tables):
    # MIS combines both; an estimator that only ever finds the sun through BSDF sampling (no envmap importance
    # sampling: a uniform 'sampling map') must converge to the same value, with far more noise
    S = oracle.OracleScene.from_arrays(quad_scene())
    cam = (0.5, (0.0, 2.0, 0.001), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    mat = np.zeros((4, 4, 4), np.float32); mat[..., :3] = 0.6; mat[..., 3] = 0.8
    I, prob, alias, pdf = sky_tables
    W = 6
    S.set_envmap(I, prob, alias, pdf)
    good = np.mean([S.render_forward(oracle.make_params("path", W, W, 1024, s, cam, (4, 4), max_depth=2), mat)[..., :3].mean() for s in range(4)])
    Hm, Wm = envmap.SAMPLE_MAP_H, envmap.SAMPLE_MAP_W
    uprob = np.ones(Hm + Hm * Wm, np.float32); ualias = np.concatenate([np.arange(Hm), np.tile(np.arange(Wm), Hm)]).astype(np.int32)
    S.set_envmap(I, uprob, ualias, np.ones(Hm * Wm, np.float32))
    flat = np.mean([S.render_forward(oracle.make_params("path", W, W, 16384, s, cam, (4, 4), max_depth=2), mat)[..., :3].mean() for s in range(4)])
    assert abs(good - flat) / good < 0.03, (good, flat)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator,accel", [("direct", "brute"), ("path", "brute"), ("path", "bvh"), ("direct", "bvh")])
def test_hip_environment_lighting_matches_oracle(integrator, accel, cbox_arrays, sky_tables):
    import torch
    from gpu_util import assert_grad_parity, assert_image_parity, make_scene, oracle_params
    I, prob, alias, pdf = sky_tables
    scene = make_scene(integrator, accel=accel)
    scene.add_envmap(sun_sky())
    assert scene.env_count == 1 and np.array_equal(scene._envmap[1], prob)
    S = oracle.OracleScene.from_arrays(cbox_arrays); Sf = oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")
    S.set_envmap(I, prob, alias, pdf); Sf.set_envmap(I, prob, alias, pdf)
    mat = fd_material_np(256, 0)
    W, spp = 96, 16
    m = torch.from_numpy(mat).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=2)
    p = oracle_params(scene, W, W, spp, 2, mat.shape[:2])
    ref = S.render_forward(p, mat)
    assert ref[..., :3].mean() > 0.05
    assert_image_parity(img.detach().cpu().numpy()[..., :3], ref[..., :3], f"envmap {integrator} forward", floor=Sf.render_forward(p, mat)[..., :3])
    img.sum().backward()
    pb = oracle_params(scene, W, W, spp, 3, mat.shape[:2]); ones = np.ones((W, W, 4), np.float32)
    assert_grad_parity(m.grad.cpu().numpy(), S.render_backward(pb, ones, mat), f"envmap {integrator} backward", floor=Sf.render_backward(pb, ones, mat))
    # removing the environment restores the plain scene
    scene.add_envmap(None)
    plain = make_scene(integrator, accel=accel).render(m.detach(), res=(32, 32), spp=4)
    assert torch.equal(scene.render(m.detach(), res=(32, 32), spp=4), plain)


@pytest.mark.gpu
def test_hip_environment_only_scene_matches_oracle(sky_tables):
    # no mesh light at all: n = 1, every light sample goes to the environment
    import torch
    from gpu_util import assert_image_parity, make_scene, oracle_params
    I, prob, alias, pdf = sky_tables
    models = [(cbox_models()[0][0], None, 0.0)]
    scene = make_scene("path", models=models)
    scene.add_envmap(sun_sky())
    S = oracle.OracleScene.from_arrays(geometry.assemble(models)); S.set_envmap(I, prob, alias, pdf)
    mat = fd_material_np(128, 1)
    img = scene.render(torch.from_numpy(mat).cuda(), res=(64, 64), spp=16, seed=1).cpu().numpy()
    Sf = oracle.OracleScene.from_arrays(geometry.assemble(models), variant="fma"); Sf.set_envmap(I, prob, alias, pdf)
    p = oracle_params(scene, 64, 64, 16, 1, mat.shape[:2])
    assert_image_parity(img[..., :3], S.render_forward(p, mat)[..., :3], "envmap only", floor=Sf.render_forward(p, mat)[..., :3])


@pytest.mark.gpu
def test_environment_and_pmj02bn_tables_are_independent_state(cbox_arrays, sky_tables):
    """render.py:150-156 (add_envmap) and pmj02bn.py:9-18 (sampler tables) are independent state of a scene: setting
    the sampler tables after the environment must leave the environment buffers alone (round 1 freed them), in
    either order, and each buffer is released exactly once (its setter or zdr_scene_destroy)."""
    import ctypes as C
    import torch
    from gpu_util import Flips, assert_grad_parity, assert_image_parity, make_scene, oracle_params
    from zdr_amd import pmj02bn_tables as T
    I, prob, alias, pdf = sky_tables
    pmj = T.pmj02_sets(n_sets=5, n_samples=256, seed=2)
    bn = T.blue_noise_textures(n_tex=4, res=32, seed=2)
    oracle.lib().zdro_set_pmj02bn_tables(pmj.ctypes.data_as(C.POINTER(C.c_uint32)), 5, 256, bn.ctypes.data_as(C.POINTER(C.c_uint16)), 4, 32)
    S = oracle.OracleScene.from_arrays(cbox_arrays); Sf = oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")
    S.set_envmap(I, prob, alias, pdf); Sf.set_envmap(I, prob, alias, pdf)
    oracle.lib("fma").zdro_set_pmj02bn_tables(pmj.ctypes.data_as(C.POINTER(C.c_uint32)), 5, 256, bn.ctypes.data_as(C.POINTER(C.c_uint16)), 4, 32)
    mat = fd_material_np(256, 0)
    W, spp = 64, 16
    ones = np.ones((W, W, 4), np.float32)
    for order in ("env_first", "tables_first"):
        scene = make_scene("path")
        scene.sampler = "pmj02bn"
        if order == "env_first":
            scene.add_envmap(sun_sky())
            scene.set_pmj02bn_tables(pmj, bn)            # round 1: this call freed the environment buffers
        else:
            scene.set_pmj02bn_tables(pmj, bn)
            scene.add_envmap(sun_sky())
        m = torch.from_numpy(mat).cuda().requires_grad_()
        img = scene.render(m, res=(W, W), spp=spp, seed=4)
        img.sum().backward()
        p = oracle_params(scene, W, W, spp, 4, mat.shape[:2], sampler=oracle.SAMPLER_PMJ02BN)
        pb = oracle_params(scene, W, W, spp, 5, mat.shape[:2], sampler=oracle.SAMPLER_PMJ02BN)
        assert_image_parity(img.detach().cpu().numpy()[..., :3], S.render_forward(p, mat)[..., :3], f"envmap + pmj02bn forward ({order})",
                            floor=Sf.render_forward(p, mat)[..., :3],
                            flips=Flips(scene, S, Sf, mat, (W, W), spp, 4, what=f"envmap + pmj02bn forward ({order})", sampler=oracle.SAMPLER_PMJ02BN))
        assert_grad_parity(m.grad.cpu().numpy(), S.render_backward(pb, ones, mat), f"envmap + pmj02bn backward ({order})",
                           floor=Sf.render_backward(pb, ones, mat),
                           flips=Flips(scene, S, Sf, mat, (W, W), spp, 5, cot=ones, what=f"envmap + pmj02bn backward ({order})", sampler=oracle.SAMPLER_PMJ02BN))
        # new tables while an environment is set, then render again: the environment is still there
        scene.set_pmj02bn_tables(pmj, bn)
        again = scene.render(m.detach(), res=(W, W), spp=spp, seed=4)
        assert torch.equal(again, img.detach())
        scene.add_envmap(None)
        dark = scene.render(m.detach(), res=(32, 32), spp=4, seed=1)
        assert torch.isfinite(dark).all()
        del scene                                        # zdr_scene_destroy: every buffer freed once
        torch.cuda.synchronize()
