"""-m gpu: the single-GPU BASELINE configurations at their REAL resolution and sample count (BASELINE.json configs[1..3]),
image and PRB gradient against the oracle — not at the reduced sizes the other parity tests use.  The oracle needs about 20 s of
the GPU box's host cores per configuration (OpenMP over pixels); the bars are the base bars of tests/gpu_util.py (cbox textures:
roughness 1, no glossy ruler needed).  At these sizes a per-path dump of the whole render (67 M paths) is out of reach, so the few
paths that take another branch (0 - 3 per 65,536 on this material, tests/test_gpu_paths.py) stay inside the image statistics; they
fit the base bars with an order of magnitude to spare (profiles/r3_full_size_parity.json: frac_bad 2.4e-4 of 2e-3).
Procedure: /root/reference/fd_validate.py:21-33 (scene, camera), benchmark.py:26-39 (forward + backward of the summed image)."""
import numpy as np
import pytest
import torch

from gpu_util import assert_grad_parity, assert_image_parity, make_scene, oracle_params

pytestmark = pytest.mark.gpu


def _cot(H, W, seed):
    return np.random.default_rng(seed).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)


def test_c2_direct_512_spp64_at_full_size(cbox_arrays, cbox_material):
    """BASELINE configs[1]: cbox, direct integrator, 512 x 512, spp 64 (16.8 M camera samples) — forward (the configuration's
    own leg) and the backward pass bench.py times beside it."""
    import oracle
    S = oracle.OracleScene.from_arrays(cbox_arrays)
    scene = make_scene("direct")
    mat = cbox_material
    W, spp, seed = 512, 64, 11
    m = torch.from_numpy(mat).cuda()
    img = scene.render_forward(m, (W, W), spp, seed).cpu().numpy()
    ref = S.render_forward(oracle_params(scene, W, W, spp, seed, mat.shape[:2]), mat)
    assert_image_parity(img[..., :3], ref[..., :3], "c2 direct 512^2 spp 64 forward")
    assert (img[..., 3] == 1.0).all()
    cot = _cot(W, W, 2)
    g = torch.zeros_like(m)
    scene.render_backward(torch.from_numpy(cot).cuda(), g, m, (W, W), spp, seed)
    gref = S.render_backward(oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]), cot, mat)
    assert_grad_parity(g.cpu().numpy(), gref, "c2 direct 512^2 spp 64 backward")
    scene.check()


def test_c3_path_512_spp256_at_full_size(cbox_arrays, cbox_material):
    """BASELINE configs[2], the headline: cbox, path integrator, 512 x 512, spp 256 — 67.1 M camera samples per pass, forward image
    and PRB gradient w.r.t. the cboxd / cboxr textures."""
    import oracle
    S = oracle.OracleScene.from_arrays(cbox_arrays)
    scene = make_scene("path")
    mat = cbox_material
    W, spp, seed = 512, 256, 7
    m = torch.from_numpy(mat).cuda().requires_grad_()
    cot = _cot(W, W, 1)
    img = scene.render(m, res=(W, W), spp=spp, seed=seed)           # through the autograd boundary, as benchmark.py does
    (img * torch.from_numpy(cot).cuda()).sum().backward()
    scene.check()
    ref = S.render_forward(oracle_params(scene, W, W, spp, seed, mat.shape[:2]), mat)
    assert_image_parity(img.detach().cpu().numpy()[..., :3], ref[..., :3], "c3 path 512^2 spp 256 forward")
    gref = S.render_backward(oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]), cot, mat)
    g = m.grad.cpu().numpy()
    st = assert_grad_parity(g, gref, "c3 path 512^2 spp 256 backward")
    assert abs(st["nnz_got"] - st["nnz_ref"]) <= 1e-5 * st["nnz_ref"], st     # the same texels receive a gradient


def test_c4_path_1024_eight_tile_shards(cbox_arrays, cbox_material):
    """BASELINE configs[3]'s frame and partition on ONE GPU: cbox, path, 1024 x 1024, rendered as the 8 interleaved tile shards the
    8 ranks of the multi-GPU path render (zdr_amd/distributed.py, include/zdr.h tile_shard_*), their union / sum against the
    oracle's unsharded render.  spp 64 instead of 1024 — 67.1 M samples per pass, what the oracle follows in 20 s; the full
    1.07 G-sample frame is profiles/r2_full_size_parity_c4.json.  The RCCL reduce of the real run adds the very tensors summed here."""
    import oracle
    S = oracle.OracleScene.from_arrays(cbox_arrays)
    scene = make_scene("path")
    mat = cbox_material
    W, spp, seed, shards = 1024, 64, 5, 8
    m = torch.from_numpy(mat).cuda()
    img = torch.zeros((W, W, 4), device="cuda")
    for r in range(shards):
        scene.render_forward(m, (W, W), spp, seed, tile_shard=(r, shards), out=img)
    ref = S.render_forward(oracle_params(scene, W, W, spp, seed, mat.shape[:2]), mat)
    assert_image_parity(img.cpu().numpy()[..., :3], ref[..., :3], "c4 path 1024^2 spp 64, union of 8 tile shards, forward")
    assert (img[..., 3] == 1.0).all()                               # every pixel belongs to exactly one shard
    # and the union IS the unsharded image (bit for bit when both cut the sample range into the same chunks; a shard owns an eighth
    # of the tiles and so cuts finer: float re-association of 64 terms)
    torch.testing.assert_close(img, scene.render_forward(m, (W, W), spp, seed), rtol=1e-5, atol=1e-6)
    cot = _cot(W, W, 3)
    cd = torch.from_numpy(cot).cuda()
    g = torch.zeros_like(m)
    for r in range(shards):
        scene.render_backward(cd, g, m, (W, W), spp, seed, tile_shard=(r, shards))
    gref = S.render_backward(oracle_params(scene, W, W, spp, seed + 1, mat.shape[:2]), cot, mat)
    assert_grad_parity(g.cpu().numpy(), gref, "c4 path 1024^2 spp 64, sum of 8 tile shards, backward")
    scene.check()
