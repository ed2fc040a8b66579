"""Golden fixtures (tests/golden/cbox_golden.npz, made by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them exactly (CPU), the HIP path must match them within
the stated fp32 tolerance (GPU)."""
import os

import numpy as np
import pytest

import oracle
from conftest import CBOX_CAMERA, GOLDEN

sys_path_golden = os.path.join(GOLDEN, "cbox_golden.npz")


def load():
    return np.load(sys_path_golden)


def cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m.CASES


@pytest.mark.parametrize("case", cases(), ids=lambda c: c[0])
def test_oracle_reproduces_golden(case, cbox_oracle):
    name, integ, W, spp, seed, tent = case
    G = load(); mat = G["material"]
    p = oracle.make_params(integ, W, W, spp, seed, CBOX_CAMERA, mat.shape[:2], use_tent=tent, nthreads=1)
    assert np.array_equal(cbox_oracle.render_forward(p, mat), G[name + "/image"])
    pb = oracle.make_params(integ, W, W, spp, seed + 1, CBOX_CAMERA, mat.shape[:2], use_tent=tent, nthreads=1)
    g = cbox_oracle.render_backward(pb, np.ones((W, W, 4), np.float32), mat)
    np.testing.assert_allclose(g, G[name + "/grad"], rtol=1e-6, atol=1e-9)


def test_oracle_sampler_reproduces_golden():
    G = load()
    got = np.stack([oracle.sampler_dump(oracle.SAMPLER_CMJ, 24, 345, 0, 16, i, nvert=3) for i in range(16)])
    assert np.array_equal(got.view(np.uint32), G["sampler_cmj_px24_py345_seed0_spp16"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases(), ids=lambda c: c[0])
def test_hip_matches_golden(case, cbox_arrays):
    import torch
    from gpu_util import Flips, assert_grad_parity, assert_image_parity, make_scene
    name, integ, W, spp, seed, tent = case
    G = load(); mat = G["material"]
    scene = make_scene(integ)
    scene.use_tent_filter = tent
    m = torch.from_numpy(mat).cuda().requires_grad_()
    img = scene.render(m, res=(W, W), spp=spp, seed=seed)
    img.sum().backward()
    # glossy material: calibrate with the fma build of the oracle (see gpu_util.assert_image_parity)
    Sf = oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")
    p = oracle.make_params(integ, W, W, spp, seed, CBOX_CAMERA, mat.shape[:2], use_tent=tent)
    pb = oracle.make_params(integ, W, W, spp, seed + 1, CBOX_CAMERA, mat.shape[:2], use_tent=tent)
    S = oracle.OracleScene.from_arrays(cbox_arrays)
    ones = np.ones((W, W, 4), np.float32)
    got_img, got_grad = img.detach().cpu().numpy()[..., :3], m.grad.cpu().numpy()
    if integ != "path":        # one or two vertices per sample: no flipped-path bookkeeping (the dump exists for the path integrator alone)
        assert_image_parity(got_img, G[name + "/image"][..., :3], "golden " + name, floor=Sf.render_forward(p, mat)[..., :3])
        assert_grad_parity(got_grad, G[name + "/grad"], "golden grad " + name, floor=Sf.render_backward(pb, ones, mat))
        return
    # the paths that measurably took another branch than the oracle's are set aside; the rest is held to the bars
    ff = Flips(scene, S, Sf, mat, (W, W), spp, seed, what="golden " + name)
    fb = Flips(scene, S, Sf, mat, (W, W), spp, seed + 1, cot=ones, what="golden grad " + name)
    if W * W * spp >= 30000:
        assert_image_parity(got_img, G[name + "/image"][..., :3], "golden " + name, floor=Sf.render_forward(p, mat)[..., :3], flips=ff)
        assert_grad_parity(got_grad, G[name + "/grad"], "golden grad " + name, floor=Sf.render_backward(pb, ones, mat), flips=fb)
    else:
        # A few thousand glossy paths: whole-image statistics are a handful of heavy-tailed terms and their ratio to the FMA
        # ruler is noise (profiles/r3_glossy_floor_ratio.txt).  Compared path by path instead: the traces add up to the golden
        # image / gradient wherever no path flipped, and the typical path agrees to rounding.
        from path_trace import all_queries, deviation_percentiles, image_from_paths, scatter_gradients
        ff.check_count("golden " + name); fb.check_count("golden grad " + name)
        q = all_queries(W, W, spp)
        st = deviation_percentiles(fb.hip, fb.ref)
        print(f"[paths] golden {name}: {st}")
        fl = deviation_percentiles(fb.fma, fb.ref)                      # the same percentiles between the oracle's two builds
        for key in ("L", "grad"):
            assert st[key][50] <= 2e-5 and st[key][90] <= max(1e-3, 2 * fl[key][90]), (key, st, fl)
        keep = ~ff.pixels
        ti = image_from_paths(ff.ref, q, W, W, spp)                     # the oracle's traces ARE the golden image ...
        assert np.abs(ti - G[name + "/image"][..., :3])[keep].max() <= 1e-5 * (1 + np.abs(ti).max())
        hi = image_from_paths(ff.hip, q, W, W, spp)                     # ... and the kernels' traces are the kernels' image
        assert np.abs(hi - got_img).max() <= 1e-5 * (1 + np.abs(hi).max())
        keep = ~fb.texels
        tg, hg = scatter_gradients(fb.ref, *mat.shape[:2]), scatter_gradients(fb.hip, *mat.shape[:2])
        assert np.abs(tg - G[name + "/grad"])[keep].sum() <= 1e-4 * np.abs(tg).sum()
        assert np.abs(hg - got_grad).sum() <= 2e-4 * np.abs(hg).sum()


@pytest.mark.gpu
def test_hip_sampler_matches_golden():
    import torch
    from gpu_util import make_scene
    G = load()
    scene = make_scene("path")
    q = torch.tensor([[24, 345, i] for i in range(16)], dtype=torch.int32, device="cuda")
    got = scene.sampler_dump(q, 16, seed=0, nvert=3).cpu().numpy()
    exp = G["sampler_cmj_px24_py345_seed0_spp16"]
    assert np.array_equal(got[:, :exp.shape[1]].view(np.uint32), exp.view(np.uint32))
