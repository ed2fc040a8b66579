"""-m gpu: the HIP path kernels against the oracle PATH BY PATH (zdr_path_dump / zdro_path_dump).

Whole-image statistics cannot separate "a comparison flipped in the last ulp and the path took another branch" from "the
arithmetic is wrong": glossy materials amplify rounding differences so much that two correct float32 builds of the SAME
source differ in several per cent of the pixels.  Here every path of a small render is compared on its own:
  1. the traces add up to what the real kernels produced (image, gradient texture): the dump is the kernels' arithmetic;
  2. paths whose discrete decisions agree with the oracle's agree in value — the median path to ~1e-6, and the tail no
     worse than what separates the oracle's own IEEE and FMA builds;
  3. the paths whose decisions differ are counted, and bounded by the same calibration.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import cbox_material_np, fd_material_np
from gpu_util import assert_grad_parity, assert_image_parity, make_scene, multi_light_arrays, oracle_params
from path_trace import Trace, all_queries, deviation_percentiles, image_from_paths, scatter_gradients

pytestmark = pytest.mark.gpu


def run_case(scene, S, Sf, mat, W, H, spp, seed, what):
    m = torch.from_numpy(mat).cuda()
    q = all_queries(W, H, spp)
    qd = torch.from_numpy(q).cuda()
    cot = np.random.default_rng(7).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)
    cd = torch.from_numpy(cot).cuda()
    # the real kernels
    img = scene.render_forward(m, (W, H), spp, seed).cpu().numpy()
    g = torch.zeros_like(m); scene.render_backward(cd, g, m, (W, H), spp, seed - 1)      # render_backward adds 1 (render.py:196)
    g = g.cpu().numpy()
    # the traces: the forward's paths with the cotangent, i.e. exactly the paths the backward call above walked
    tr = Trace(scene.path_dump(m, qd, (W, H), spp, seed, d_image=cd).cpu().numpy())
    p = oracle_params(scene, W, H, spp, seed, mat.shape[:2])
    ref = Trace(S.path_dump(p, mat, q, d_image=cot))
    flo = Trace(Sf.path_dump(p, mat, q, d_image=cot))
    # 1. the traces are the kernels' arithmetic
    ti = image_from_paths(tr, q, W, H, spp)
    tg = scatter_gradients(tr, *mat.shape[:2])
    di = np.abs(ti - img[..., :3]); dg = np.abs(tg - g)
    cons = {"image_max_rel": float((di / (1e-6 + np.abs(ti))).max()), "image_frac_off": float((di > 1e-5 * (1 + np.abs(ti))).mean()),
            "grad_rel_l1": float(dg.sum() / max(np.abs(tg).sum(), 1e-30)), "grad_frac_off": float((dg > 1e-5 * np.abs(tg).max() + 1e-4 * np.abs(tg)).mean())}
    st = deviation_percentiles(tr, ref)
    fl = deviation_percentiles(flo, ref)
    print(f"[paths] {what}: kernels vs own traces {cons}\n[paths] {what}: HIP vs oracle {st}\n[paths] {what}: oracle fma vs ieee {fl}")
    return cons, st, fl


def check(cons, st, fl, what, tight):
    n = st["paths"]
    # 1. dump == kernels (same device functions; only the order of float additions differs)
    assert cons["image_frac_off"] <= 2e-3 and cons["grad_frac_off"] <= 2e-3 and cons["grad_rel_l1"] <= 2e-4, (what, cons)
    # 3. flipped paths: no more than twice the floor (at least 5: tiny counts fluctuate)
    assert st["flipped"] <= max(5, 2 * fl["flipped"], 2e-4 * n), (what, st["flipped"], fl["flipped"])
    # 2. agreeing paths: the typical path is exact to rounding; the tail is bounded by the calibration
    for key in ("L", "grad"):
        assert st[key][50] <= 2e-5, (what, key, st[key])
        if tight:
            assert st[key][99] <= 1e-3 and st[key][100] <= 5e-2, (what, key, st[key])
        else:
            assert st[key][90] <= max(2e-4, 3 * fl[key][90]) and st[key][99] <= max(2e-3, 3 * fl[key][99]), (what, key, st[key], fl[key])


@pytest.fixture(scope="module")
def oracles(cbox_arrays):
    return oracle.OracleScene.from_arrays(cbox_arrays), oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")


@pytest.mark.parametrize("accel", ["brute", "bvh"])
@pytest.mark.parametrize("material", ["A", "B"])
def test_cbox_paths(accel, material, oracles):
    mat = cbox_material_np() if material == "A" else fd_material_np(256, 0)
    scene = make_scene("path", accel=accel)
    what = f"cbox material {material} / {accel}"
    cons, st, fl = run_case(scene, oracles[0], oracles[1], mat, 40, 32, 16, 12345, what)
    check(cons, st, fl, what, tight=(material == "A"))


@pytest.mark.parametrize("W,H,spp", [
    (4, 4, 4096),        # w = 4095: the permutations' (i & w) >> 11 term is live — the packed pass takes its general branch
    (2, 2, 65536),       # w = 0xffff: the largest sample count whose permutation state fits the 16-bit halves
    (2, 1, 131072),      # beyond it: every draw goes through the one-by-one sampler again
    (8, 4, 48),          # not a power of two: the cycle-walking permutation and exact divisions, one by one
])
def test_paths_at_sample_counts_on_the_samplers_other_branches(W, H, spp, oracles):
    """cmj_vertex_samples (sampler.h) draws a vertex's numbers with two permutations per register when spp and the strata grid are
    powers of two and spp <= 65536; these renders sit on and beyond the edges of that, path by path against the oracle."""
    scene = make_scene("path")
    mat = cbox_material_np()
    what = f"cbox material A, {W}x{H} spp {spp}"
    cons, st, fl = run_case(scene, oracles[0], oracles[1], mat, W, H, spp, 31, what)
    check(cons, st, fl, what, tight=True)


def test_box_filter_and_roulette_from_the_first_vertex(oracles):
    scene = make_scene("path")
    scene.use_tent_filter = False
    scene.rr_depth = 0
    mat = fd_material_np(256, 0)
    what = "cbox material B, box filter, rr_depth 0"
    cons, st, fl = run_case(scene, oracles[0], oracles[1], mat, 32, 32, 16, 99, what)
    check(cons, st, fl, what, tight=False)


def test_three_lights_paths():
    A = multi_light_arrays()
    S, Sf = oracle.OracleScene.from_arrays(A), oracle.OracleScene.from_arrays(A, variant="fma")
    scene = make_scene("path", arrays=A)
    mat = fd_material_np(256, 0)
    what = "three lights, material B"
    cons, st, fl = run_case(scene, S, Sf, mat, 32, 32, 16, 5, what)
    check(cons, st, fl, what, tight=False)


def test_environment_paths(cbox_arrays):
    from test_envmap import sun_sky
    from zdr_amd import envmap
    I = envmap.prepare_image(sun_sky())
    tabs = envmap.build_tables(I)
    S, Sf = oracle.OracleScene.from_arrays(cbox_arrays), oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")
    S.set_envmap(I, *tabs); Sf.set_envmap(I, *tabs)
    scene = make_scene("path")
    scene.add_envmap(sun_sky())
    mat = fd_material_np(256, 0)
    what = "cbox + environment, material B"
    cons, st, fl = run_case(scene, S, Sf, mat, 32, 32, 16, 8, what)
    check(cons, st, fl, what, tight=False)


def test_queries_outside_the_render_read_as_zero_rows():
    """zdr.h, zdr_path_dump: a pixel outside the image or a sample index >= spp must not be walked (the cotangent
    would be read out of bounds); the row is all zeros and the valid rows are untouched by their neighbours."""
    scene = make_scene("path")
    m = torch.from_numpy(fd_material_np(64, 0)).cuda()
    W, H, spp = 16, 8, 4
    q = torch.tensor([[3, 2, 1], [-1, 0, 0], [W, 0, 0], [0, H, 0], [0, -5, 0], [0, 0, spp], [3, 2, 1]], dtype=torch.int32).cuda()
    cot = torch.ones((H, W, 4), device="cuda")
    out = scene.path_dump(m, q, (W, H), spp, 3, d_image=cot, maxv=4).cpu().numpy()
    assert np.all(out[1:6] == 0.0)
    assert np.array_equal(out[0], out[6]) and np.any(out[0] != 0.0)
    scene.check()
