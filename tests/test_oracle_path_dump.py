"""The oracle's per-path trace (zdro_path_dump) is the oracle's own render, taken apart: the paths' radiances add up to
its image and the per-vertex gradients, scattered like interaction.py:73-89, add up to its gradient texture."""
import numpy as np

import oracle
from conftest import CBOX_CAMERA, fd_material_np
from path_trace import Trace, all_queries, image_from_paths, scatter_gradients


def test_paths_add_up_to_the_render(cbox_oracle):
    mat = fd_material_np(64, 1)
    W, H, spp, seed = 20, 12, 16, 7
    p = oracle.make_params("path", W, H, spp, seed, CBOX_CAMERA, mat.shape[:2])
    q = all_queries(W, H, spp)
    cot = np.random.default_rng(0).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)
    tr = Trace(cbox_oracle.path_dump(p, mat, q, d_image=cot))
    assert tr.nvert.max() <= 16 and tr.nvert.min() == 0 and (tr.nvert >= 3).any()
    img, cnt = cbox_oracle.render_forward(p, mat, counters=True)
    assert int(tr.nvert.sum()) == cnt["shaded_vertices"]
    np.testing.assert_allclose(image_from_paths(tr, q, W, H, spp), img[..., :3], rtol=2e-6, atol=1e-7)
    grad = cbox_oracle.render_backward(p, cot, mat)          # same seed: the dump walks the paths of `p`
    np.testing.assert_allclose(scatter_gradients(tr, *mat.shape[:2]), grad, rtol=1e-5, atol=1e-6 * np.abs(grad).max())
    # flags: a vertex whose path goes on has a sampled direction and a pdf; Russian roulette only from depth 2 on
    went_on = (tr.flags & 2) != 0
    assert (tr.pdf[went_on & tr.live] > 0).all() and np.allclose(np.linalg.norm(tr.wi[went_on & tr.live], axis=1), 1.0, atol=1e-5)
    assert ((tr.flags[:, :2] >> 2) == 0).all()
    assert ((tr.flags >> 2) != 0).any()


def test_the_fma_build_flips_few_paths_and_agrees_on_the_rest(cbox_arrays, cbox_oracle):
    """Calibration of the fp32 floor, path by path: the same source compiled with FMA contraction takes another branch
    on ~0.1 % of the paths of a glossy material; on the rest the median path agrees to ~5e-6 of its own scale, but the
    tail is heavy (visible-normal sampling takes sqrt(1 - |p|^2) near the disk rim: a path can keep every discrete
    decision and still leave a vertex in a direction 0.1 rad away)."""
    from path_trace import deviation_percentiles
    Sf = oracle.OracleScene.from_arrays(cbox_arrays, variant="fma")
    mat = fd_material_np(256, 0)
    W, H, spp = 32, 32, 16
    p = oracle.make_params("path", W, H, spp, 12345, CBOX_CAMERA, mat.shape[:2])
    q = all_queries(W, H, spp)
    a, b = Trace(cbox_oracle.path_dump(p, mat, q)), Trace(Sf.path_dump(p, mat, q))
    st = deviation_percentiles(b, a)
    print("fma vs ieee:", st)
    assert st["flipped"] <= 0.004 * a.n
    assert st["L"][50] < 2e-5 and st["grad"][50] < 2e-5 and st["L"][90] < 5e-4 and st["grad"][90] < 5e-4
    # flags alone would call 6 % of the paths flipped: every ceiling vertex sees the ceiling light edge-on and the
    # light-sample acceptance test (prb.py:62) is a coin toss there — with zero radiance at stake (path_trace.Trace.decisions)
    raw = (a.nvert != b.nvert) | np.any((a.flags != b.flags) & a.live & b.live, axis=1)
    assert raw.mean() > 5 * max(st["flipped"], 1) / a.n


def test_queries_outside_the_image_or_sample_range_read_as_zero_rows(cbox_oracle):
    """The twin of zdr_path_dump's guard (include/zdr.h): a query whose pixel lies outside the image or whose
    sample_index >= spp yields an all-zero row — also with a cotangent image, which must not be read out of bounds."""
    mat = fd_material_np(64, 0)
    W, H, spp = 16, 12, 4
    p = oracle.make_params("path", W, H, spp, 7, CBOX_CAMERA, mat.shape[:2])
    q = np.array([[3, 4, 1], [-1, 4, 1], [3, -2, 0], [W, 4, 1], [3, H, 1], [3, 4, spp], [3, 4, spp + 100], [W - 1, H - 1, spp - 1]], np.int32)
    cot = np.ones((H, W, 4), np.float32)
    for d_image in (None, cot):
        out = cbox_oracle.path_dump(p, mat, q, d_image=d_image)
        assert out[0].view(np.int32)[0] > 0 and out[7].view(np.int32)[0] >= 0
        assert (out[1:7] == 0).all()
