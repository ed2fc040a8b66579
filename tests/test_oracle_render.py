"""Pins the oracle's estimators without the reference:
closed-form collocated, AD-vs-FD for direct (exactly linear in diffuse) and path, the PRB
adjoint correction (App. B-3), shard unions, and the path statistics of SURVEY §8d."""
import numpy as np
import pytest

import oracle
from conftest import CBOX_CAMERA
from test_oracle_brdf import brdf64
from zdr_amd import geometry


def quad_scene():
    # unit-ish quad in the y=0 plane, normal +y, uv = (x,z) mapped to [0,1]
    v = np.array([[-1, 0, -1, 0, 0, 0, 1, 0], [-1, 0, 1, 0, 1, 0, 1, 0], [1, 0, 1, 1, 1, 0, 1, 0], [1, 0, -1, 1, 0, 0, 1, 0]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return geometry.from_arrays(v, t)


def test_collocated_matches_closed_form():
    A = quad_scene()
    S = oracle.OracleScene.from_arrays(A)
    cam = (0.6, (0.1, 2.0, 0.2), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    mat = np.zeros((4, 4, 4), np.float32)
    mat[..., :3] = (0.3, 0.5, 0.7)
    mat[..., 3] = 0.6
    W = H = 16
    p = oracle.make_params("collocated", W, H, 1, 0, cam, (4, 4), use_tent=False)
    # box filter: pixel offset = first CMJ 2-D draw; rebuild the same rays in float64
    img = S.render_forward(p, mat)
    o = np.array(cam[1], np.float64)
    fwd = np.array(cam[2]) - o; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, cam[3]); right /= np.linalg.norm(right)
    upp = np.cross(right, fwd)
    for y in range(H):
        for x in range(W):
            off = oracle.sampler_dump(oracle.SAMPLER_CMJ, x, y, 0, 1, 0, nvert=0)
            px = (2.0 / W * (x + off[0]) - 1.0) * np.tan(0.3)
            py = (2.0 / H * (y + off[1]) - 1.0) * np.tan(0.3)
            d = px * right - py * upp + fwd; d /= np.linalg.norm(d)
            t = -o[1] / d[1]
            hit = o + t * d
            exp = np.zeros(3)
            if abs(hit[0]) <= 1 and abs(hit[2]) <= 1 and -d[1] >= 1e-4:
                wo = np.array([0.0, 0.0, -d[1]])  # local frame: only the z component matters for wo==wi
                s = np.sqrt(max(0.0, 1 - wo[2] ** 2)); wo[0] = s
                exp = brdf64(wo, wo, np.array([0.3, 0.5, 0.7]), 0.6) / (t * t)
            np.testing.assert_allclose(img[y, x, :3], exp, rtol=2e-4, atol=1e-7)
            assert img[y, x, 3] == 1.0


def directional(S, integ, mat, delta, W, H, spp, seed, wimg, eps, **kw):
    """returns (AD, FD) of d/dt sum(wimg * I(mat + t delta)) with the SAME seed in both."""
    th = mat.shape[:2]
    p = oracle.make_params(integ, W, H, spp, seed, CBOX_CAMERA, th, **kw)
    g = S.render_backward(p, wimg, mat)
    ad = float((g.astype(np.float64) * delta).sum())
    ip = S.render_forward(p, (mat + eps * delta).astype(np.float32)).astype(np.float64)
    im = S.render_forward(p, (mat - eps * delta).astype(np.float32)).astype(np.float64)
    fd = float(((ip - im) * wimg).sum() / (2 * eps))
    return ad, fd


def _weights(H, W, seed=3):
    w = np.random.default_rng(seed).uniform(0.5, 1.5, (H, W, 4)).astype(np.float32)
    w[..., 3] = 0
    return w


def test_direct_gradient_is_exact_for_diffuse(cbox_oracle, fd_material):
    # direct is exactly linear in the diffuse texels and its sampling ignores diffuse
    rng = np.random.default_rng(5)
    delta = np.zeros_like(fd_material); delta[..., :3] = rng.uniform(-1, 1, fd_material.shape[:2] + (3,))
    ad, fd = directional(cbox_oracle, "direct", fd_material, delta, 48, 48, 16, 11, _weights(48, 48), 1e-2)
    assert abs(ad - fd) / abs(fd) < 2e-4, (ad, fd)


def test_path_gradient_without_rr_matches_fd(cbox_oracle, fd_material):
    # RR off => the estimator is a polynomial in the diffuse texels; central FD error is O(eps^2)
    rng = np.random.default_rng(6)
    delta = np.zeros_like(fd_material); delta[..., :3] = rng.uniform(-1, 1, fd_material.shape[:2] + (3,))
    ad, fd = directional(cbox_oracle, "path", fd_material, delta, 40, 40, 16, 3, _weights(40, 40), 2e-3, max_depth=5, rr_depth=99)
    assert abs(ad - fd) / abs(fd) < 1e-3, (ad, fd)


def test_prb_literal_weight_is_not_the_derivative(cbox_oracle, fd_material):
    # App. B-3: prb.py:162 weights the remainder by beta/pdf instead of dividing it by f
    rng = np.random.default_rng(6)
    delta = np.zeros_like(fd_material); delta[..., :3] = rng.uniform(0, 1, fd_material.shape[:2] + (3,))
    kw = dict(max_depth=5, rr_depth=99)
    ad_ok, fd = directional(cbox_oracle, "path", fd_material, delta, 32, 32, 16, 3, _weights(32, 32), 2e-3, **kw)
    ad_lit, _ = directional(cbox_oracle, "path", fd_material, delta, 32, 32, 16, 3, _weights(32, 32), 2e-3, prb_mode=oracle.PRB_LITERAL, **kw)
    assert abs(ad_ok - fd) / abs(fd) < 1e-3
    assert abs(ad_lit - fd) / abs(fd) > 2e-2, (ad_lit, fd)


def test_path_gradient_with_rr_and_roughness_statistical(cbox_oracle, fd_material):
    # detached-sampling AD and same-seed FD agree in expectation only (roughness moves the
    # samples, RR makes FD discontinuous): compare at moderately high sample counts
    rng = np.random.default_rng(7)
    delta = rng.uniform(0, 1, fd_material.shape).astype(np.float32)  # diffuse AND roughness
    ad, fd = directional(cbox_oracle, "path", fd_material, delta, 32, 32, 256, 1, _weights(32, 32), 1e-2)
    assert abs(ad - fd) / abs(fd) < 3e-2, (ad, fd)


def test_shard_unions_equal_the_full_render(cbox_oracle, cbox_material):
    W = H = 32
    th = cbox_material.shape[:2]
    full = cbox_oracle.render_forward(oracle.make_params("path", W, H, 16, 2, CBOX_CAMERA, th), cbox_material)
    tiles = np.zeros_like(full)
    for (x0, y0, x1, y1) in [(0, 0, 16, 32), (16, 0, 32, 8), (16, 8, 32, 32)]:
        part = cbox_oracle.render_forward(oracle.make_params("path", W, H, 16, 2, CBOX_CAMERA, th, rect=(x0, y0, x1, y1)), cbox_material)
        tiles[y0:y1, x0:x1] = part[y0:y1, x0:x1]
    assert (tiles == full).all()                        # pixel tiles: bit-for-bit
    acc = np.zeros_like(full)
    for s0, s1 in [(0, 4), (4, 12), (12, 16)]:
        acc += cbox_oracle.render_forward(oracle.make_params("path", W, H, 16, 2, CBOX_CAMERA, th, samples=(s0, s1)), cbox_material)
    np.testing.assert_allclose(acc, full, rtol=1e-5, atol=1e-6)   # sample ranges: re-association only
    g_full = cbox_oracle.render_backward(oracle.make_params("path", W, H, 16, 3, CBOX_CAMERA, th), np.ones_like(full), cbox_material)
    g_acc = np.zeros_like(g_full)
    for (x0, y0, x1, y1) in [(0, 0, 32, 16), (0, 16, 32, 32)]:
        g_acc += cbox_oracle.render_backward(oracle.make_params("path", W, H, 16, 3, CBOX_CAMERA, th, rect=(x0, y0, x1, y1)), np.ones_like(full), cbox_material)
    np.testing.assert_allclose(g_acc, g_full, rtol=1e-5, atol=1e-7)


def test_cbox_path_statistics_match_survey(cbox_oracle, cbox_material):
    # ~2.35 closest rays, ~1.92 shadow rays and shaded vertices per sample
    p = oracle.make_params("path", 64, 64, 16, 0, CBOX_CAMERA, cbox_material.shape[:2])
    _, c = cbox_oracle.render_forward(p, cbox_material, counters=True)
    n = c["samples"]
    assert n == 64 * 64 * 16
    assert 2.2 < c["closest_rays"] / n < 2.5
    assert 1.8 < c["shaded_vertices"] / n < 2.05
    assert c["shadow_rays"] == c["shaded_vertices"]
    assert c["nan_samples"] == 0


def test_update_lights_switches_emitters(cbox_arrays, cbox_material):
    S = oracle.OracleScene.from_arrays(cbox_arrays)
    p = oracle.make_params("direct", 16, 16, 4, 0, CBOX_CAMERA, cbox_material.shape[:2])
    lit = S.render_forward(p, cbox_material)
    S.set_emissions(np.zeros((2, 3), np.float32))
    dark = S.render_forward(p, cbox_material)
    assert lit[..., :3].max() > 1 and dark[..., :3].max() == 0


def test_uvgrad_matches_float64_ray_differences():
    """render_duvdxy (uvgrad.py): on a planar quad with an affine uv map the Jacobian equals the uv
    difference between the hits of the rays through (x+1, y) / (x, y+1) and through (x, y); v is inverted."""
    A = quad_scene()
    S = oracle.OracleScene.from_arrays(A)
    cam = (0.6, (0.3, 2.0, 0.4), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    W = 12
    p = oracle.make_params("uvgrad", W, W, 1, 0, cam, (4, 4), use_tent=False)
    img = S.render_forward(p, np.zeros((4, 4, 4), np.float32))
    o = np.array(cam[1], np.float64)
    fwd = np.array(cam[2]) - o; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, cam[3]); right /= np.linalg.norm(right)
    upp = np.cross(right, fwd)

    def uv_at(fx, fy):
        px = (2.0 / W * fx - 1.0) * np.tan(0.3); py = (2.0 / W * fy - 1.0) * np.tan(0.3)
        d = px * right - py * upp + fwd; d /= np.linalg.norm(d)
        hit = o + (-o[1] / d[1]) * d
        return np.array([(hit[0] + 1) / 2, (hit[2] + 1) / 2]), hit     # quad_scene: u = (x+1)/2, v = (z+1)/2

    checked = 0
    for y in range(W):
        for x in range(W):
            off = oracle.sampler_dump(oracle.SAMPLER_CMJ, x, y, 0, 1, 0, nvert=0)
            fx, fy = x + off[0], y + off[1]
            uv, hit = uv_at(fx, fy)
            if abs(hit[0]) > 0.95 or abs(hit[2]) > 0.95:
                continue
            ux, _ = uv_at(fx + 1, fy); uy, _ = uv_at(fx, fy + 1)
            exp = [ux[0] - uv[0], -(ux[1] - uv[1]), uy[0] - uv[0], -(uy[1] - uv[1])]
            np.testing.assert_allclose(img[y, x], exp, rtol=2e-4, atol=2e-6)
            checked += 1
    assert checked > 20


def test_adjoint_follows_the_unclamped_russian_roulette(cbox_oracle):
    """prb.py:83 divides by q = max(lum(beta), 0.05) without clamping q to 1: for lum(beta) >= 1 the
    throughput is renormalised and the forward's expectation depends on q(material).  On a bright
    material (albedo 0.8-0.98: lum(beta) >= 1 is the rule) the adjoint that keeps q constant is ~40 % off
    finite differences; PRB_CORRECT follows them (deviation 8)."""
    rng = np.random.default_rng(0)
    m = np.empty((128, 128, 4), np.float32)
    m[..., :3] = rng.uniform(0.8, 0.98, (128, 128, 3)); m[..., 3] = rng.uniform(0.3, 0.9, (128, 128))
    W, spp = 16, 1024
    wimg = rng.uniform(0.5, 1.5, (W, W, 4)).astype(np.float32); wimg[..., 3] = 0
    delta = np.zeros_like(m); delta[..., :3] = rng.uniform(0, 1, m.shape[:2] + (3,))
    res = {}
    for mode in (oracle.PRB_CORRECT, oracle.PRB_DETACHED):
        ad = []
        for s in range(3):
            p = oracle.make_params("path", W, W, spp, 100 + s, CBOX_CAMERA, m.shape[:2], prb_mode=mode)
            ad.append((cbox_oracle.render_backward(p, wimg, m).astype(np.float64) * delta).sum())
        res[mode] = np.mean(ad)
    fd = []
    for s in range(6):
        p = oracle.make_params("path", W, W, spp, 300 + s, CBOX_CAMERA, m.shape[:2])
        ip = cbox_oracle.render_forward(p, (m + 0.01 * delta).astype(np.float32)).astype(np.float64)
        im = cbox_oracle.render_forward(p, (m - 0.01 * delta).astype(np.float32)).astype(np.float64)
        fd.append(((ip - im) * wimg).sum() / 0.02)
    fd_mean, fd_se = np.mean(fd), np.std(fd, ddof=1) / np.sqrt(len(fd))
    assert abs(res[oracle.PRB_CORRECT] - fd_mean) < max(4 * fd_se, 0.02 * abs(fd_mean)), (res, fd_mean, fd_se)
    assert abs(res[oracle.PRB_DETACHED] - fd_mean) > 0.2 * abs(fd_mean), (res, fd_mean)


def test_the_oracles_search_structure_changes_no_answer():
    """Scenes above 256 triangles are searched through the oracle's own binary BVH (oracle/zdr_oracle.c, build_bvh): the
    same plane-form test on fewer triangles.  Closest hits (instance, primitive, barycentrics, t — also on exact ties, which
    go to the smallest triangle index as in the loop), occlusion queries and a whole path-traced image with its gradient are
    BIT-IDENTICAL to the loop over every triangle."""
    from conftest import fd_material_np
    from gpu_util import terrain_arrays
    A = terrain_arrays(n=40)                                   # 3,202 triangles, with exactly shared edges (ties)
    S = oracle.OracleScene.from_arrays(A)
    rng = np.random.default_rng(3)
    n = 20000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform((-3, -0.5, -3), (3, 3.5, 3), (n, 3))
    d = rng.standard_normal((n, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = 1e30
    # rays aimed exactly at mesh vertices and edge midpoints: the ties
    V = A.verts[A.tris[:2000].reshape(-1), :3].reshape(-1, 3, 3)
    targets = np.concatenate([V[:, 0], 0.5 * (V[:, 0] + V[:, 1])])
    o = np.array([0.3, 3.9, 0.2], np.float32)
    aimed = np.zeros((targets.shape[0], 8), np.float32); aimed[:, 0:3] = o
    dd = targets - o; aimed[:, 4:7] = dd / np.linalg.norm(dd, axis=1, keepdims=True); aimed[:, 7] = 1e30
    rays = np.concatenate([rays, aimed])
    L = oracle.lib()
    try:
        ip, bt = S.trace_closest(rays)
        rays2 = rays.copy(); rays2[:, 3] = 1e-4; rays2[:, 7] = rng.uniform(0.05, 5.0, rays.shape[0])
        occ = S.trace_any(rays2)
        mat = fd_material_np(64, 5)
        p = oracle.make_params("path", 24, 24, 8, 2, (0.9, (0.5, 3.0, 6.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), mat.shape[:2])
        img = S.render_forward(p, mat); g = S.render_backward(p, np.ones((24, 24, 4), np.float32), mat)
        L.zdro_debug_force_brute(1)
        ip_b, bt_b = S.trace_closest(rays)
        assert np.array_equal(ip, ip_b) and np.array_equal(bt.view(np.uint32), bt_b.view(np.uint32))
        assert (ip[:, 0] >= 0).mean() > 0.3
        assert np.array_equal(occ, S.trace_any(rays2))
        assert np.array_equal(img.view(np.uint32), S.render_forward(p, mat).view(np.uint32))
        assert np.array_equal(g, S.render_backward(p, np.ones((24, 24, 4), np.float32), mat))
        assert img[..., :3].mean() > 0.01
    finally:
        L.zdro_debug_force_brute(0)
