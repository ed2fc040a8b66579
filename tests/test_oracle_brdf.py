"""Pins the oracle's microfacet model: float64 NumPy re-evaluation of
microfacet.py:7-58, analytic derivative vs float64 central differences, sampling/pdf consistency."""
import ctypes as C

import numpy as np

import oracle

PI = np.pi


def brdf64(wo, wi, d, r, spec=0.04):  # microfacet.py:7-30 in float64
    a2 = (r * r) ** 2
    h = (wi + wo) / np.linalg.norm(wi + wo)
    nh = max(1e-5, h[2])
    D = a2 / (PI * (nh * nh * (a2 - 1) + 1) ** 2)
    c = min(max(np.dot(wo, h), 1e-5), 1.0)
    F = spec + (1 - spec) * (1 - c) ** 5
    def G1(v):
        nv = max(1e-5, v[2])
        return 2 / (1 + np.sqrt(1 + a2 * (1 - nv * nv) / (nv * nv)))
    G = G1(wi) * G1(wo)
    return (D * F * G / (4 * max(1e-5, wi[2]) * max(1e-5, wo[2])) + d / PI) * wi[2]


def pdf64(wo, wi, r):  # microfacet.py:52-58,68-69
    a2 = (r * r) ** 2
    wm = (wi + wo) / np.linalg.norm(wi + wo)
    nv = max(1e-5, wo[2])
    G1 = 2 / (1 + np.sqrt(1 + a2 * (1 - nv * nv) / (nv * nv)))
    nh = max(1e-5, wm[2])
    D = a2 / (PI * (nh * nh * (a2 - 1) + 1) ** 2)
    glossy = G1 / abs(wo[2]) * D * abs(np.dot(wo, wm)) / (4 * abs(np.dot(wo, wm)))
    return 0.5 * wi[2] / PI + 0.5 * glossy


def rand_dir(rng, zmin=0.05):
    while True:
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        v[2] = abs(v[2])
        if v[2] > zmin:
            return v


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def c_brdf(wo, wi, d, r):
    out = np.zeros(3, np.float32)
    a, b, c = (np.ascontiguousarray(x, np.float32) for x in (wo, wi, d))
    oracle.lib().zdro_ggx_brdf(_fp(a), _fp(b), _fp(c), float(r), _fp(out))
    return out


def c_grad(wo, wi, d, r, g):
    out = np.zeros(4, np.float32)
    a, b, c, e = (np.ascontiguousarray(x, np.float32) for x in (wo, wi, d, g))
    oracle.lib().zdro_ggx_brdf_grad(_fp(a), _fp(b), _fp(c), float(r), _fp(e), _fp(out))
    return out


def test_brdf_and_pdf_match_float64():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        wo, wi = rand_dir(rng), rand_dir(rng)
        d, r = rng.uniform(0, 1, 3), rng.uniform(0.15, 1.0)  # below ~0.15 the float32 GGX denominator cancels badly
        wo32, wi32, d32, r32 = (np.float32(x).astype(np.float64) for x in (wo, wi, d, r))
        ref = brdf64(wo32, wi32, d32, float(r32))
        got = c_brdf(wo, wi, d, r)
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-6)
        a, b = np.ascontiguousarray(wo, np.float32), np.ascontiguousarray(wi, np.float32)
        p = oracle.lib().zdro_ggx_sample_pdf(_fp(a), _fp(b), float(np.float32(r)))
        np.testing.assert_allclose(p, pdf64(wo32, wi32, float(r32)), rtol=1e-3, atol=1e-6)


def test_brdf_gradient_matches_float64_central_differences():
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(2000):
        wo, wi = rand_dir(rng, 0.1), rand_dir(rng, 0.1)
        d, r, g = rng.uniform(0.05, 1, 3), rng.uniform(0.15, 1.0), rng.uniform(-1, 1, 3)
        wo32, wi32, d32, r32, g32 = (np.float32(x).astype(np.float64) for x in (wo, wi, d, r, g))
        eps = 1e-6
        fd = np.zeros(4)
        for c in range(3):
            e = np.zeros(3); e[c] = eps
            fd[c] = np.dot(g32, brdf64(wo32, wi32, d32 + e, float(r32)) - brdf64(wo32, wi32, d32 - e, float(r32))) / (2 * eps)
        fd[3] = np.dot(g32, brdf64(wo32, wi32, d32, float(r32) + eps) - brdf64(wo32, wi32, d32, float(r32) - eps)) / (2 * eps)
        got = c_grad(wo, wi, d, r, g)
        err = np.abs(got - fd) / (np.abs(fd) + 1e-3)
        worst = max(worst, err.max())
    assert worst < 2e-3, worst


def test_sampling_estimates_the_brdf_integral():
    """E[f/pdf] under ggx_sample == quadrature of f over the hemisphere (both lobes exercised)."""
    rng = np.random.default_rng(2)
    L = oracle.lib()
    for r in (0.2, 0.5, 1.0):
        wo = rand_dir(rng, 0.3)
        d = np.array([0.5, 0.3, 0.8])
        # quadrature in (cos theta, phi)
        n = 400
        ct = (np.arange(n) + 0.5) / n
        ph = (np.arange(2 * n) + 0.5) / (2 * n) * 2 * PI
        acc = 0.0
        for c in ct:
            st = np.sqrt(1 - c * c)
            for p in ph[::4]:
                acc += brdf64(wo, np.array([st * np.cos(p), st * np.sin(p), c]), d, r)[0]
        quad = acc * (1.0 / n) * (2 * PI / (2 * n / 4))
        # Monte Carlo with the oracle's sampler
        N = 40000
        u = rng.uniform(0, 1, (N, 3)).astype(np.float32)
        wo32 = np.ascontiguousarray(wo, np.float32)
        est = 0.0
        wi = np.zeros(3, np.float32)
        for k in range(N):
            u2 = np.ascontiguousarray(u[k, 1:3])
            L.zdro_ggx_sample(_fp(wo32), float(r), float(u[k, 0]), _fp(u2), _fp(wi))
            if wi[2] < 1e-4:
                continue
            p = L.zdro_ggx_sample_pdf(_fp(wo32), _fp(wi), float(r))
            est += c_brdf(wo32, wi, d, r)[0] / p
        est /= N
        assert abs(est - quad) / quad < 0.03, (r, est, quad)
