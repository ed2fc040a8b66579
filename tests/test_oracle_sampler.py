"""Pins the oracle's samplers: independent big-int Python restatements of
corrmj.py / pmj02bn.py arithmetic, bijectivity of the permutation, CMJ stratification."""
import numpy as np
import pytest

import oracle

M = 0xFFFFFFFF


def xxhash_py(x, y, z, w):  # pmj02bn.py:60-74 with Python big ints masked to 32 bits
    P2, P3, P4, P5 = 2246822519, 3266489917, 668265263, 374761393
    rot = lambda h: ((h << 17) | (h >> 15)) & M
    h = (w + P5 + x * P3) & M
    h = (P4 * rot(h)) & M
    h = (h + y * P3) & M
    h = (P4 * rot(h)) & M
    h = (h + z * P3) & M
    h = (P4 * rot(h)) & M
    h = (P2 * (h ^ (h >> 15))) & M
    h = (P3 * (h ^ (h >> 13))) & M
    return h ^ (h >> 16)


def perm_py(i, l, w, p):  # corrmj.py:6-28
    while True:
        i ^= p; i = (i * 0xe170893d) & M; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8
        i = (i * 0x0929eb3f) & M; i ^= p >> 23; i ^= (i & w) >> 1; i = (i * (1 | p >> 27)) & M
        i = (i * 0x6935fa69) & M; i ^= (i & w) >> 11; i = (i * 0x74dcb303) & M; i ^= (i & w) >> 2
        i = (i * 0x9e501cc3) & M; i ^= (i & w) >> 2; i = (i * 0xc860a3df) & M; i &= w; i ^= i >> 5
        if i < l:
            break
    return ((i + p) & M) % l


def smear(w):
    for s in (1, 2, 4, 8, 16):
        w |= w >> s
    return w


class CMJPy:  # corrmj.py:60-117
    def __init__(self, px, py, seed, spp, idx):
        f32 = np.float32
        self.idx, self.dim, self.spp, self.w = idx, 0, spp, smear(spp - 1)
        self.res = int(np.sqrt(f32(spp) + f32(0.4)))
        self.resw = smear(self.res - 1)
        self.ps = xxhash_py(px & M, py & M, seed & M, 0)
        self.state = xxhash_py(px & M, py & M, seed & M, idx)

    def lcg(self):
        self.state = (1664525 * self.state + 1013904223) & M
        return np.float32(self.state & 0xFFFFFF) * np.float32(1.0 / 0x1000000)

    def next(self):
        ps = (self.ps + self.dim) & M
        index = perm_py(self.idx, self.spp, self.w, ((ps * 0x45fbe943) & M) & 0x70ffffff)
        u = (np.float32(index) + self.lcg()) / np.float32(self.spp)
        self.dim += 1
        return min(max(u, np.float32(0)), np.float32(float.fromhex("0x1.fffffep-1")))

    def next2(self):
        ps = (self.ps + self.dim) & M
        index = perm_py(self.idx, self.spp, self.w, ((ps * 0x51633e2d) & M) & 0x70ffffff)
        y, x = index // self.res, index % self.res
        sx = perm_py(x, self.res, self.resw, ((ps * 0x68bc21eb) & M) & 0x70ffffff)
        sy = perm_py(y, self.res, self.resw, ((ps * 0x02e5be93) & M) & 0x70ffffff)
        dx, dy = self.lcg(), self.lcg()
        r = np.float32(self.res)
        one = np.float32(float.fromhex("0x1.fffffep-1"))
        ux = (np.float32(x) + (np.float32(sy) + dx) / r) / r
        uy = (np.float32(y) + (np.float32(sx) + dy) / r) / r
        self.dim += 2
        return min(max(ux, np.float32(0)), one), min(max(uy, np.float32(0)), one)


def test_xxhash_matches_bigint_python():
    rng = np.random.default_rng(1)
    L = oracle.lib()
    cases = [(0, 0, 0, 0), (M, M, M, M), (24, 345, 0, 0), (511, 511, 1, 255)]
    cases += [tuple(int(v) for v in rng.integers(0, 2**32, 4)) for _ in range(500)]
    for c in cases:
        assert L.zdro_xxhash32_4(*c) == xxhash_py(*c)


def test_xxhash_is_not_canonical_xxh32():
    # the reference's 4-word hash omits XXH32's length term: document it
    import struct
    import xxhash
    data = struct.pack("<4I", 1, 2, 3, 4)
    assert xxhash.xxh32(data, seed=0).intdigest() != xxhash_py(1, 2, 3, 4)


@pytest.mark.parametrize("l", [1, 2, 3, 7, 16, 22, 64, 100, 256, 1000, 1024])
def test_permutation_is_bijection_and_matches_python(l):
    L = oracle.lib()
    w = smear(l - 1)
    for p in (0, 1, 0x12345678 & 0x70ffffff, 0x70ffffff, 0xdeadbeef):
        out = [L.zdro_permutation_element(i, l, w, p) for i in range(l)]
        assert sorted(out) == list(range(l))
        assert out == [perm_py(i, l, w, p) for i in range(l)]


@pytest.mark.parametrize("spp", [1, 4, 16, 64, 256])
def test_cmj_sequence_bit_exact_vs_python(spp):
    for (px, py, seed) in [(0, 0, 0), (24, 345, 0), (511, 3, 12345), (7, 9, 853402567)]:
        for idx in sorted(set([0, 1, spp // 2, spp - 1])):
            got = oracle.sampler_dump(oracle.SAMPLER_CMJ, px, py, seed, spp, idx, nvert=4, rr_depth=2)
            s = CMJPy(px, py, seed, spp, idx)
            exp = list(s.next2())
            for k in range(4):
                exp += [s.next(), s.next(), *s.next2(), s.next(), *s.next2()]
                if k >= 2:
                    exp.append(s.next())
            exp = np.asarray(exp, np.float32)
            assert got.shape == exp.shape
            assert (got.view(np.uint32) == exp.view(np.uint32)).all()
            assert (got >= 0).all() and (got < 1).all()


@pytest.mark.parametrize("spp", [16, 64, 256, 1024])
def test_cmj_2d_points_are_one_per_stratum(spp):
    res = int(round(spp ** 0.5))
    pts = np.array([oracle.sampler_dump(oracle.SAMPLER_CMJ, 24, 345, 7, spp, i, nvert=0)[:2] for i in range(spp)])
    cell = (np.floor(pts[:, 0] * res).astype(int), np.floor(pts[:, 1] * res).astype(int))
    occ = np.zeros((res, res), int)
    np.add.at(occ, cell, 1)
    assert (occ == 1).all()                                   # jittered res x res grid
    # multi-jitter: each of the spp fine strata of either axis is hit exactly once
    assert sorted(np.floor(pts[:, 0] * spp).astype(int)) == list(range(spp))
    assert sorted(np.floor(pts[:, 1] * spp).astype(int)) == list(range(spp))


def test_cmj_1d_is_stratified():
    spp = 64
    for dim_skip in range(3):
        vals = []
        for i in range(spp):
            d = oracle.sampler_dump(oracle.SAMPLER_CMJ, 5, 6, 1, spp, i, nvert=1)
            vals.append(d[2 + dim_skip if dim_skip < 2 else 6])
        assert sorted(np.floor(np.array(vals) * spp).astype(int)) == list(range(spp))
