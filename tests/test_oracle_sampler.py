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


def perm_py(i, l, w, p):  # corrmj.py:6-28, with the walk bounded as in the oracle (see zdro_permutation_element)
    budget = w - l + 2
    while True:
        i ^= p; i = (i * 0xe170893d) & M; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8
        i = (i * 0x0929eb3f) & M; i ^= p >> 23; i ^= (i & w) >> 1; i = (i * (1 | p >> 27)) & M
        i = (i * 0x6935fa69) & M; i ^= (i & w) >> 11; i = (i * 0x74dcb303) & M; i ^= (i & w) >> 2
        i = (i * 0x9e501cc3) & M; i ^= (i & w) >> 2; i = (i * 0xc860a3df) & M; i &= w; i ^= i >> 5
        budget -= 1
        if i < l or budget == 0:
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
        res = max(1, int(np.sqrt(f32(spp) + f32(0.4))))
        if res * res == spp:                     # the reference's domain (corrmj.py:67)
            self.resx = self.resy = res
        elif spp & (spp - 1) == 0:               # generalised grid, see zdro_cmj_grid
            lg = spp.bit_length() - 1
            self.resx = 1 << ((lg + 1) // 2); self.resy = spp // self.resx
        else:
            m = res
            while m * m < spp:
                m += 1
            self.resx, self.resy = m, (spp + m - 1) // m
        self.reswx, self.reswy = smear(self.resx - 1), smear(self.resy - 1)
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
        y, x = index // self.resx, index % self.resx
        sx = perm_py(x, self.resx, self.reswx, ((ps * 0x68bc21eb) & M) & 0x70ffffff)
        sy = perm_py(y, self.resy, self.reswy, ((ps * 0x02e5be93) & M) & 0x70ffffff)
        dx, dy = self.lcg(), self.lcg()
        rx, ry = np.float32(self.resx), np.float32(self.resy)
        one = np.float32(float.fromhex("0x1.fffffep-1"))
        ux = (np.float32(x) + (np.float32(sy) + dx) / ry) / rx
        uy = (np.float32(y) + (np.float32(sx) + dy) / rx) / ry
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


def test_permutation_walk_is_bounded_for_out_of_domain_starts():
    # the reference's `while True` spins forever on these (found by search); the bounded walk returns
    L = oracle.lib()
    for (i, l, w, p) in [(7, 7, 7, 0x30a9915b), (11, 10, 15, 0x7e7d7f), (50, 50, 63, 0x60093006), (101, 100, 127, 0x604e2746)]:
        assert L.zdro_permutation_element(i, l, w, p) < l
        assert L.zdro_permutation_element(i, l, w, p) == perm_py(i, l, w, p)


@pytest.mark.parametrize("spp", [1, 4, 16, 64, 256, 2, 8, 32, 128, 50, 12])
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


@pytest.mark.parametrize("spp", [2, 8, 32, 128, 512])
def test_cmj_2d_non_square_power_of_two_is_stratified(spp):
    # generalised 2^(k+1) x 2^k grid: one point per cell, each fine stratum of either axis hit once
    import ctypes as C
    rx, ry = C.c_uint32(), C.c_uint32()
    oracle.lib().zdro_cmj_grid(spp, C.byref(rx), C.byref(ry))
    rx, ry = rx.value, ry.value
    assert rx * ry == spp and rx in (ry, 2 * ry)
    pts = np.array([oracle.sampler_dump(oracle.SAMPLER_CMJ, 3, 4, 7, spp, i, nvert=0)[:2] for i in range(spp)])
    assert (pts >= 0).all() and (pts < 1).all()
    occ = np.zeros((rx, ry), int)
    np.add.at(occ, (np.floor(pts[:, 0] * rx).astype(int), np.floor(pts[:, 1] * ry).astype(int)), 1)
    assert (occ == 1).all()
    assert sorted(np.floor(pts[:, 0] * spp).astype(int)) == list(range(spp))
    assert sorted(np.floor(pts[:, 1] * spp).astype(int)) == list(range(spp))


def test_cmj_any_spp_stays_in_unit_square_and_terminates():
    for spp in (3, 5, 12, 50, 99, 1000):
        pts = np.array([oracle.sampler_dump(oracle.SAMPLER_CMJ, 9, 1, 2, spp, i, nvert=1) for i in range(spp)])
        assert (pts >= 0).all() and (pts < 1).all()


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
