"""Tables for the PMJ02bn sampler (/root/reference/pmj02bn.py:9-18) — SURVEY §8f-4.

The reference imports pbrt-v4's ``PMJ02bnSamples`` (5 sets x 65536 samples) and ``BlueNoiseTextures``
(48 textures of 128 x 128) from two modules that are not shipped (.MISSING_LARGE_BLOBS).  These
generators produce tables with the same shapes, types and structural properties:

* sample sets: Owen-scrambled Sobol' (0,2)-sequences.  Every power-of-two prefix of such a sequence is
  a (0,2)-net — each elementary interval of area 1/n holds exactly one point — which is the defining
  property of pmj02 (Christensen et al. 2018); pbrt's tables are additionally optimised for blue-noise
  point spacing, these are not.
* blue-noise textures: void-and-cluster (Ulichney 1993), toroidal Gaussian energy, ranks scaled to uint16.

The values differ from pbrt's, so images differ from the reference's PMJ02bn renders sample for sample;
the correlated-multi-jitter sampler (corrmj.py) is the one that is pinned bit-exactly.
"""
from __future__ import annotations

import os

import numpy as np

N_SETS, N_SAMPLES, N_TEXTURES, BN_RES = 5, 65536, 48, 128
_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache_pmj02bn_tables.npz")


def _reverse_bits32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x = ((x >> 1) & np.uint32(0x55555555)) | ((x & np.uint32(0x55555555)) << 1)
    x = ((x >> 2) & np.uint32(0x33333333)) | ((x & np.uint32(0x33333333)) << 2)
    x = ((x >> 4) & np.uint32(0x0F0F0F0F)) | ((x & np.uint32(0x0F0F0F0F)) << 4)
    x = ((x >> 8) & np.uint32(0x00FF00FF)) | ((x & np.uint32(0x00FF00FF)) << 8)
    return (x >> 16) | (x << 16)


def _laine_karras(x: np.ndarray, seed: int) -> np.ndarray:
    """hash whose output bit k depends only on input bits <= k: applied to bit-reversed values it is a
    nested uniform (Owen) scramble (Laine & Karras 2011; Burley 2020)."""
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x = x + np.uint32(seed)
        x ^= x * np.uint32(0x6c50b47c)
        x ^= x * np.uint32(0xb82f1e52)
        x ^= x * np.uint32(0xc7afe638)
        x ^= x * np.uint32(0x8d22f6e6)
    return x


def _owen(x: np.ndarray, seed: int) -> np.ndarray:
    return _reverse_bits32(_laine_karras(_reverse_bits32(x), seed))


def sobol02(n: int) -> np.ndarray:
    """first n points of the 2-D Sobol' (0,2)-sequence as uint32 fixed point (value / 2^32)."""
    i = np.arange(n, dtype=np.uint32)
    x = _reverse_bits32(i)                              # van der Corput
    y = np.zeros(n, np.uint32)
    v = np.uint32(1 << 31)
    k = i.copy()
    while k.any():
        y ^= np.where(k & np.uint32(1), v, np.uint32(0)).astype(np.uint32)
        k >>= np.uint32(1)
        v ^= v >> np.uint32(1)
    return np.stack([x, y], axis=1)


def pmj02_sets(n_sets: int = N_SETS, n_samples: int = N_SAMPLES, seed: int = 0) -> np.ndarray:
    base = sobol02(n_samples)
    rng = np.random.default_rng(seed)
    out = np.empty((n_sets, n_samples, 2), np.uint32)
    for s in range(n_sets):
        sx, sy = (int(v) for v in rng.integers(1, 2 ** 32, 2, dtype=np.uint64))
        out[s, :, 0] = _owen(base[:, 0], sx)
        out[s, :, 1] = _owen(base[:, 1], sy)
    return out


def void_and_cluster(res: int, seed: int, sigma: float = 1.9) -> np.ndarray:
    """rank matrix (0 .. res*res-1) of a toroidal void-and-cluster dither array."""
    rng = np.random.default_rng(seed)
    n = res * res
    d = np.minimum(np.arange(res), res - np.arange(res)).astype(np.float64)
    kernel = np.exp(-(d[:, None] ** 2 + d[None, :] ** 2) / (2 * sigma * sigma))

    def add(energy, idx, sign):
        y, x = divmod(int(idx), res)
        energy += sign * np.roll(np.roll(kernel, y, axis=0), x, axis=1)

    ones = max(4, n // 10)
    pattern = np.zeros(n, bool)
    pattern[rng.choice(n, ones, replace=False)] = True
    energy = np.zeros((res, res))
    for idx in np.flatnonzero(pattern):
        add(energy, idx, +1.0)
    flat = energy.reshape(-1)
    while True:                                         # phase 0: relax the initial pattern
        cluster = int(np.argmax(np.where(pattern, flat, -np.inf)))
        pattern[cluster] = False; add(energy, cluster, -1.0)
        void = int(np.argmin(np.where(pattern, np.inf, flat)))
        pattern[void] = True; add(energy, void, +1.0)
        if void == cluster:
            break
    rank = np.zeros(n, np.int64)
    p1, e1 = pattern.copy(), energy.copy()
    f1 = e1.reshape(-1)
    for r in range(ones - 1, -1, -1):                   # phase 1: remove tightest clusters
        c = int(np.argmax(np.where(p1, f1, -np.inf)))
        p1[c] = False; add(e1, c, -1.0); rank[c] = r
    for r in range(ones, n):                            # phases 2+3: fill largest voids
        v = int(np.argmin(np.where(pattern, np.inf, flat)))
        pattern[v] = True; add(energy, v, +1.0); rank[v] = r
    return rank.reshape(res, res)


def blue_noise_textures(n_tex: int = N_TEXTURES, res: int = BN_RES, seed: int = 0) -> np.ndarray:
    out = np.empty((n_tex, res, res), np.uint16)
    n = res * res
    for t in range(n_tex):
        rank = void_and_cluster(res, seed * 1000 + t)
        out[t] = ((rank.astype(np.float64) + 0.5) / n * 65536.0).astype(np.uint16)
    return out


def default_tables(verbose: bool = False):
    """(pmj [5][65536][2] uint32, bn [48][128][128] uint16), generated once (about two minutes) and cached."""
    if os.path.exists(_CACHE):
        z = np.load(_CACHE)
        return z["pmj"], z["bn"]
    if verbose:
        print("generating PMJ02bn tables (one-off, ~2 min) ...", flush=True)
    pmj, bn = pmj02_sets(), blue_noise_textures()
    try:
        np.savez(_CACHE, pmj=pmj, bn=bn)
    except OSError:
        pass
    return pmj, bn
