// sampler.h — the two samplers of the reference on uint32:
// corrmj.py:6-117 (correlated multi-jitter) and pmj02bn.py:20-126 (PMJ02 + blue noise tables).
// Integer work is bit-exact by construction; float conversions use exact reciprocal multiplies
// when the divisor is a power of two and IEEE division otherwise, so the drawn values are
// bit-identical to the oracle's (tests/test_gpu_sampler.py).
#pragma once
#include "vecmath.h"

struct SamplerTables {          // pmj02bn.py:9-18; null when no tables were supplied
    const uint32_t *pmj;        // [nsets][nsamples][2], value / 2^32
    const uint16_t *bn;         // [ntex][res][res], value / 2^16
    uint32_t nsets, nsamples, ntex, bnres;
};

ZD uint32_t xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {  // pmj02bn.py:60-74
    const uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    uint32_t h = w + P5 + x * P3;
    h = P4 * __builtin_rotateleft32(h, 17);
    h += y * P3;
    h = P4 * __builtin_rotateleft32(h, 17);
    h += z * P3;
    h = P4 * __builtin_rotateleft32(h, 17);
    h = P2 * (h ^ (h >> 15));
    h = P3 * (h ^ (h >> 13));
    return h ^ (h >> 16);
}

ZD uint32_t smear_mask(uint32_t w) { w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16; return w; }

// Kensler's permute (corrmj.py:6-28). When l is a power of two (l == w + 1) the cycle walk never
// repeats and the final modulo is a mask — true for every BASELINE spp.
// The reference's `while True` never ends for a start value >= l whose cycle stays outside [0, l);
// a valid walk rejects at most w + 1 - l values, so the loop is bounded by exactly that: every wave
// is guaranteed to leave it.
// w is wave-uniform (a launch constant).  Below 2048 — every spp up to 2048, every strata grid — `(i & w) >> 11` is zero, the xor
// falls away and the two multiplies around it are one (mod 2^32); same values, 4 instructions of 31 fewer on the dependent chain.
ZD uint32_t kensler_hash(uint32_t i, uint32_t w, uint32_t p) {
    i ^= p; i *= 0xe170893du; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8;
    i *= 0x0929eb3fu; i ^= p >> 23; i ^= (i & w) >> 1; i *= 1u | p >> 27;
    if (w < 2048u) i *= 0x6935fa69u * 0x74dcb303u;
    else { i *= 0x6935fa69u; i ^= (i & w) >> 11; i *= 0x74dcb303u; }
    i ^= (i & w) >> 2;
    i *= 0x9e501cc3u; i ^= (i & w) >> 2; i *= 0xc860a3dfu; i &= w; i ^= i >> 5;
    return i;
}
ZD uint32_t permutation_element(uint32_t i, uint32_t l, uint32_t w, uint32_t p) {
    if (l == w + 1u) return (kensler_hash(i, w, p) + p) & w;     // wave-uniform: straight-line code, no cycle walk
    uint32_t budget = w - l + 2u;
    do { i = kensler_hash(i, w, p); } while (i >= l && --budget);
    return (i + p) % l;
}

// Wave-uniform sampler configuration, computed once per launch on the host.
struct SamplerCfg {
    int32_t kind;               // ZDR_SAMPLER_*
    uint32_t seed, spp, w;      // w = smear(spp - 1)
    // CMJ 2-D strata grid: res_x = res_y = int(sqrt(spp + 0.4)) when spp is a perfect square
    // (corrmj.py:67); otherwise a res_x x res_y >= spp grid (the reference's formula leaves the
    // permutation's domain there, see oracle/zdr_oracle.c zdro_cmj_grid)
    uint32_t res_x, res_y, resw_x, resw_y;
    float inv_spp, inv_res_x, inv_res_y;   // exact when the divisors are powers of two, else unused
    int32_t spp_pow2, res_pow2, res_x_shift;
    SamplerTables tab;
};

struct Sampler {                // per-lane state (pmj02bn.py:78-85, corrmj.py:48-57)
    uint32_t px, py, sample_index, dimension;
    uint32_t permutation_seed, state;
};

#define ZDR_ONE_MINUS_EPS 0x1.fffffep-1f

template <int KIND>
ZD Sampler sampler_make(const SamplerCfg &c, uint32_t px, uint32_t py, uint32_t pixel_perm_seed, uint32_t sample_index) {
    Sampler s;
    s.px = px; s.py = py; s.sample_index = sample_index; s.dimension = 0;
    s.permutation_seed = pixel_perm_seed;                           // corrmj.py:78 (per pixel)
    s.state = (KIND == 0) ? xxhash32_4(px, py, c.seed, sample_index) : 0u;  // corrmj.py:79
    return s;
}

ZD float next_lcg(Sampler &s) {                                     // corrmj.py:88-92
    s.state = 1664525u * s.state + 1013904223u;
    return (float)(s.state & 0x00ffffffu) * (1.0f / 16777216.0f);
}

ZD float strat(const SamplerCfg &c, uint32_t index, float delta) {  // (index + delta) / spp, clamped
    float a = (float)index + delta;
    float u = c.spp_pow2 ? a * c.inv_spp : __fdiv_rn(a, (float)c.spp);
    return clampf(u, 0.0f, ZDR_ONE_MINUS_EPS);
}

ZD float blue_noise(const SamplerTables &t, uint32_t tex, uint32_t x, uint32_t y) {  // pmj02bn.py:20-24, pbrt layout (App. B-5)
    uint32_t ti = tex % t.ntex, cx = x % t.bnres, cy = y % t.bnres;
    return (float)t.bn[((size_t)ti * t.bnres + cx) * t.bnres + cy] * (1.0f / 65536.0f);
}

template <int KIND>
ZD float sampler_next(const SamplerCfg &c, Sampler &s) {
    if (KIND == 0) {                                                // corrmj.py:95-102
        uint32_t ps = s.permutation_seed + s.dimension;
        uint32_t index = permutation_element(s.sample_index, c.spp, c.w, (ps * 0x45fbe943u) & 0x70ffffffu);
        float delta = next_lcg(s);
        s.dimension += 1;
        return strat(c, index, delta);
    } else {                                                        // pmj02bn.py:105-112
        uint32_t h = xxhash32_4(s.px, s.py, s.dimension, c.seed);
        uint32_t index = permutation_element(s.sample_index, c.spp, c.w, h);
        float delta = blue_noise(c.tab, s.dimension, s.px ^ c.seed, s.py ^ c.seed);
        s.dimension += 1;
        return strat(c, index, delta);
    }
}

template <int KIND>
ZD f2 sampler_next2(const SamplerCfg &c, Sampler &s) {
    f2 u;
    if (KIND == 0) {                                                // corrmj.py:105-117
        uint32_t ps = s.permutation_seed + s.dimension;
        uint32_t index = permutation_element(s.sample_index, c.spp, c.w, (ps * 0x51633e2du) & 0x70ffffffu);
        uint32_t y, x;
        if (c.res_pow2) { y = index >> c.res_x_shift; x = index & (c.res_x - 1u); }
        else { y = index / c.res_x; x = index % c.res_x; }
        uint32_t sx = permutation_element(x, c.res_x, c.resw_x, (ps * 0x68bc21ebu) & 0x70ffffffu);
        uint32_t sy = permutation_element(y, c.res_y, c.resw_y, (ps * 0x02e5be93u) & 0x70ffffffu);
        float dx = next_lcg(s), dy = next_lcg(s);
        float ax = (float)sy + dx, ay = (float)sx + dy;
        if (c.res_pow2) {
            u.x = ((float)x + ax * c.inv_res_y) * c.inv_res_x;
            u.y = ((float)y + ay * c.inv_res_x) * c.inv_res_y;
        } else {
            float frx = (float)c.res_x, fry = (float)c.res_y;
            u.x = __fdiv_rn((float)x + __fdiv_rn(ax, fry), frx);
            u.y = __fdiv_rn((float)y + __fdiv_rn(ay, frx), fry);
        }
        s.dimension += 2;
        u.x = clampf(u.x, 0.0f, ZDR_ONE_MINUS_EPS); u.y = clampf(u.y, 0.0f, ZDR_ONE_MINUS_EPS);
    } else {                                                        // pmj02bn.py:115-126
        uint32_t index = s.sample_index;
        uint32_t inst = s.dimension / 2u;
        if (inst >= c.tab.nsets) {
            uint32_t h = xxhash32_4(s.px, s.py, s.dimension, c.seed);
            index = permutation_element(s.sample_index, c.spp, c.w, h);
        }
        size_t i = (size_t)(inst % c.tab.nsets) * c.tab.nsamples + index;
        // table / 2**32 in float64 then rounded to float32 (pmj02bn.py:9): uint32 -> double is exact
        float tx = (float)((double)c.tab.pmj[2 * i] * (1.0 / 4294967296.0));
        float ty = (float)((double)c.tab.pmj[2 * i + 1] * (1.0 / 4294967296.0));
        float ux = tx + blue_noise(c.tab, s.dimension, s.px ^ c.seed, s.py ^ c.seed);
        float uy = ty + blue_noise(c.tab, s.dimension + 1u, s.px ^ c.seed, s.py ^ c.seed);
        s.dimension += 2;
        u.x = ux - floorf(ux); u.y = uy - floorf(uy);
    }
    return u;
}
