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

// ------------------------------------------------------------------ the seven CMJ numbers of a path vertex, two hashes per register
// A path vertex draws, in this order, next() (light pick), next() (triangle pick), next2() (point on the light), next() (lobe),
// next2() (direction): nine Kensler permutations — five of the sample index over [0, spp), four of a stratum coordinate over
// [0, res) — each 9 multiplies and ~20 logic operations on the VALU-bound kernels' critical resource.  The permutation only ever
// looks at the bits of its state below the mask's top bit (multiplication, xor and `(i & w) >> k` never carry information downwards
// past that), so for w <= 0xffff the state fits 16 bits and TWO permutations run in the halves of one register: v_pk_mul_lo_u16,
// v_pk_lshrrev_b16, v_pk_add_u16 and plain logic.  Same values bit for bit (tests/test_gpu_paths.py compares every vertex of
// every path with the oracle); five passes instead of nine — ten with the Russian-roulette draw, whose permutation shares the fifth.  Valid for the CMJ sampler with power-of-two spp and strata grid
// (every BASELINE configuration); anything else takes the calls one by one.
typedef unsigned short zdr_us2 __attribute__((ext_vector_type(2)));
ZD uint32_t pk_mul16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (zdr_us2)(__builtin_bit_cast(zdr_us2, a) * __builtin_bit_cast(zdr_us2, b))); }
ZD uint32_t pk_add16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (zdr_us2)(__builtin_bit_cast(zdr_us2, a) + __builtin_bit_cast(zdr_us2, b))); }
template <int K> ZD uint32_t pk_shr16(uint32_t a) { return __builtin_bit_cast(uint32_t, (zdr_us2)(__builtin_bit_cast(zdr_us2, a) >> (unsigned short)K)); }

// permutation_element(i.lo, w.lo + 1, w.lo, pa) in the low half, (i.hi, w.hi + 1, w.hi, pb) in the high half; l == w + 1 in both
ZD uint32_t permutation_element2(uint32_t i, uint32_t w, uint32_t pa, uint32_t pb, bool small_w) {
    const uint32_t P0 = __builtin_amdgcn_perm(pb, pa, 0x05040100u);      // {pa & 0xffff, pb & 0xffff}
    const uint32_t P16 = __builtin_amdgcn_perm(pb, pa, 0x07060302u);     // {pa >> 16, pb >> 16}
    const uint32_t P8 = __builtin_amdgcn_perm(pb, pa, 0x06050201u);      // {(pa >> 8) & 0xffff, (pb >> 8) & 0xffff}
    i ^= P0; i = pk_mul16(i, 0x893d893du); i ^= P16; i ^= pk_shr16<4>(i & w); i ^= P8;
    i = pk_mul16(i, 0xeb3feb3fu); i ^= pk_shr16<7>(P16); i ^= pk_shr16<1>(i & w); i = pk_mul16(i, pk_shr16<11>(P16) | 0x00010001u);
    if (small_w) i = pk_mul16(i, ((0x6935fa69u * 0x74dcb303u) & 0xffffu) * 0x00010001u);
    else { i = pk_mul16(i, 0xfa69fa69u); i ^= pk_shr16<11>(i & w); i = pk_mul16(i, 0xb303b303u); }
    i ^= pk_shr16<2>(i & w);
    i = pk_mul16(i, 0x1cc31cc3u); i ^= pk_shr16<2>(i & w); i = pk_mul16(i, 0xa3dfa3dfu); i &= w; i ^= pk_shr16<5>(i);
    return pk_add16(i, P0) & w;
}

struct VertexSamples { float u_pick, u_prim; f2 u_pt; float u_lobe; f2 u_dir; uint32_t i_rr; };   // i_rr: the permuted index of the NEXT 1-D draw (Russian roulette), should the vertex make it
__host__ ZD bool cmj_can_batch(const SamplerCfg &c) {               // wave-uniform (host: zdr_vertex_sampler_dump reports which route its kernel took)
    return c.spp_pow2 && c.res_pow2 && c.spp <= 65536u && c.w == c.spp - 1u && c.resw_x == c.res_x - 1u && c.resw_y == c.res_y - 1u;
}
ZD VertexSamples cmj_vertex_samples(const SamplerCfg &c, Sampler &s) {
    const uint32_t M = 0x70ffffffu;
    const uint32_t ps = s.permutation_seed + s.dimension;
    const uint32_t W = c.w | (c.w << 16), I = s.sample_index | (s.sample_index << 16);
    const bool small_w = c.w < 2048u;
    // the five permutations of the sample index (corrmj.py:95-102 and 105-108), dimensions +0 +1 +2 +4 +5
    const uint32_t h01 = permutation_element2(I, W, (ps * 0x45fbe943u) & M, ((ps + 1u) * 0x45fbe943u) & M, small_w);
    const uint32_t h23 = permutation_element2(I, W, ((ps + 2u) * 0x51633e2du) & M, ((ps + 4u) * 0x45fbe943u) & M, small_w);
    const uint32_t h45 = permutation_element2(I, W, ((ps + 5u) * 0x51633e2du) & M, ((ps + 7u) * 0x45fbe943u) & M, small_w);   // the fifth goes with the roulette's, which follows at dimension +7 if it is drawn at all
    const uint32_t i_pick = h01 & 0xffffu, i_prim = h01 >> 16, i_pt = h23 & 0xffffu, i_lobe = h23 >> 16, i_dir = h45 & 0xffffu;
    // the strata of the two 2-D draws (corrmj.py:109-112): x and y permuted in one pass each
    const uint32_t WR = c.resw_x | (c.resw_y << 16);
    const bool small_r = (c.resw_x | c.resw_y) < 2048u;
    const uint32_t x_pt = i_pt & (c.res_x - 1u), y_pt = i_pt >> c.res_x_shift, x_dir = i_dir & (c.res_x - 1u), y_dir = i_dir >> c.res_x_shift;
    const uint32_t s_pt = permutation_element2(x_pt | (y_pt << 16), WR, ((ps + 2u) * 0x68bc21ebu) & M, ((ps + 2u) * 0x02e5be93u) & M, small_r);
    const uint32_t s_dir = permutation_element2(x_dir | (y_dir << 16), WR, ((ps + 5u) * 0x68bc21ebu) & M, ((ps + 5u) * 0x02e5be93u) & M, small_r);
    // the jitters, in call order (corrmj.py:88-92)
    const float d_pick = next_lcg(s), d_prim = next_lcg(s), dx_pt = next_lcg(s), dy_pt = next_lcg(s), d_lobe = next_lcg(s), dx_dir = next_lcg(s), dy_dir = next_lcg(s);
    VertexSamples v;
    v.u_pick = strat(c, i_pick, d_pick); v.u_prim = strat(c, i_prim, d_prim); v.u_lobe = strat(c, i_lobe, d_lobe);
    {
        const float ax = (float)(s_pt >> 16) + dx_pt, ay = (float)(s_pt & 0xffffu) + dy_pt;       // (sy + dx, sx + dy)
        v.u_pt.x = clampf(((float)x_pt + ax * c.inv_res_y) * c.inv_res_x, 0.0f, ZDR_ONE_MINUS_EPS);
        v.u_pt.y = clampf(((float)y_pt + ay * c.inv_res_x) * c.inv_res_y, 0.0f, ZDR_ONE_MINUS_EPS);
    }
    {
        const float ax = (float)(s_dir >> 16) + dx_dir, ay = (float)(s_dir & 0xffffu) + dy_dir;
        v.u_dir.x = clampf(((float)x_dir + ax * c.inv_res_y) * c.inv_res_x, 0.0f, ZDR_ONE_MINUS_EPS);
        v.u_dir.y = clampf(((float)y_dir + ay * c.inv_res_x) * c.inv_res_y, 0.0f, ZDR_ONE_MINUS_EPS);
    }
    v.i_rr = h45 >> 16;
    s.dimension += 7;
    return v;
}
// The pixel's 2-D draw (integrator.py:19, corrmj.py:105-117) with the two stratum permutations in the halves of one register: the index permutation in its
// scalar form, then sx and sy in one packed pass — two passes instead of three.  Same values bit for bit; valid under cmj_can_batch.
ZD f2 cmj_next2_packed(const SamplerCfg &c, Sampler &s) {
    const uint32_t M = 0x70ffffffu;
    const uint32_t ps = s.permutation_seed + s.dimension;
    const uint32_t index = permutation_element(s.sample_index, c.spp, c.w, (ps * 0x51633e2du) & M);
    const uint32_t x = index & (c.res_x - 1u), y = index >> c.res_x_shift;
    const uint32_t WR = c.resw_x | (c.resw_y << 16);
    const bool small_r = (c.resw_x | c.resw_y) < 2048u;
    const uint32_t sxy = permutation_element2(x | (y << 16), WR, (ps * 0x68bc21ebu) & M, (ps * 0x02e5be93u) & M, small_r);
    const float dx = next_lcg(s), dy = next_lcg(s);
    const float ax = (float)(sxy >> 16) + dx, ay = (float)(sxy & 0xffffu) + dy;          // (sy + dx, sx + dy)
    f2 u;
    u.x = clampf(((float)x + ax * c.inv_res_y) * c.inv_res_x, 0.0f, ZDR_ONE_MINUS_EPS);
    u.y = clampf(((float)y + ay * c.inv_res_x) * c.inv_res_y, 0.0f, ZDR_ONE_MINUS_EPS);
    s.dimension += 2;
    return u;
}
// the first draw of every camera sample, by the route the kernels take (pixel_ray, and zdr_vertex_sampler_dump beside it)
template <int KIND>
ZD f2 sampler_pixel_offset(const SamplerCfg &c, Sampler &s) {
    if (KIND == 0 && cmj_can_batch(c)) return cmj_next2_packed(c, s);       // wave-uniform
    return sampler_next2<KIND>(c, s);
}

// sampler_next<cmj> with the permutation already done (cmj_vertex_samples' i_rr): corrmj.py:95-102
ZD float cmj_next_with_index(const SamplerCfg &c, Sampler &s, uint32_t index) {
    const float delta = next_lcg(s);
    s.dimension += 1;
    return strat(c, index, delta);
}
