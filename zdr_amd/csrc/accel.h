// accel.h — closest-hit / any-hit queries (LuisaCompute Accel.trace_closest / trace_any).
//  * BruteAccel: for scenes of a few dozen triangles (cbox: 32).  The slot loop is wave-uniform,
//    so triangle data arrives through scalar loads into SGPRs and costs no VGPRs or LDS.
//  * BvhAccel: BVH4 with 64-byte quantised nodes, a per-lane traversal stack (first entries in LDS laid out
//    [entry][lane], one bank per lane; the rest in scratch) and 48-byte triangles fetched with dwordx4 loads.
// Both run the same two-sided plane-form triangle test accepting tmin < t < tmax, so they return
// the same hit (up to exact ties in t).
#pragma once
#include "scene.h"

// __launch_bounds__ second argument of the path kernels = minimum waves per SIMD (caps the VGPR budget at 512 / n)
#ifndef ZDR_MIN_WAVES
#define ZDR_MIN_WAVES 4        // cbox 512^2 spp 256 forward with 3 / 4 / 5 waves per SIMD: 9.8 / 9.1 / 10.3 ms (127 VGPRs and no spills at 4; profiles/r2_fwd_occupancy.txt)
#endif
#ifndef ZDR_MIN_WAVES_ENV
#define ZDR_MIN_WAVES_ENV 4    // the environment-light instantiation: 10 registers spilled at 4, still 11.9 -> 11.4 ms
#endif
#ifndef ZDR_MIN_WAVES_BVH
#define ZDR_MIN_WAVES_BVH 5    // 1 M triangles, forward ms at 1024^2 spp 32 with 4 / 5 / 6 waves per SIMD since the walk steals subtrees (walk_steal: more live state in the loop): 21.7 / 20.6 / 21.0 (round 2, 4 / 5 / 6 / 7 / 8 waves: 32.8 / 31.4 / 30.3 / 31.4 / 39.4)
#endif
// The backward path kernel's waves per CU are decided by LDS, which gfx950 hands out in 128 blocks of 1,280 bytes per CU:
// 16 waves = 8 blocks = 10,240 bytes per wave.  The kernel keeps neither the CMJ seeds nor the pixel cotangents of its two item
// banks in LDS (a popped path hashes its seed again and loads its cotangent from the image), which leaves room for a 103-slot
// record pool in 8 blocks — with 128 VGPRs (no spill since the kernarg reload) 16 waves per CU instead of 12: 12.84 -> 11.37 ms on
// cbox 512^2 spp 256 (profiles/r3_bwd_records_and_atomics.txt).
#ifndef ZDR_MIN_WAVES_BWD
#define ZDR_MIN_WAVES_BWD 4
#endif
#ifndef ZDR_POOL_SLOTS
#define ZDR_POOL_SLOTS 103            // brute force: 10,176 bytes of LDS per wave (103 x 81 + the scatter queue + 48)
#endif
#ifndef ZDR_POOL_SLOTS_BVH
#define ZDR_POOL_SLOTS_BVH 56         // BVH: + the traversal stack (ZDR_BVH_LDS_STACK_BWD entries x 256 bytes) + the 1,280 bytes of walk_steal, 8 blocks as well (72 slots before stealing; 62 slots + 8 stack entries and 48 + 12 measured 1 % slower)
#endif
#ifndef ZDR_MIN_WAVES_BWD_BVH
#define ZDR_MIN_WAVES_BWD_BVH 4
#endif


// The triangle array is read-only for the whole launch.  Reading it through the CONSTANT address
// space makes every wave-uniform fetch a scalar load (s_load_dwordx4 into SGPRs) even in kernels
// that also issue atomics or scratch stores — without it the backward kernel fell back to per-lane
// global loads (6.5e8 VMEM reads per launch, 59 % of wave time in s_waitcnt).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(4))) *const_v4f_ptr;
ZD const_v4f_ptr as_constant(const float4 *p) { return (const_v4f_ptr)(uintptr_t)p; }
ZD float4 f4(v4f a) { return make_float4(a.x, a.y, a.z, a.w); }

// Plane-form ray/triangle test (Havel & Herout style).  isect[3 slot + {0,1,2}] = N {n, n.p0},
// U {nu, du}, V {nv, dv}, precomputed in float64 on the host (zdr_api.cpp):
//   t = (N.w - n.o) / (n.d),  p = o + t d,  u = nu.p + du,  v = nv.p + dv,  hit: tmin < t < tmax, u, v >= 0, u + v <= 1
// 24 VALU per triangle against 46 for Moeller-Trumbore with precomputed edges; the closest-hit loops
// track only (t, slot) and evaluate the barycentrics of the winner once, after the loop.
// The operation order (explicit fmaf) is the one of the packed pair test below, so that the BVH
// leaves and the brute-force loop decide every ray identically.
ZD bool tri_test(float4 N, float4 U, float4 V, f3 o, f3 d, float tmin, float tmax, float &t) {
    float nd = fmaf(N.z, d.z, fmaf(N.y, d.y, N.x * d.x));
    float no = fmaf(N.z, o.z, fmaf(N.y, o.y, N.x * o.x));
    float tt = (N.w - no) * rcp(nd);
    float px = fmaf(d.x, tt, o.x), py = fmaf(d.y, tt, o.y), pz = fmaf(d.z, tt, o.z);
    float uu = fmaf(U.z, pz, fmaf(U.y, py, U.x * px)) + U.w;
    float vv = fmaf(V.z, pz, fmaf(V.y, py, V.x * px)) + V.w;
    float c = fminf(fminf(uu, vv), 1.0f - (uu + vv));
    t = tt;
    return (tt > tmin) & (tt < tmax) & (c >= 0.0f);
}

ZD void hit_barycentrics(const DScene &S, Hit &h, f3 o, f3 d) {
    if (h.slot < 0) return;
    float4 U = S.isect[3 * (size_t)h.slot + 1], V = S.isect[3 * (size_t)h.slot + 2];
    f3 p = o + d * h.t;
    h.u = U.x * p.x + U.y * p.y + U.z * p.z + U.w;
    h.v = V.x * p.x + V.y * p.y + V.z * p.z + V.w;
}

// ---- packed pair test -------------------------------------------------------------------------
// On gfx950 an fp32 VALU instruction occupies its SIMD for 4 cycles per wave64 whether it is v_fma_f32
// or v_pk_fma_f32 (profiles/r1_valu_issue_rate.txt): the packed forms do two floats per lane for the
// price of one.  The brute-force loops therefore test TWO primitives per trip, one in each half of a
// float2.  A primitive is a planar convex QUAD — two triangles that the host found to share an edge and a
// plane (zdr_api.cpp, find_quads): slots 2q and 2q + 1 — or a single triangle: one plane N, one hit point,
// and four edge functions e = n.p + d that are all >= 0 inside: u and v of both triangles of a quad (their
// corners are ordered so that the shared diagonal is the w = 0 edge of both), or u, v, w, w of a single
// triangle.  39 VALU per pair of primitives in the any-hit walk, i.e. for up to FOUR triangles, against 31
// per pair of triangles; the Cornell box is 15 quads and 2 triangles: 9 trips instead of 16.  S.pairs holds per
// pair of primitives (2k, 2k+1) ten float4
//   {Nx Nx' Ny Ny'} {Nz Nz' Nw Nw'}  and for each of the four edge functions  {Ex Ex' Ey Ey'} {Ez Ez' Ew Ew'}.
// They are wave-uniform and arrive as SGPR pairs, the ray is broadcast to both halves through op_sel.  An odd
// primitive count is padded with an all-zero record (0 * inf = NaN fails every comparison).  Which triangle of
// a quad was hit is decided afterwards, for the winner only (brute_resolve): the first one unless the point
// lies beyond its diagonal.  Hits on the outer edges and on the plane are the triangles' own arithmetic (same
// records, same fmaf order as tri_test); a ray through the quad's second triangle uses the plane of the first,
// which the host accepted as the same plane to 5e-7 of the quad's size (float32 rounding of the corners).
typedef float v2f __attribute__((ext_vector_type(2)));
ZD v2f splat(float a) { v2f r = {a, a}; return r; }
ZD v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

struct PairHit { v2f t, c; };   // c >= 0 <=> inside all four edges; t = ray parameter
ZD PairHit pair_test(const_v4f_ptr q, f3 o, f3 d) {
    v4f q0 = q[0], q1 = q[1];
    v2f dx = splat(d.x), dy = splat(d.y), dz = splat(d.z), ox = splat(o.x), oy = splat(o.y), oz = splat(o.z);
    v2f nd = pfma(q1.xy, dz, pfma(q0.zw, dy, q0.xy * dx));
    v2f no = pfma(q1.xy, oz, pfma(q0.zw, oy, q0.xy * ox));
    v2f tn = q1.zw - no;
    v2f r = {rcp(nd.x), rcp(nd.y)};
    PairHit h;
    h.t = tn * r;
    v2f px = pfma(dx, h.t, ox), py = pfma(dy, h.t, oy), pz = pfma(dz, h.t, oz);
    v2f e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v4f a = q[2 + 2 * k], b = q[3 + 2 * k];
        e[k] = pfma(b.xy, pz, pfma(a.zw, py, a.xy * px)) + b.zw;
    }
    h.c.x = fminf(fminf(fminf(e[0].x, e[1].x), e[2].x), e[3].x);   // v_min3_f32 + v_min_f32; a NaN here implies a NaN or infinite t, which the range test rejects
    h.c.y = fminf(fminf(fminf(e[0].y, e[1].y), e[2].y), e[3].y);
    return h;
}

// Two PARALLELOGRAMS (the host puts their pairs first, S.nppairs of them): with p = a0 + u e1 + v e2 the outer edges of the
// second triangle are u <= 1 and v <= 1, so its two edge functions are 1 - u and 1 - v of the first — six float4 per pair
// {N} {U} {V} interleaved (S.ppairs), 33 VALU instead of 39 (11 of the Cornell box's 15 quads are parallelograms).
ZD PairHit pair_test_par(const_v4f_ptr q, f3 o, f3 d) {
    v4f q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
    v2f dx = splat(d.x), dy = splat(d.y), dz = splat(d.z), ox = splat(o.x), oy = splat(o.y), oz = splat(o.z);
    v2f nd = pfma(q1.xy, dz, pfma(q0.zw, dy, q0.xy * dx));
    v2f no = pfma(q1.xy, oz, pfma(q0.zw, oy, q0.xy * ox));
    v2f tn = q1.zw - no;
    v2f r = {rcp(nd.x), rcp(nd.y)};
    PairHit h;
    h.t = tn * r;
    v2f px = pfma(dx, h.t, ox), py = pfma(dy, h.t, oy), pz = pfma(dz, h.t, oz);
    v2f uu = pfma(q3.xy, pz, pfma(q2.zw, py, q2.xy * px)) + q3.zw;
    v2f vv = pfma(q5.xy, pz, pfma(q4.zw, py, q4.xy * px)) + q5.zw;
    v2f mu = splat(1.0f) - uu, mv = splat(1.0f) - vv;
    h.c.x = fminf(fminf(fminf(uu.x, vv.x), mu.x), mv.x);
    h.c.y = fminf(fminf(fminf(uu.y, vv.y), mu.y), mv.y);
    return h;
}
// pair k of the walk, whichever kind it is (k is wave-uniform: a scalar branch)
ZD PairHit pair_test_k(const DScene &S, int k, f3 o, f3 d) {
    return (k < S.nppairs) ? pair_test_par(as_constant(S.ppairs) + 6 * k, o, d) : pair_test(as_constant(S.pairs) + 10 * k, o, d);
}

// primitive of the pair walk -> slot and barycentrics of the triangle that was hit
ZD void brute_resolve(const DScene &S, Hit &h, int prim, f3 o, f3 d) {
    if (prim < 0) { h.slot = -1; return; }
    const bool quad = prim < S.nquads2;
    int slot = quad ? 2 * prim : prim + S.nquads2;
    const f3 p = o + d * h.t;
    float4 U = S.isect[3 * (size_t)slot + 1], V = S.isect[3 * (size_t)slot + 2];
    float u = U.x * p.x + U.y * p.y + U.z * p.z + U.w, v = V.x * p.x + V.y * p.y + V.z * p.z + V.w;
    if (quad && 1.0f - (u + v) < 0.0f) {                       // beyond the diagonal: the quad's second triangle
        slot += 1;
        U = S.isect[3 * (size_t)slot + 1]; V = S.isect[3 * (size_t)slot + 2];
        u = U.x * p.x + U.y * p.y + U.z * p.z + U.w; v = V.x * p.x + V.y * p.y + V.z * p.z + V.w;
    }
    h.slot = slot; h.u = u; h.v = v;
}

struct BruteAccel {
    static constexpr bool kNeedsLds = false;
    static constexpr int kMinWavesFwd = ZDR_MIN_WAVES, kMinWavesFwdEnv = ZDR_MIN_WAVES_ENV;
    static constexpr int kMinWavesBwd = ZDR_MIN_WAVES_BWD;
    static constexpr int kPoolSlots = ZDR_POOL_SLOTS;        // record pool of the backward path kernel (zdr_kernels.hip)
    static constexpr bool kFuseRays = false;                 // one walk over the pairs for both rays of a vertex measured no gain
    ZD static Hit closest(const DScene &S, int *, f3 o, f3 d, float tmin, float tmax) {
        Hit h; h.slot = -1; h.u = 0.0f; h.v = 0.0f; h.t = tmax;
        int prim = -1;
        auto take = [&](const PairHit &ph, int s) {
            bool ok = (ph.t.x > tmin) & (ph.t.x < h.t) & (ph.c.x >= 0.0f);
            h.t = ok ? ph.t.x : h.t; prim = ok ? s : prim;
            ok = (ph.t.y > tmin) & (ph.t.y < h.t) & (ph.c.y >= 0.0f);
            h.t = ok ? ph.t.y : h.t; prim = ok ? s + 1 : prim;
        };
        const_v4f_ptr q = as_constant(S.ppairs);
        int s = 0;
#pragma unroll 1
        for (; s < 2 * S.nppairs; s += 2, q += 6) take(pair_test_par(q, o, d), s);
        q = as_constant(S.pairs) + 5 * s;
#pragma unroll 1
        for (; s < S.nquads; s += 2, q += 10) take(pair_test(q, o, d), s);
        brute_resolve(S, h, prim, o, d);
        return h;
    }
    // Camera rays of one 8x8 tile: only the pairs whose bit is set in the tile's mask can be hit
    // (k_tile_masks); the mask is wave-uniform, so the walk over its set bits is scalar code.
    ZD static Hit closest_camera(const DScene &S, int *, f3 o, f3 d, unsigned long long mask) {
        if (S.ntris > 128) return closest(S, nullptr, o, d, 0.0f, 1e30f);   // more pairs than mask bits
        Hit h; h.slot = -1; h.u = 0.0f; h.v = 0.0f; h.t = 1e30f;
        int prim = -1;
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)mask), hi = __builtin_amdgcn_readfirstlane((unsigned)(mask >> 32));
        unsigned long long m = ((unsigned long long)hi << 32) | lo;
#pragma unroll 1
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1ull;
            PairHit ph = pair_test_k(S, k, o, d);
            const int s = 2 * k;
            bool ok = (ph.t.x > 0.0f) & (ph.t.x < h.t) & (ph.c.x >= 0.0f);
            h.t = ok ? ph.t.x : h.t; prim = ok ? s : prim;
            ok = (ph.t.y > 0.0f) & (ph.t.y < h.t) & (ph.c.y >= 0.0f);
            h.t = ok ? ph.t.y : h.t; prim = ok ? s + 1 : prim;
        }
        brute_resolve(S, h, prim, o, d);
        return h;
    }
    // The shadow segment of next-event estimation (prb.py:59, direct.py:44): only the pairs of S.shadow_pairs can lie between a
    // surface point and a point of a light (the host proves it for the others: zdr_api.cpp, never_occluders); the mask is
    // wave-uniform, the walk over its set bits scalar code.  Same answer as any(), bit for bit.
    ZD static bool any_shadow(const DScene &S, int *, f3 o, f3 d, float tmin, float tmax) {
        bool occ = false;
        unsigned long long m = S.shadow_pairs;
        const int npairs = (S.nquads + 1) >> 1;
        if (npairs < 64) m &= (1ull << npairs) - 1ull;
        if (S.ntris > 128) return any(S, nullptr, o, d, tmin, tmax);   // more pairs than mask bits
#pragma unroll 1
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1ull;
            const PairHit ph = pair_test_k(S, k, o, d);
            occ |= ((ph.t.x > tmin) & (ph.t.x < tmax) & (ph.c.x >= 0.0f)) | ((ph.t.y > tmin) & (ph.t.y < tmax) & (ph.c.y >= 0.0f));
        }
        return occ;
    }
    ZD static bool any(const DScene &S, int *, f3 o, f3 d, float tmin, float tmax) {
        bool occ = false;
        auto take = [&](const PairHit &ph) {
            occ |= ((ph.t.x > tmin) & (ph.t.x < tmax) & (ph.c.x >= 0.0f)) | ((ph.t.y > tmin) & (ph.t.y < tmax) & (ph.c.y >= 0.0f));
        };
        const_v4f_ptr q = as_constant(S.ppairs);
        int s = 0;
#pragma unroll 1
        for (; s < 2 * S.nppairs; s += 2, q += 6) take(pair_test_par(q, o, d));
        q = as_constant(S.pairs) + 5 * s;
#pragma unroll 1
        for (; s < S.nquads; s += 2, q += 10) take(pair_test(q, o, d));
        return occ;
    }
};

// Slab test against child K of a quantised node (byte K of each q word); returns the entry distance, or 3e38 on a miss.
// nq / fq: per axis the word of the planes the ray ENTERS through and the one it LEAVES through — chosen once per node
// from the sign of the direction (for all four children at once), so a child costs six conversions, three packed FMAs
// {near, far} = q {A, A} + {B, B}, one max3 / min3 pair with the ray interval and a compare (25 -> 17 VALU; the BVH
// kernels are VALU-issue-bound).  fminf / fmaxf drop NaNs (0 * inf when a direction component is 0): an axis that cannot
// be decided is ignored, which keeps the test conservative.
template <int K> ZD float ubyte(uint32_t w) {
    return (float)((w >> (8 * K)) & 0xffu);              // v_cvt_f32_ubyteK
}
template <int K>
ZD float qbox_entry(uint32_t nxq, uint32_t nyq, uint32_t nzq, uint32_t fxq, uint32_t fyq, uint32_t fzq, f3 A, f3 B, float tmin, float tmax) {
    v2f qx = {ubyte<K>(nxq), ubyte<K>(fxq)}, qy = {ubyte<K>(nyq), ubyte<K>(fyq)}, qz = {ubyte<K>(nzq), ubyte<K>(fzq)};
    v2f tx = pfma(qx, splat(A.x), splat(B.x)), ty = pfma(qy, splat(A.y), splat(B.y)), tz = pfma(qz, splat(A.z), splat(B.z));
    float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
    float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax));
    return (tn <= tf) ? tn : 3.0e38f;
}

struct BvhAccel {
    static constexpr bool kNeedsLds = true;
    static constexpr int kMinWavesFwdEnv = ZDR_MIN_WAVES_BVH;
    static constexpr int kMinWavesFwd = ZDR_MIN_WAVES_BVH;   // 5 waves per SIMD (<= 96 VGPRs: the path state that is cold during the walk is spilled around it); sweep at ZDR_MIN_WAVES_BVH
    static constexpr int kMinWavesBwd = ZDR_MIN_WAVES_BWD_BVH;   // backward: LDS decides the waves per CU; one record in LDS and
    static constexpr int kPoolSlots = ZDR_POOL_SLOTS_BVH;
    ZD static Hit closest_camera(const DScene &S, int *stack, f3 o, f3 d, unsigned long long) { return closest(S, stack, o, d, 0.0f, 1e30f); }
    static constexpr bool kFuseRays = true;                  // path_shade hands over both rays of a vertex at once (walk<true, true>)
    // 4-wide BVH, one 64-byte quantised node per visit (4 dwordx4 loads), nearest hit child first.
    // stack: this wave's LDS region, min(S.stack_entries, ZDR_BVH_LDS_STACK) x 64 ints; entry e of lane l at stack[e * 64 + l].
    // A work item is (id, cnt): cnt == 0 -> node id, 1..4 -> leaf slots [id, id + cnt); 7 marks an unused child and is never pushed.
    struct Walker {                  // one ray in flight on this lane
        f3 o, d, inv; float tmin; Hit h;          // h.slot while walking: the hit triangle's record as a 16-byte offset from S.walk_base (slot_of), -1 = none
        int sp; uint32_t off; int cnt;            // what the ray stands on: byte offset from S.walk_base of a node (cnt == 0) or of a leaf's first plane record (cnt 1..2); 7 = an unused child slot
    };
    ZD static int root_count(const DScene &S) { return (S.nnodes == 0) ? S.ntris : 0; }
    // Watchdog: a correct walk visits every node and leaf at most once, so a lane's two rays need at most twice that many trips of the
    // wave's loop — which counts its trips in ONE wave-uniform (scalar) counter: the bound makes it impossible for a wave to spin forever
    // whatever the node data or the rays (NaNs) look like, at no cost to the vector ALU (a per-lane budget was two VALU in every trip).
    ZD static int walk_budget(const DScene &S) { return (S.debug_bvh_budget > 0) ? S.debug_bvh_budget : 2 * (S.nnodes + S.ntris) + 8; }
    ZD static void start(const DScene &S, Walker &w, f3 o, f3 d, float tmin, float tmax) {
        w.o = o; w.d = d; w.tmin = tmin; w.inv = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
        w.h.slot = -1; w.h.u = 0.0f; w.h.v = 0.0f; w.h.t = tmax;
        w.sp = 0; w.off = (S.nnodes == 0) ? S.isect_off : 0u; w.cnt = root_count(S);
    }
    // the slot of the triangle whose plane record starts `units` 16-byte units behind S.walk_base (records are 48 bytes)
    ZD static int slot_of(const DScene &S, int units) { return (int)(__umulhi((uint32_t)units - (S.isect_off >> 4), 0xAAAAAAABu) >> 1); }
    // One visit (a node or a leaf) and the pop that follows it, in two halves: fetch() issues the loads of whatever the
    // ray stands on, consume() uses them and returns true while the ray has more to visit.
    // ONE memory round trip per trip of the wave.  The walk is latency-bound (waves sit in s_waitcnt 2/3 of the time):
    // what counts is how many dependent round trips a wave makes, and a wave whose lanes are partly at nodes and partly
    // at leaves used to make one for the node branch, then one per triangle of the leaf branch.  Here every lane first
    // issues its loads — a node (64 B) or the <= 2 triangles of a leaf (48 B each) sit behind ONE per-lane pointer, four
    // dwordx4 loads for everybody, two more for a second triangle — and only then do the branches consume them.
    // Entries [0, LN) of the stack live in LDS, deeper ones in per-lane scratch (`deep`): the builder's bound (up to 44
    // entries on a 1 M triangle tree, 11 KiB of LDS per wave) is a worst case that real rays almost never approach,
    // and LDS is what limits the waves per CU of the BVH kernels.
    struct Fetched { float4 n0, n1, n2, n3, n4, n5; bool dead; };
    // No slot of a Fetched is zero-filled: n0..n3 are loaded by every lane (a lane that stands on an unused slot reads the first records of
    // the triangle array instead — any valid address — and consume() looks at `dead` before anything else), n4 / n5 are written only
    // for a leaf of more than one triangle, which is exactly when consume() reads them.  Zero-filling the 24 registers cost 23 v_mov in
    // EVERY trip of the walk (a frozen undef is materialised as a zero too): 7 % of its instructions.
    // Nodes and plane records live in one allocation (zdr_api.cpp) and a child word IS the byte offset of what it names (| its count
    // in the low bits: offsets are multiples of 16): the fetch address costs one AND — no branch between "node" and "triangle", no
    // multiply by the record size (that was a four-pass v_mad_u64_u32 in every trip).
    // An unused child slot carries an inverted box that no ray with a finite direction can enter, so the node branch does not test
    // for it; should a ray of NaNs be sent into one all the same (every comparison of its slab test is undecided), the walk ends
    // here: such a ray hits nothing.
    ZD static Fetched fetch(const DScene &S, int *stack, Walker &w) {
        Fetched f;
        f.dead = (w.cnt == 7);
        const uint32_t off = f.dead ? S.isect_off : w.off;      // (a dead lane reads the first records of the triangle array: >= 2 records = 6 float4 are always there)
        const float4 *p = (const float4 *)(S.walk_base + off);
        f.n0 = p[0]; f.n1 = p[1]; f.n2 = p[2]; f.n3 = p[3];
        if (w.cnt > 1) { f.n4 = p[4]; f.n5 = p[5]; }
        return f;
    }
    ZD static bool consume(const DScene &S, int *stack, const int LN, int *deep, Walker &w, const Fetched &f, const bool anyhit, const int bot = 0) {
        const int lane = threadIdx.x & 63;
        bool ray_done = f.dead;
        const f3 o = w.o, d = w.d, inv = w.inv; const float tmin = w.tmin;
        const int cnt = w.cnt;
        const float4 n0 = f.n0, n1 = f.n1, n2 = f.n2, n3 = f.n3;
        if (!ray_done && cnt == 0) {
            // slab distances on the node's quantisation grid: t = (origin + scale q - o) / d = q A + B
            const f3 A = mk3(n0.w * inv.x, n1.x * inv.y, n1.y * inv.z);
            const f3 B = mk3((n0.x - o.x) * inv.x, (n0.y - o.y) * inv.y, (n0.z - o.z) * inv.z);
            const uint32_t lxq = __float_as_uint(n1.z), lyq = __float_as_uint(n1.w), lzq = __float_as_uint(n2.x);
            const uint32_t hxq = __float_as_uint(n2.y), hyq = __float_as_uint(n2.z), hzq = __float_as_uint(n2.w);
            // the planes the ray enters / leaves through, for the four children at once
            const bool ngx = inv.x < 0.0f, ngy = inv.y < 0.0f, ngz = inv.z < 0.0f;
            const uint32_t nxq = ngx ? hxq : lxq, fxq = ngx ? lxq : hxq, nyq = ngy ? hyq : lyq, fyq = ngy ? lyq : hyq, nzq = ngz ? hzq : lzq, fzq = ngz ? lzq : hzq;
            const float e0 = qbox_entry<0>(nxq, nyq, nzq, fxq, fyq, fzq, A, B, tmin, w.h.t);
            const float e1 = qbox_entry<1>(nxq, nyq, nzq, fxq, fyq, fzq, A, B, tmin, w.h.t);
            const float e2 = qbox_entry<2>(nxq, nyq, nzq, fxq, fyq, fzq, A, B, tmin, w.h.t);
            const float e3 = qbox_entry<3>(nxq, nyq, nzq, fxq, fyq, fzq, A, B, tmin, w.h.t);
            const int p0 = __float_as_int(n3.x), p1 = __float_as_int(n3.y), p2 = __float_as_int(n3.z), p3 = __float_as_int(n3.w);
            // nearest child: visit now; the other hit children go on the stack.  On the fast path the four
            // stack writes are unconditional (LDS stores are cheap, branches are not): a slot is kept only
            // if sp advances past it.
            const float em = fminf(fminf(e0, e1), fminf(e2, e3));
            if (em < 2.0e38f) {
                const bool t0 = (e0 == em), t1 = !t0 & (e1 == em), t2 = !(t0 | t1) & (e2 == em), t3 = !(t0 | t1 | t2);
                const int next = t0 ? p0 : (t1 ? p1 : (t2 ? p2 : p3));
                int sp = w.sp;
                if (sp + 4 <= LN) {
                    int at = sp * 64 + lane;                 // (an index, not a pointer: the pointer form compiled to 64-bit arithmetic)
                    stack[at] = p0; at += (!t0 & (e0 < 2.0e38f)) ? 64 : 0;
                    stack[at] = p1; at += (!t1 & (e1 < 2.0e38f)) ? 64 : 0;
                    stack[at] = p2; at += (!t2 & (e2 < 2.0e38f)) ? 64 : 0;
                    stack[at] = p3; at += (!t3 & (e3 < 2.0e38f)) ? 64 : 0;
                    sp = (at - lane) >> 6;
                } else {                                // near or past the LDS part: one entry at a time
                    const int pp[4] = {p0, p1, p2, p3};
                    const bool keep[4] = {!t0 && e0 < 2.0e38f, !t1 && e1 < 2.0e38f, !t2 && e2 < 2.0e38f, !t3 && e3 < 2.0e38f};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (keep[j]) {
                            if (sp < LN) stack[sp * 64 + lane] = pp[j]; else deep[sp - LN] = pp[j];
                            sp++;
                        }
                    }
                }
                w.sp = sp; w.off = (uint32_t)next & ~15u; w.cnt = next & 7;
                return true;
            }
        } else if (!ray_done) {
            const int units = (int)(w.off >> 4);
            float t;
            if (tri_test(n0, n1, n2, o, d, tmin, w.h.t, t)) { w.h.t = t; w.h.slot = units; }
            if (cnt > 1 && tri_test(n3, f.n4, f.n5, o, d, tmin, w.h.t, t)) { w.h.t = t; w.h.slot = units + 3; }
#if ZDR_BVH_LEAF > 2
            for (int s = 2; s < cnt; s++) {               // leaves of more than two triangles
                const float4 *q = (const float4 *)(S.walk_base + w.off) + 3 * s;
                if (tri_test(q[0], q[1], q[2], o, d, tmin, w.h.t, t)) { w.h.t = t; w.h.slot = units + 3 * s; }
            }
#endif
            if (anyhit && w.h.slot >= 0) ray_done = true;         // any-hit: the first hit settles it
        }
        if (!ray_done && w.sp != bot) {               // (bot: entries below it were taken by other lanes, walk_steal; 0 otherwise)
            w.sp--;
            // The pop is on the critical path of every trip.  Written as one conditional expression the compiler merges the LDS and the
            // scratch access into a FLAT load (generic pointer, aperture check, vector-memory issue and latency even for the LDS case);
            // an unconditional ds_read of a clamped entry plus a branch that is skipped unless some lane is beyond the LDS part keeps it in LDS.
            typedef __attribute__((address_space(3))) int lds_int_t;      // an explicit LDS pointer: ds_read_b32, whatever the optimiser thinks of the scratch access next to it
            int e = ((lds_int_t *)stack)[((w.sp < LN) ? w.sp : 0) * 64 + lane];
            if (w.sp >= LN) e = deep[w.sp - LN];
            w.off = (uint32_t)e & ~15u; w.cnt = e & 7;
            return true;
        }
        return false;
    }
    ZD static bool step(const DScene &S, int *stack, const int LN, int *deep, Walker &w, const bool anyhit, const int bot = 0) {
        const Fetched f = fetch(S, stack, w);
        return consume(S, stack, LN, deep, w, f, anyhit, bot);
    }
    // One loop walks up to two rays per lane back to back: first (HAS_A) an any-hit ray — the shadow segment of a
    // path vertex — then (HAS_B, lanes with needB) a closest-hit ray — the continuation ray.  A lane starts its second
    // ray the moment its first one ends, so the wave's trip count is the longest SUM of the two walks over its
    // lanes, not the sum of the two longest walks.
    template <bool HAS_A, bool HAS_B>
    ZD static void walk(const DScene &S, int *stack, f3 oA, f3 dA, float tminA, float tmaxA,
                        bool needB, f3 oB, f3 dB, float tminB, float tmaxB, bool &occ, Hit &hit, bool needA = true) {
        occ = false;
        hit.slot = -1; hit.u = 0.0f; hit.v = 0.0f; hit.t = tmaxB;
        bool first = HAS_A && needA;                         // this lane is still on its any-hit ray
        bool active = first || (HAS_B && needB);             // this lane has a ray in flight
        const int LN = S.lds_stack;
        int deep[ZDR_BVH_STACK];
        Walker w;
        if (first) start(S, w, oA, dA, tminA, tmaxA); else start(S, w, oB, dB, tminB, tmaxB);
        // ONE loop with a wave-uniform exit: a lane whose first ray ends starts its second ray inside the trip, under its own exec
        // mask.  (Written as `for (;;) { if (step()) continue; ...restart...; continue; }` the compiler splits the loop in two nested
        // ones — an inner one that runs until EVERY lane's current ray has ended, an outer one that restarts them together — and the
        // wave's trip count becomes max(first walks) + max(second walks) instead of max(first + second): found in the ISA in round 3.)
        int trips_left = ((HAS_A && HAS_B) ? 2 : 1) * walk_budget(S);   // wave-uniform
        while (__ballot(active) != 0ull) {
            if (--trips_left < 0) { raise_device_error(S, ZDR_DEVERR_BVH_BUDGET); break; }   // the walks were cut short: whatever they return is not a result
            if (active) {
                if (!step(S, stack, LN, deep, w, first)) {   // this lane's current ray has ended
                    if (first) {
                        occ = w.h.slot >= 0;
                        first = false;
                        if (HAS_B && needB) start(S, w, oB, dB, tminB, tmaxB); else active = false;
                    } else active = false;
                }
            }
        }
        if (HAS_B && needB) { hit = w.h; if (hit.slot >= 0) hit.slot = slot_of(S, hit.slot); hit_barycentrics(S, hit, oB, dB); }
    }
#if ZDR_BVH_STEAL
    // The fused walk with SUBTREE STEALING (round 4; costed in profiles/r4_steal_sim.txt, measured in profiles/r4_subtree_stealing.txt).  The
    // wave's trip count is set by its slowest lane — 67 trips for 40 visits of the mean lane on the 1 M-triangle scene — and the stacks of all
    // lanes live in the wave's LDS.  So a lane that is through takes the BOTTOM entry (the farthest subtree still waiting) of the LDS stack of a
    // lane that still has entries there — every ZDR_BVH_STEAL-th trip, the k-th idle lane from the k-th lane with entries, matched through a
    // 64-entry LDS list — and walks that subtree with a copy of the owner's ray (nine ds_bpermute: no LDS held for rays).  A robbed lane learns
    // its new bottom from LDS after the steal trip (consume() pops down to `bot`, not to 0).  Every hit goes to the LDS cell of the lane that
    // OWNS the ray — ds_min_u64 on {t, record} for the closest-hit ray, ds_or for the any-hit ray — and that is where the results are read when
    // the loop ends; on steal trips a closest-hit walker also takes over the best distance the others have found.  The answers are those of
    // walk(): the same triangles are tested against the same rays by the same code, only by other lanes (min over {t, record} instead of "first
    // found" decides an exact tie of two triangles, which needs a ray through a shared edge to the last bit); images and gradients of the c5
    // bench are identical to the last digit printed with and without; tests/test_gpu_c5.py compares every path of 73,728 with the oracle.
    // LDS beyond the stack's LN x 64 ints (bvh_dyn_lds, zdr_kernels.hip): cellB 64 x u64, cellA 64, bottoms 64, match list 64 ints = 1,280 bytes.
    typedef __attribute__((address_space(3))) int lds_i32;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    ZD static float pull(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
    ZD static uint32_t rank_in(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }
    template <bool HAS_A, bool HAS_B>
    ZD static void walk_steal(const DScene &S, int *stack, f3 oA, f3 dA, float tminA, float tmaxA,
                              bool needB, f3 oB, f3 dB, float tminB, float tmaxB, bool &occ, Hit &hit, bool needA = true) {
        const int lane = threadIdx.x & 63;
        const int LN = S.lds_stack;
        lds_i32 *stk = (lds_i32 *)stack;
        lds_u64 *cellB = (lds_u64 *)(stk + LN * 64);
        lds_i32 *cellA = stk + LN * 64 + 128, *lbot = cellA + 64, *llist = cellA + 128;
        cellB[lane] = ((unsigned long long)__float_as_uint(tmaxB) << 32) | 0xffffffffull;
        cellA[lane] = 0; lbot[lane] = 0;
        bool first = HAS_A && needA;
        bool active = first || (HAS_B && needB);
        bool mine = true;                                    // the lane is on its own rays (not on a subtree it took)
        int owner = lane, bot = 0;
        int deep[ZDR_BVH_STACK];
        Walker w;
        if (first) start(S, w, oA, dA, tminA, tmaxA); else start(S, w, oB, dB, tminB, tmaxB);
        auto end_task = [&]() {                              // this lane's current ray (or the subtree it took) is through
            if (mine && first && HAS_B && needB) { first = false; start(S, w, oB, dB, tminB, tmaxB); bot = 0; lbot[lane] = 0; }
            else active = false;
        };
        int trips_left = ((HAS_A && HAS_B) ? 2 : 1) * walk_budget(S);
        while (__ballot(active) != 0ull) {
            if (--trips_left < 0) { raise_device_error(S, ZDR_DEVERR_BVH_BUDGET); break; }
            if (active) {
                const float t0 = w.h.t;
                const bool more = step(S, stack, LN, deep, w, first, bot);
                if (w.h.t != t0) {                           // a hit: to the owner's cell
                    if (first) __hip_atomic_fetch_or(cellA + owner, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    else __hip_atomic_fetch_min(cellB + owner, ((unsigned long long)__float_as_uint(w.h.t) << 32) | (unsigned int)w.h.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                if (!more) end_task();
            }
            if ((trips_left % ZDR_BVH_STEAL) == 0) {
                const unsigned long long im = __ballot(!active);
                const bool cand = active && (((w.sp < LN) ? w.sp : LN) - bot) >= ZDR_BVH_STEAL_MIN;
                const unsigned long long cm = __ballot(cand);
                if (im != 0ull && cm != 0ull) {              // wave-uniform
                    if (cand) llist[rank_in(cm)] = lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int r = (int)rank_in(im);
                    const bool thief = !active && r < __popcll(cm);
                    const int victim = thief ? llist[r] : lane;
                    const int va = victim << 2;
                    const f3 vo = mk3(pull(va, w.o.x), pull(va, w.o.y), pull(va, w.o.z)), vd = mk3(pull(va, w.d.x), pull(va, w.d.y), pull(va, w.d.z));
                    const float vtmin = pull(va, w.tmin), vt = pull(va, w.h.t);
                    const int vbot = __builtin_amdgcn_ds_bpermute(va, bot), vowner = __builtin_amdgcn_ds_bpermute(va, owner), vfirst = __builtin_amdgcn_ds_bpermute(va, first ? 1 : 0);
                    if (thief) {
                        const int e = stk[vbot * 64 + victim];
                        lbot[victim] = vbot + 1;
                        lbot[lane] = 0;
                        w.o = vo; w.d = vd; w.inv = mk3(rcp(vd.x), rcp(vd.y), rcp(vd.z)); w.tmin = vtmin;
                        w.h.t = vt; w.h.slot = -1; w.h.u = 0.0f; w.h.v = 0.0f;
                        w.sp = 0; bot = 0; w.off = (uint32_t)e & ~15u; w.cnt = e & 7;
                        owner = vowner; first = vfirst != 0; mine = false; active = true;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (cand) bot = lbot[lane];
                }
                // what the others found for this lane's ray meanwhile prunes its walk as well
                if (active && !first) w.h.t = fminf(w.h.t, __uint_as_float((uint32_t)(cellB[owner] >> 32)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        occ = HAS_A && needA && cellA[lane] != 0;
        hit.slot = -1; hit.u = 0.0f; hit.v = 0.0f; hit.t = tmaxB;
        if (HAS_B && needB) {
            const unsigned long long c = cellB[lane];
            hit.t = __uint_as_float((uint32_t)(c >> 32));
            const int units = (int)(uint32_t)c;
            if (units >= 0) hit.slot = slot_of(S, units);
            hit_barycentrics(S, hit, oB, dB);
        }
    }
#endif
    ZD static Hit closest(const DScene &S, int *stack, f3 o, f3 d, float tmin, float tmax) {
        bool occ; Hit h;
        walk<false, true>(S, stack, o, d, tmin, tmax, true, o, d, tmin, tmax, occ, h);
        return h;
    }
    ZD static bool any(const DScene &S, int *stack, f3 o, f3 d, float tmin, float tmax) {
        bool occ; Hit h;
        walk<true, false>(S, stack, o, d, tmin, tmax, false, o, d, tmin, tmax, occ, h);
        return occ;
    }
    ZD static bool any_shadow(const DScene &S, int *stack, f3 o, f3 d, float tmin, float tmax) { return any(S, stack, o, d, tmin, tmax); }
    // need1: the lane has a shadow ray (o1, d1) at all; need2: it has a continuation ray (o2, d2)
    ZD static void shadow_and_closest(const DScene &S, int *stack, bool need1, f3 o1, f3 d1, float tmin1, float tmax1, bool need2, f3 o2, f3 d2, bool &occ, Hit &h) {
#if ZDR_BVH_STEAL
        walk_steal<true, true>(S, stack, o1, d1, tmin1, tmax1, need2, o2, d2, 0.0f, 1e30f, occ, h, need1);
#else
        walk<true, true>(S, stack, o1, d1, tmin1, tmax1, need2, o2, d2, 0.0f, 1e30f, occ, h, need1);
#endif
    }
};
