// scene.h — device-side scene: what LuisaCompute's Accel + bindless heap held for the reference
// (render.py:73-128).  Instances are flattened to world space at build time; triangles live in
// "slot" order (the BVH's leaf order, or input order for the brute-force accel).
#pragma once
#include "vecmath.h"

// largest leaf of the BVH builder (zdr_api.cpp); the walk fetches two triangles with the node fetch and loops over any further ones
#ifndef ZDR_BVH_STEAL
#define ZDR_BVH_STEAL 2     // subtree stealing in the fused walk of the BVH path kernels (accel.h, walk_steal): lanes that are through may take work in every
#endif                      // ZDR_BVH_STEAL-th trip (0: off).  1 M triangles, path 1024^2 spp 32, forward / backward ms: off 23.7 / 29.9, every trip 21.2 / 27.4, 2nd 20.9 / 26.8, 4th 21.6 / 27.7
#ifndef ZDR_BVH_STEAL_MIN
#define ZDR_BVH_STEAL_MIN 1 // a lane can be robbed while it has this many entries in the LDS part of its stack (2: forward 22.3 instead of 21.1 ms; 3: 23.5)
#endif
#ifndef ZDR_BVH_LEAF
#define ZDR_BVH_LEAF 2   // 1 M triangles, path fwd / bwd ms at 1024^2 spp 32: leaf 1: 46 / 61, 2: 42 / 56, 3: 43 / 58, 4: 48 / 64, 6: 54 / 72 (a triangle costs three per-lane loads, a node four)
#endif
#define ZDR_BVH_STACK 64   // upper bound of per-lane traversal stack entries; the builder bounds the tree depth below it
// of which this many live in LDS (the rest, rarely reached, in scratch), per kind of kernel: the backward kernel's waves per CU are
// decided by LDS, the forward kernel's by VGPRs.  1 M triangles, 1024^2 spp 32, ms with 6 / 8 / 12 / 16 entries in LDS: forward
// 32.9 / 31.8 / 31.4 / 31.3, backward 39.9 / 41.5 / 40.8 / 44.1.
#ifndef ZDR_BVH_LDS_STACK
#define ZDR_BVH_LDS_STACK 12
#endif
#ifndef ZDR_BVH_LDS_STACK_BWD
#define ZDR_BVH_LDS_STACK_BWD 10   // (6 until the backward kernel's records moved into a pool: 1 M triangles, 1024^2 spp 32, stack entries / pool slots in the same 8 LDS blocks: 4 / 90 35.2 ms, 6 / 84 34.0, 10 / 72 32.8, 12 / 65 32.8, 14 / 59 32.8)
#endif
// (A per-wave LDS copy of the top of the tree measured slower — K = 5 nodes: equal, 21: +5 %, 64: +20 %, profiles/r3_bvh_lds_top.txt: the top nodes
// are the ones every wave finds in L1 anyway — and was removed in round 4; the code is kept in profiles/r4_pruned_experiment_branches.patch.)

// Per-slot records, 16-byte aligned so a record is fetched with dwordx4 loads:
//   isect[3*slot + {0,1,2}] = {n, n.p0} {nu, du} {nv, dv}              (48 B, plane-form triangle test)
//   shade[8*slot + ...]     = one 128-byte line:                       (surface_interact, lights)
//     r0 {p0.xyz, uv0.x} r1 {p1.xyz, uv0.y} r2 {p2.xyz, uv1.x}
//     r3 {n0.xyz, uv1.y} r4 {n1.xyz, uv2.x} r5 {n2.xyz, uv2.y}   n_i = inverse-transpose(M) * vn_i
//     r6 {ng.xyz, bits(inst)}                                    ng = normalize(cross(p1-p0, p2-p0))
//     r7 {area, bits(prim), bits(rotation), 0}                                read only on an emitter hit / by the ray-query kernels
// BVH4 node (64 B, four float4; child boxes quantised to 8 bits per plane on the node's own grid, rounded outwards):
//   {origin.xyz, scale.x} {scale.y, scale.z, qlo.x[4], qlo.y[4]} {qlo.z[4], qhi.x[4], qhi.y[4], qhi.z[4]} {child[4]}
//   plane = origin + scale * q (byte k of a q word belongs to child k); child = index << 3 | count:
//   count == 0: index is a node; 1..4: first slot of a leaf of `count` triangles; 7: unused child.
struct DScene {
    const float4 *isect;
    const float4 *pairs;            // brute-force accel only: plane + four edge functions of primitives (2k, 2k+1) interleaved, 10 float4 per pair (accel.h)
    const float4 *shade;
    const float4 *nodes;
    const float *emission;          // ninst x 3   (heap slot 23333)
    const int32_t *light_insts;     // ninst       (heap slot 23334)
    const int32_t *inst_tri_begin;  // ninst + 1   (heap slot 23335 holds the counts)
    const int32_t *slot_of_tri;     // input triangle index -> slot
    // flat light table: light l covers entries [light_range[2l], + light_range[2l+1]); entry = five float4
    //   {p0} {p1} {p2} {ng, area} {emission, 0}   (the floats of the shade records and of `emission`)
    const float4 *light_tris; const int32_t *light_range; const float4 *emission4;   // emission4: ninst x {e.rgb, 0}
    int32_t light0_T;               // triangle count of light 0 (the whole table when light_count == 1)
    int32_t ntris, ninst, light_count, nnodes;
    int32_t nquads, nquads2;        // brute-force accel: primitives of the pair walk (quads first, then single triangles) and the number of quads
    unsigned long long shadow_pairs;   // brute-force accel: bit k set = pair k holds a primitive that a shadow segment (surface point -> light point) can meet; the others support the scene from outside (zdr_api.cpp, never_occluders).  Set per launch.
    const float4 *ppairs; int32_t nppairs;   // the first nppairs pairs hold two parallelograms each: plane + u, v of the first triangle, 6 float4 per pair
    // environment light (envmap.py; heap slots 23330-23332): lat-long RGBA texture + importance tables
    const float4 *env_tex; const float *alias_prob; const int32_t *alias_idx; const float *env_pdf;
    int32_t env_count, env_h, env_w, map_w, map_h;
    const char *walk_base; uint32_t isect_off;   // BVH: nodes[] and isect[] live in ONE allocation, nodes first; isect[] starts isect_off bytes behind walk_base
    int32_t stack_entries;          // per-lane traversal stack entries this tree needs
    int32_t lds_stack;              // how many of them this launch keeps in LDS (dynamic LDS: lds_stack x 64 ints per wave), set by the launcher
    // Device error word (sticky until the host reads it: zdr_scene_check, zdr_render_stats, ZDR_CHECK=1).  A watchdog
    // that ends work early ORs its bit in, so an incomplete image or gradient can never pass as a good one.
    unsigned int *error_word;
    int32_t debug_bvh_budget;       // > 0: iteration budget of every BVH walk (env ZDR_DEBUG_BVH_BUDGET at scene creation) — lets a test trip the watchdog
};
#define ZDR_DEVERR_STALL 1u          // a persistent path wave left its loop without having drained its work
#define ZDR_DEVERR_BVH_BUDGET 2u     // a BVH walk ran out of its iteration budget (corrupt nodes or a NaN ray that never ends)
#define ZDR_DEVERR_POOL 4u           // a backward path wave ended with slots of its record pool leaked or doubly allocated (zdr_kernels.hip, lds_free)
ZD void raise_device_error(const DScene &S, unsigned int bit) { if (S.error_word) atomicOr(S.error_word, bit); }

struct Hit { int slot; float u, v, t; };   // slot < 0: miss (LuisaCompute Hit{inst, prim, bary, ray_t})

struct Interaction {                       // interaction.py:6
    f3 p; f2 uv; f3 ns, ng; int inst, prim;
};

ZD f3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }

ZD Interaction surface_interact(const DScene &S, const Hit &h) {   // interaction.py:9-30
    const float4 *r = S.shade + 8 * (size_t)h.slot;
    float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6];   // seven of the record's eight float4
    float w0 = 1.0f - h.u - h.v, w1 = h.u, w2 = h.v;               // Hit::interpolate
    Interaction it;
    it.p = xyz(r0) * w0 + xyz(r1) * w1 + xyz(r2) * w2;
    it.uv.x = w0 * r0.w + w1 * r2.w + w2 * r4.w;
    it.uv.y = w0 * r1.w + w1 * r3.w + w2 * r5.w;
    it.ns = normalize(xyz(r3) * w0 + xyz(r4) * w1 + xyz(r5) * w2);
    it.ng = xyz(r6);
    it.inst = __float_as_int(r6.w); it.prim = 0;                   // prim is only reported by the ray-query kernels (r7.y)
    return it;
}

// read_bsdf (interaction.py:47-60): bilinear, CLAMP addressing, texel (x, y) at x + tex_w * y
struct TexFoot { int i00, i01, i10, i11; float ox, oy; };   // element indices (x4 floats) of the 4 texels

ZD TexFoot tex_footprint(f2 uv, int tex_h, int tex_w) {
    float px = uv.x * (float)(tex_w - 1), py = (1.0f - uv.y) * (float)(tex_h - 1);
    int ix = (int)px, iy = (int)py;
    TexFoot f;
    f.ox = px - (float)ix; f.oy = py - (float)iy;
    int x0 = clampi(ix, 0, tex_w - 1), x1 = clampi(ix + 1, 0, tex_w - 1);
    int y0 = clampi(iy, 0, tex_h - 1), y1 = clampi(iy + 1, 0, tex_h - 1);
    f.i00 = x0 + tex_w * y0; f.i01 = x0 + tex_w * y1; f.i10 = x1 + tex_w * y0; f.i11 = x1 + tex_w * y1;
    return f;
}

ZD float4 read_bsdf(const float4 *__restrict__ mat, f2 uv, int tex_h, int tex_w) {
    TexFoot f = tex_footprint(uv, tex_h, tex_w);
    float4 c00 = mat[f.i00], c01 = mat[f.i01], c10 = mat[f.i10], c11 = mat[f.i11];
    float4 r;
    r.x = lerpf(lerpf(c00.x, c01.x, f.oy), lerpf(c10.x, c11.x, f.oy), f.ox);
    r.y = lerpf(lerpf(c00.y, c01.y, f.oy), lerpf(c10.y, c11.y, f.oy), f.ox);
    r.z = lerpf(lerpf(c00.z, c01.z, f.oy), lerpf(c10.z, c11.z, f.oy), f.ox);
    r.w = lerpf(lerpf(c00.w, c01.w, f.oy), lerpf(c10.w, c11.w, f.oy), f.ox);
    return r;
}

// ---------------------------------------------------------------- gradient scatter (backward)
// The reference adds k_ij * dmat to the four texels of the bilinear footprint with 16 float atomics
// (interaction.py:63-89) — 16 scattered 4-byte requests to the memory-side atomic unit per shaded
// vertex, which is what bounds the backward pass (README.md:21 warns about it; measured here:
// 60 of 95 ms).  Same arithmetic, different bookkeeping:
//  * Corner staging cells.  The four products k00*g, k01*g, k10*g, k11*g of one vertex are added to
//    ONE 64-byte cell indexed by the footprint's base texel (ix, iy), cell = 16 floats
//    [corner m = 2*dx + dy][channel].  k_cells_to_grad then gathers, per texel, the (up to four)
//    cells whose footprint covers it.  CLAMP addressing folds out-of-range bases onto the border
//    cells, which is exact because the clamped corners coincide (weights sum to the same texel).
//  * Per-wavefront LDS queue, transposed flush.  Lanes push (g, uv); on flush 16 lanes
//    serve one vertex (lane j adds float j of the cell), so a wave instruction carries four whole
//    64-byte cells instead of 64 unrelated dwords: one atomic request per vertex instead of 16.
#ifndef ZDR_SCATTER_CAP
#define ZDR_SCATTER_CAP 64           // queue entries per wavefront (7 dwords each)
#endif

// The queue is emptied as soon as it holds ZDR_SCATTER_FLUSH_AT entries (and before it would overflow).  Small
// bursts matter: a 64-byte atomic occupies the CU's vector-memory path for about 30 cycles
// (profiles/r1_atomic_rate.txt) and every load of every wave of the CU queues behind the burst.
#ifndef ZDR_SCATTER_FLUSH_AT
#define ZDR_SCATTER_FLUSH_AT ZDR_SCATTER_CAP
#endif
static_assert(ZDR_SCATTER_CAP >= 64, "one push can add an entry per lane");

//  * Few texels (README.md:21 of the reference: "atomic_fetch_add will become extremely slow" when the gradients
//    concentrate on few texels — a constant or low-resolution material).  Every atomic of the launch then lands on a
//    handful of addresses, which the memory-side atomic units serialise: measured on cbox 512^2 spp 256, 64x64 texels
//    34 ms, 16x16 101 ms, 4x4 952 ms, 1x1 4.9 s instead of 17 ms — and one float32 accumulator that receives 1e8 terms
//    is off by 40 %.  Two measures, both exact re-associations of the same sum:
//      - cell COPIES: below 2^16 cells the staging array is replicated (up to 1024 times, 2^20 cells in all) and a
//        wave adds into copy blockIdx % copies; k_cells_to_grad sums the copies (in float64);
//      - at most ZDR_LDS_CELLS cells (textures up to 4x4): the wave keeps the WHOLE cell array in LDS (the queue's
//        block), adds with ds_add_f32 and writes it out once, when the kernel ends.
#define ZDR_LDS_CELLS 28             // 28 cells x 16 floats = the 448 floats of the queue's LDS block
#define ZDR_MAX_CELL_COPIES 1024
struct ScatterQueue {                // pointers into this wave's LDS block
    int *cell; float *g; float *ox; float *oy;
    int count;                       // wave-uniform
    int copy_base;                   // first cell of this wave's copy of the staging array
    int ncells;                      // (tex_h + 1) x (tex_w + 1)
    float *lds_cells;                // != nullptr: the whole cell array lives here (ncells <= ZDR_LDS_CELLS)
    bool small;                      // all copies of the cell array together hold fewer than 2^30 floats (wave-uniform)
#ifdef ZDR_MEASURE_STATS
    unsigned long long st_flushes, st_entries, st_dups;   // measurement build: flushes, entries flushed, entries whose cell an earlier entry of the same flush holds
#endif
};
#define ZDR_SCATTER_LDS_FLOATS (7 * ZDR_SCATTER_CAP)
static_assert(16 * ZDR_LDS_CELLS <= ZDR_SCATTER_LDS_FLOATS, "the LDS cell array aliases the queue's block");

// must be called by the whole wave
ZD ScatterQueue scatter_queue_init(float *lds, int tex_h, int tex_w, int cell_copies) {
    ScatterQueue q;
    q.cell = (int *)lds; q.g = lds + ZDR_SCATTER_CAP; q.ox = lds + 5 * ZDR_SCATTER_CAP; q.oy = lds + 6 * ZDR_SCATTER_CAP;
    q.count = 0;
#ifdef ZDR_MEASURE_STATS
    q.st_flushes = q.st_entries = q.st_dups = 0;
#endif
    q.ncells = (tex_h + 1) * (tex_w + 1);
    q.copy_base = (int)(blockIdx.x % (unsigned)cell_copies) * q.ncells;
    q.small = (unsigned long long)q.ncells * (unsigned long long)cell_copies * 16ull < (1ull << 30);
    q.lds_cells = (q.ncells <= ZDR_LDS_CELLS) ? lds : nullptr;
    if (q.lds_cells) {
        for (int i = threadIdx.x & 63; i < 16 * q.ncells; i += 64) lds[i] = 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return q;
}

// Footprint of a gradient at uv: base cell (clamped so that out-of-range bases fold onto the border cells) and the bilinear offsets.
ZD int scatter_cell(f2 uv, int tex_h, int tex_w, float &ox, float &oy) {
    float px = uv.x * (float)(tex_w - 1), py = (1.0f - uv.y) * (float)(tex_h - 1);   // interaction.py:78-80
    int ix = (int)px, iy = (int)py;
    ox = px - (float)ix; oy = py - (float)iy;
    int cx = clampi(ix, -1, tex_w - 1) + 1, cy = clampi(iy, -1, tex_h - 1) + 1;
    return cx + (tex_w + 1) * cy;
}

// must be called by the whole wave (reconverged control flow).  The queue holds (g, uv) as pushed; the flush first turns the uv of
// all its entries into (cell, ox, oy) — lane = entry, once per ~57 entries instead of once per push (5.45 pushes per trip of the
// backward path kernel, each by the whole wave for the few lanes that hold a gradient) — then adds them, 16 lanes per entry.
ZD void scatter_flush(ScatterQueue &q, float *__restrict__ cells, int tex_h, int tex_w, int ablate) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int lane = threadIdx.x & 63, sub = lane >> 4, j = lane & 15;
    for (int e = lane; e < q.count; e += 64) {
        f2 uv; uv.x = q.ox[e]; uv.y = q.oy[e];
        float ox, oy;
        const int cell = scatter_cell(uv, tex_h, tex_w, ox, oy);
        q.cell[e] = (ablate == 2) ? (cell & 1023) : (q.copy_base + cell);   // ablation 2: all atomics hit 64 KiB of L2
        q.ox[e] = ox; q.oy[e] = oy;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef ZDR_MEASURE_STATS
    if (q.count > 0) {
        bool dup = false;
        if (lane < q.count) for (int k = 0; k < lane; k++) dup = dup || (q.cell[k] == q.cell[lane]);
        q.st_flushes++; q.st_entries += (unsigned long long)q.count; q.st_dups += (unsigned long long)__popcll(__ballot(dup));
    }
#endif
    for (int base = 0; base < q.count; base += 4) {
        int e = base + sub;
        if (e < q.count) {
            int cell = q.cell[e];
            float gc = q.g[4 * e + (j & 3)];
            float ox = q.ox[e], oy = q.oy[e];
            float wx = (j & 8) ? ox : 1.0f - ox;        // corner m = j >> 2: bit 1 -> x + 1, bit 0 -> y + 1
            float wy = (j & 4) ? oy : 1.0f - oy;       // (per-lane constants a, b with w = fma(o, a, b) save two VALU per iteration and cost four VGPRs the kernel does not have: four more spills, +0.8 ms — measured, round 4)
            const float add = (wx * wy) * gc;                                  // k_m * dmat.c, interaction.py:82-89
            if (ablate == 6) asm volatile("" ::"v"(add), "v"(cell));          // ablation 6: the whole queue and flush, but no atomic is issued
            else if (q.small) {                                                // the whole cell array within 4 GiB (textures up to 8190^2): scalar base + 32-bit offset,
                const uint32_t idx = 16u * (uint32_t)cell + (uint32_t)j;       // no 64-bit shift and add per lane and request
                __builtin_assume(idx < (1u << 30));
                unsafeAtomicAdd(cells + idx, add);
            } else unsafeAtomicAdd(cells + 16 * (size_t)cell + j, add);
        }
    }
    __builtin_amdgcn_wave_barrier();
    q.count = 0;
}

// end of the kernel: whatever is still queued, and the LDS cell array if the wave kept one
ZD void scatter_finish(ScatterQueue &q, float *__restrict__ cells, int tex_h, int tex_w, int ablate) {
    scatter_flush(q, cells, tex_h, tex_w, ablate);
    if (q.lds_cells) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = threadIdx.x & 63; i < 16 * q.ncells; i += 64) {
            const float v = q.lds_cells[i];
            if (v != 0.0f) unsafeAtomicAdd(cells + 16 * (size_t)q.copy_base + i, v);
        }
    }
}

// must be called by the whole wave; lanes with active == false push nothing
ZD void scatter_push(ScatterQueue &q, float *__restrict__ cells, bool active, f2 uv, float4 g, int tex_h, int tex_w, int ablate) {
    if (ablate == 1) { asm volatile("" ::"v"(g.x), "v"(g.y), "v"(g.z), "v"(g.w), "v"(uv.x), "v"(uv.y)); return; }
    unsigned long long mask = __ballot(active);
    int n = __popcll(mask);
    if (n == 0) return;
    if (q.lds_cells) {                   // few texels: the cell array is in LDS, 16 ds_add_f32 per vertex
        if (active) {
            float ox, oy;
            const int cell = scatter_cell(uv, tex_h, tex_w, ox, oy);
            float *c = q.lds_cells + 16 * cell;
            const float k00 = (1.0f - ox) * (1.0f - oy), k01 = (1.0f - ox) * oy, k10 = ox * (1.0f - oy), k11 = ox * oy;   // corner m = 2 dx + dy
            const float gg[4] = {g.x, g.y, g.z, g.w}, kk[4] = {k00, k01, k10, k11};
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int ch = 0; ch < 4; ch++) __hip_atomic_fetch_add(c + 4 * m + ch, kk[m] * gg[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        return;
    }
    if (q.count + n > ZDR_SCATTER_CAP) scatter_flush(q, cells, tex_h, tex_w, ablate);
    if (active) {
        int slot = q.count + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        *(float4 *)(q.g + 4 * slot) = g;
        q.ox[slot] = uv.x; q.oy[slot] = uv.y;      // raw uv: the flush turns it into cell and offsets
    }
    q.count += n;
    if (q.count >= ZDR_SCATTER_FLUSH_AT) scatter_flush(q, cells, tex_h, tex_w, ablate);
}

// ------------------------------------------------------------------------------------ lights
struct LightSample { f3 wi; float dist, pdf; f3 eval; };       // light.py:11

ZD f3 sample_uniform_triangle(f2 u) {                           // light.py:16-20
    f2 uv;
    if (u.x < u.y) { uv.x = 0.5f * u.x; uv.y = -0.5f * u.x + u.y; }
    else { uv.x = -0.5f * u.y + u.x; uv.y = 0.5f * u.y; }
    return mk3(uv.x, uv.y, 1.0f - uv.x - uv.y);
}

// pdf = d^2 / (n T area cos_light)  (light.py:69-73, 105-110); ng and area are precomputed per slot
ZD float light_pdf(f3 origin, f3 p, f3 ln, float area, int n_times_T, f3 &wi, float &cos_light, float &sqr_dist) {
    f3 dp = p - origin;
    wi = normalize(dp);
    cos_light = -dot(ln, wi);
    sqr_dist = dot(dp, dp);
    return sqr_dist * rcp((float)n_times_T * area * cos_light);
}

// ------------------------------------------------------------------------ environment light
// heap.texture2d_sample(23332, uv): bilinear between texel centres, clamp to edge (unpinned, see zdr_amd/envmap.py)
ZD f3 env_lookup(const DScene &S, f2 uv) {
    float x = uv.x * (float)S.env_w - 0.5f, y = uv.y * (float)S.env_h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    int x0 = clampi((int)x0f, 0, S.env_w - 1), x1 = clampi((int)x0f + 1, 0, S.env_w - 1);
    int y0 = clampi((int)y0f, 0, S.env_h - 1), y1 = clampi((int)y0f + 1, 0, S.env_h - 1);
    float4 c00 = S.env_tex[(size_t)y0 * S.env_w + x0], c10 = S.env_tex[(size_t)y0 * S.env_w + x1];
    float4 c01 = S.env_tex[(size_t)y1 * S.env_w + x0], c11 = S.env_tex[(size_t)y1 * S.env_w + x1];
    f3 top = xyz(c00) + (xyz(c10) - xyz(c00)) * fx, bot = xyz(c01) + (xyz(c11) - xyz(c01)) * fx;
    return top + (bot - top) * fy;
}
ZD f3 uv_to_direction(f2 uv) {                                   // envmap.py:206-214
    float phi = 2.0f * ZDR_PI * (1.0f - uv.x), theta = ZDR_PI * uv.y;
    float st = sinf(theta);
    return normalize(mk3(sinf(phi) * st, cosf(theta), cosf(phi) * st));
}
ZD f2 direction_to_uv(f3 d) {                                    // envmap.py:216-220
    f2 uv; uv.x = 1.0f - atan2f(d.x, d.z) * (1.0f / (2.0f * ZDR_PI)); uv.y = acosf(d.y) * ZDR_INV_PI;
    return uv;
}
ZD void sample_alias_table(const DScene &S, int n, float u_in, int offset, int &index, float &uu) {   // envmap.py:85-106
    float u = u_in * (float)n;
    int i = clampi((int)u, 0, n - 1);
    float ur = u - floorf(u);
    float prob = S.alias_prob[i + offset];
    if (ur < prob) { index = i; uu = ur * rcp(prob); }
    else { index = S.alias_idx[i + offset]; uu = (ur - prob) * rcp(1.0f - prob); }
}
ZD float env_pdf_scale(float v, int n) {                         // 1 / (sin(pi v) 2 pi^2 n); App. B-9: 1/n added
    float sn = sinf(ZDR_PI * v);
    float inv_s = (sn > 0.0f) ? rcp(sn) : 0.0f;
    return inv_s * rcp(2.0f * ZDR_PI * ZDR_PI * (float)n);
}
ZD float env_sampled_light_pdf(const DScene &S, f3 dir, int n) {  // envmap.py:240-248
    f2 uv = direction_to_uv(dir);
    int index = clampi((int)(uv.y * (float)S.map_h), 0, S.map_h - 1) * S.map_w + clampi((int)(uv.x * (float)S.map_w), 0, S.map_w - 1);
    return S.env_pdf[index] * env_pdf_scale(uv.y, n);
}

// sample_light (light.py:23-81): u_pick = next() was drawn by the caller; the environment branch then
// draws only next2f(), the mesh branch next() and next2f() (light.py:29-31 vs 50-63).
template <bool ENV, class NEXT1, class NEXT2>
ZD LightSample sample_light(const DScene &S, f3 origin, float u_pick, NEXT1 next1, NEXT2 next2) {
    LightSample L;
    int n = (ENV ? S.env_count : 0) + S.light_count;
    if (n <= 0) {  // the reference would index out of bounds; consume the mesh branch's dimensions, contribute nothing
        (void)next1(); (void)next2();
        L.wi = mk3(0.0f, 0.0f, 1.0f); L.dist = 0.0f; L.pdf = 1.0f; L.eval = mk3(0.0f);
        return L;
    }
    int idx = clampi((int)(u_pick * (float)n), 0, n - 1);
    if (ENV && idx < S.env_count) {                              // envmap.py:223-238
        f2 u = next2();
        int iy, ix; float uy, ux;
        sample_alias_table(S, S.map_h, u.y, 0, iy, uy);
        sample_alias_table(S, S.map_w, u.x, S.map_h + iy * S.map_w, ix, ux);
        f2 uv; uv.x = ((float)ix + ux) * rcp((float)S.map_w); uv.y = ((float)iy + uy) * rcp((float)S.map_h);
        L.wi = uv_to_direction(uv); L.dist = 1e30f;
        L.pdf = S.env_pdf[iy * S.map_w + ix] * env_pdf_scale(uv.y, n);
        L.eval = env_lookup(S, uv);
        return L;
    }
    if (ENV) idx -= S.env_count;
    float u_prim = next1();
    f2 u_pt = next2();
    // light.py:33-48 walks light -> instance -> triangle range -> triangle -> emission; the flat table makes that one
    // lookup (none for a single light) + the entry
    int base = 0, T = S.light0_T;
    if (S.light_count > 1) { base = S.light_range[2 * idx]; T = S.light_range[2 * idx + 1]; }
    int prim = clampi((int)(u_prim * (float)T), 0, T - 1);
    const float4 *r = S.light_tris + 5 * (size_t)(base + prim);
    float4 r0 = r[0], r1 = r[1], r2 = r[2], r6 = r[3], r4 = r[4];
    f3 abc = sample_uniform_triangle(u_pt);
    f3 p = xyz(r0) * abc.x + xyz(r1) * abc.y + xyz(r2) * abc.z;
    float cos_light, sqr_dist;
    L.pdf = light_pdf(origin, p, xyz(r6), r6.w, n * T, L.wi, cos_light, sqr_dist);
    L.dist = 0.9999f * fsqrt(sqr_dist);
    L.eval = (cos_light > 1e-4f) ? xyz(r4) : mk3(0.0f);
    return L;
}

// sample_light_pdf (light.py:84-111): pdf of having light-sampled point p on (inst, slot)
template <bool ENV>
ZD float sample_light_pdf(const DScene &S, f3 origin, int inst, int slot, f3 p) {
    float4 r6 = S.shade[8 * (size_t)slot + 6];
    float area = S.shade[8 * (size_t)slot + 7].x;
    int T = S.inst_tri_begin[inst + 1] - S.inst_tri_begin[inst];
    f3 wi; float c, d2;
    return light_pdf(origin, p, xyz(r6), area, ((ENV ? S.env_count : 0) + S.light_count) * T, wi, c, d2);
}

ZD float balanced_heuristic(float a, float b) { return a * rcp(fmaxf(a + b, 1e-4f)); }   // prb.py:12-13

// LuisaCompute offset_ray_origin (Waechter & Binder, RT Gems ch.6) — prb.py:75, direct.py:64
ZD float offset1(float p, float n) {
    int of_i = (int)(256.0f * n);
    float p_i = __int_as_float(__float_as_int(p) + ((p < 0.0f) ? -of_i : of_i));
    return (fabsf(p) < (1.0f / 32.0f)) ? p + (1.0f / 65536.0f) * n : p_i;
}
ZD f3 offset_ray_origin(f3 p, f3 n) { return mk3(offset1(p.x, n.x), offset1(p.y, n.y), offset1(p.z, n.z)); }

// --------------------------------------------------------------------------------------- onb
struct Onb { f3 tangent, binormal, normal; };                   // onb.py:7
ZD Onb make_onb(f3 n) {                                         // onb.py:21-28
    Onb o;
    o.binormal = normalize((fabsf(n.x) > fabsf(n.z)) ? mk3(-n.y, n.x, 0.0f) : mk3(0.0f, -n.z, n.y));
    o.tangent = normalize(cross(o.binormal, n));
    o.normal = n;
    return o;
}
ZD f3 to_world(const Onb &o, f3 v) { return v.x * o.tangent + v.y * o.binormal + v.z * o.normal; }
ZD f3 to_local(const Onb &o, f3 v) { return mk3(dot(v, o.tangent), dot(v, o.binormal), dot(v, o.normal)); }
