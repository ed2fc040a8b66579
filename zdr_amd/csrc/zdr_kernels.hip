// zdr_kernels.hip — gfx950 kernels of the zdr hot path and their launchers.
//
// Work item = one 8x8 pixel tile x one chunk of the sample range; workgroups are single 64-lane waves, so
// no workgroup barrier exists anywhere.  The direct / collocated / uvgrad kernels run one workgroup per item
// (lane = pixel), the block index remapped so that each XCD (blocks b, b+8, ... share one) receives a contiguous
// run of tiles: primary-hit texel footprints of neighbouring tiles then share that XCD's L2.  The path kernels are
// persistent waves that draw items from per-XCD counters and treat lanes as workers (see k_path).
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include "integrators.h"
#include "zdr.h"

#define WAVE 64
// __launch_bounds__ second argument = minimum waves per SIMD (caps the VGPR budget at 512 / n).
// Measured on cbox 512^2 spp 256 (profiles/r1_ab_flags.txt): 3 (<= 168 VGPRs, no spills) is best for both.

struct WorkItem { int x, y, pix; bool valid; uint32_t s_begin, s_end; int chunk, tile, tile_local; };   // tile: number within the rectangle; tile_local: within this call's shard

// Work item `logical` = (tile, sample chunk); consecutive items are the chunks of one tile, then the next tile.
ZD WorkItem decode_item(const RenderCfg &R, int logical) {
    const int nblocks = R.ntiles * R.nchunks;
    WorkItem w;
    w.valid = logical >= 0 && logical < nblocks;
    const int tl = w.valid ? logical / R.nchunks : 0;
    w.chunk = w.valid ? logical - tl * R.nchunks : 0;
    const int number = tl * R.shard_count + R.shard_index;   // interleaved tile shard (zdr.h): tile numbers index, index + count, ...
    const int ty = number / R.tiles_x;
    int tx = number - ty * R.tiles_x + ty * R.shard_skew;    // row ty is numbered from column ty * skew on: a shard's tiles run along diagonals
    tx -= (tx / R.tiles_x) * R.tiles_x;
    w.tile = ty * R.tiles_x + tx; w.tile_local = tl;         // .tile: position in the tile grid (tile masks), .tile_local: position in the shard's workspace
    const int lane = threadIdx.x;
    w.x = R.x0 + tx * 8 + (lane & 7);
    w.y = R.y0 + ty * 8 + (lane >> 3);
    w.valid = w.valid && (w.x < R.x1) && (w.y < R.y1);
    w.pix = w.x + w.y * R.width;
    w.s_begin = R.sample_begin + (uint32_t)w.chunk * R.chunk;
    uint32_t e = w.s_begin + R.chunk;
    w.s_end = (e < R.sample_end) ? e : R.sample_end;
    return w;   // the sample range is wave-uniform; lanes outside the shard are masked with w.valid
}

// one workgroup per item (direct / collocated / uvgrad kernels): XCD-contiguous tile runs
ZD WorkItem decode_block(const RenderCfg &R) {
    const int nblocks = R.ntiles * R.nchunks;
    const int per_xcd = (nblocks + 7) >> 3;
    const int b = blockIdx.x;
    const int logical = (b & 7) * per_xcd + (b >> 3);
    return decode_item(R, logical < nblocks ? logical : -1);
}

// Persistent waves (path kernels) draw items from eight counters, one per XCD: a wave on XCD x
// (workgroups are dealt round-robin, x = blockIdx.x & 7) walks x's contiguous run of tiles, so the texel
// footprints of neighbouring tiles share that XCD's L2; when its own run is used up it takes from the others.
// Wave-uniform; returns -1 when no item is left.  Counters only grow, by at most 8 per call.
ZD int fetch_item(const RenderCfg &R, unsigned int *counters) {
    const int nblocks = R.ntiles * R.nchunks;
    const int per_xcd = (nblocks + 7) >> 3;
    int got = -1;
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7;
        for (int j = 0; j < 8 && got < 0; j++) {
            const int y = (x + j) & 7;
            const unsigned int local = atomicAdd(&counters[y], 1u);
            const long long logical = (long long)y * per_xcd + local;
            if (local < (unsigned int)per_xcd && logical < nblocks) got = (int)logical;
        }
    }
    return __builtin_amdgcn_readfirstlane(got);
}

// candidate triangle pairs of this wave's camera rays (BruteAccel::closest_camera): the tile's mask, or every pair
ZD unsigned long long camera_mask(const DScene &S, const KernelIO &io, const WorkItem &w) {
    const int npairs = (S.nquads + 1) >> 1;
    const unsigned long long all = (npairs >= 64) ? ~0ull : ((1ull << npairs) - 1ull);
    return io.tile_masks ? io.tile_masks[w.tile] : all;
}

ZD void store_pixel(const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, const WorkItem &w, f3 sum) {
    if (!w.valid) return;
    if (R.nchunks == 1) {   // integrator.py:29: (s / spp, 1)
        float fs = (float)C.spp;
        io.image[w.pix] = make_float4(__fdiv_rn(sum.x, fs), __fdiv_rn(sum.y, fs), __fdiv_rn(sum.z, fs), R.alpha);
    } else {
        // chunk partials: [chunk][tile of this shard][lane] — a wave writes one contiguous 1 KiB line
        io.partial[((size_t)w.chunk * R.ntiles + w.tile_local) * WAVE + threadIdx.x] = make_float4(sum.x, sum.y, sum.z, 0.0f);
    }
}

ZD f3 load_le_grad(const SamplerCfg &C, const KernelIO &io, const WorkItem &w) {   // integrator.py:38-40
    f3 g = mk3(0.0f);
    if (w.valid) {
        float4 gi = io.d_image[w.pix];
        float fs = (float)C.spp;
        g = mk3(__fdiv_rn(gi.x, fs), __fdiv_rn(gi.y, fs), __fdiv_rn(gi.z, fs));
        if (any_nan(g)) g = mk3(0.0f);
    }
    return g;
}

template <bool STATS>
ZD void flush_counters(const KernelIO &io, const Counters &cnt) {
    if (!STATS) return;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        unsigned long long v = cnt.c[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
        if (threadIdx.x == 0 && v) atomicAdd(io.counters + i, v);
    }
}

// --------------------------------------------------------------------------- tile masks
// One wave per 8x8 tile, lane = pair of primitives of the brute-force walk: bit k of the tile's mask is set unless ALL triangles
// of pair k lie entirely outside one side plane of the tile's camera frustum (or behind the camera).
// The frustum is padded by one pixel: the tent filter (camera.py:20-31) moves a sample up to half a
// pixel outside its pixel.  Conservative by construction — a set bit only costs a test.
__global__ __launch_bounds__(WAVE) void k_tile_masks(DScene S, RenderCfg R, unsigned long long *masks) {
    const int tile = blockIdx.x, ty = tile / R.tiles_x, tx = tile - ty * R.tiles_x, lane = threadIdx.x;
    const float X0 = (float)(R.x0 + tx * 8) - 1.0f, X1 = (float)(R.x0 + tx * 8 + 8) + 1.0f;
    const float Y0 = (float)(R.y0 + ty * 8) - 1.0f, Y1 = (float)(R.y0 + ty * 8 + 8) + 1.0f;
    const f3 right = ld3(R.cam_right), upp = ld3(R.cam_upp), fwd = ld3(R.cam_fwd), co = ld3(R.cam_o);
    auto dir = [&](float X, float Y) {                      // pixel_ray without the normalisation
        float px = (R.two_over_w * X - 1.0f) * R.cam_tan, py = ((R.two_over_h * Y - 1.0f) * R.aspect) * R.cam_tan;
        return (right * px - upp * py) + fwd;
    };
    const f3 c00 = dir(X0, Y0), c10 = dir(X1, Y0), c01 = dir(X0, Y1), c11 = dir(X1, Y1);
    f3 n[4] = { cross(c00, c01), cross(c11, c10), cross(c10, c00), cross(c01, c11) };   // left, right, top, bottom
    const f3 inside = (c00 + c11) + (c10 + c01);
#pragma unroll
    for (int k = 0; k < 4; k++) if (dot(n[k], inside) < 0.0f) n[k] = n[k] * -1.0f;
    const int npairs = (S.nquads + 1) >> 1;
    bool keep = false;
    if (lane < npairs) {
        // the (up to four) triangles of primitives 2 lane and 2 lane + 1: quads cover slots 2q, 2q + 1, single triangles follow (accel.h)
        const int q0 = 2 * lane, q1 = 2 * lane + 1;
        const int b0 = q0 < S.nquads2 ? 2 * q0 : q0 + S.nquads2, n0 = q0 < S.nquads2 ? 2 : 1;
        const int b1 = q1 < S.nquads2 ? 2 * q1 : q1 + S.nquads2, n1 = q1 >= S.nquads ? 0 : (q1 < S.nquads2 ? 2 : 1);
        for (int i = 0; i < n0 + n1; i++) {
            const int t = i < n0 ? b0 + i : b1 + (i - n0);
            const float4 *r = S.shade + 8 * (size_t)t;
            f3 v0 = xyz(r[0]) - co, v1 = xyz(r[1]) - co, v2 = xyz(r[2]) - co;
            float vm = fmaxf(fmaxf(length(v0), length(v1)), length(v2));
            bool culled = (dot(v0, fwd) <= 0.0f) && (dot(v1, fwd) <= 0.0f) && (dot(v2, fwd) <= 0.0f);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float e = -1e-4f * length(n[k]) * vm;
                culled = culled || ((dot(n[k], v0) < e) && (dot(n[k], v1) < e) && (dot(n[k], v2) < e));
            }
            keep = keep || !culled;
        }
    }
    unsigned long long m = __ballot(keep);
    if (lane == 0) masks[tile] = m;
}

// ------------------------------------------------------------------------------------- path
// Forward (and the counting variant).  PERSISTENT waves: the grid is what the chip holds at once, and each
// wave draws work items (tile x sample chunk) until none is left.  Two alternating phases:
//   refill  lane = pixel: ZDR_RING_BATCH camera samples per pixel of the current item are generated, traced
//           and classified by the whole wave and the vertices to shade are parked in the wave's FIFO;
//   flat loop  a lane is just a worker: it takes the next parked vertex (of ANY pixel) and each trip
//           shades one vertex:  shade (light sample, BSDF sample, the vertex's two rays) -> classify the new hit.
// Items overlap: when the current item has no camera sample left the wave starts the next one while the last
// paths of the old one are still running, so lanes never wait for a tile's longest path.  Per-pixel state
// (radiance sums, CMJ seeds) therefore lives in LDS in two banks, item n in bank n & 1; a path carries its
// bank and pixel.  Radiance is added with ds_add_f32 — one wave, program order, so sums are reproducible.
struct ItemBanks {
    int logical[2];          // item held by each bank, -1 = free (wave-uniform)
    uint32_t inflight[2];    // parked + running paths of each bank (wave-uniform)
};

// The persistent kernels read their launch constants (scene pointers, camera, sampler grid ...) through a pointer to the kernarg
// segment that is laundered once per trip, so a constant is s_load-ed again where a trip uses it instead of being kept for the whole
// kernel in an SGPR that is spilled to a VGPR lane (v_readlane / v_writelane / s_nop on the VALU port: 320 reloads in
// k_path<cmj, brute> before; profiles/r3_bwd_records_and_atomics.txt).  PathKArgs must be the kernels' parameter list, in order:
// every kernel that uses the macros is declared through ZDR_PATH_KERNEL_PARAMS, and the layout rules the kernarg segment follows
// (each by-value struct at its natural alignment, at most 8 here) are asserted below.
struct PathKArgs { DScene S; RenderCfg R; SamplerCfg C; KernelIO io; };
#define ZDR_PATH_KERNEL_PARAMS DScene S_, RenderCfg R_, SamplerCfg C_, KernelIO io_
static constexpr size_t zdr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static_assert(std::is_trivially_copyable<PathKArgs>::value && std::is_standard_layout<PathKArgs>::value, "kernel arguments are copied bytewise");
static_assert(offsetof(PathKArgs, S) == 0, "first kernel argument at offset 0 of the kernarg segment");
static_assert(offsetof(PathKArgs, R) == zdr_align_up(sizeof(DScene), alignof(RenderCfg)), "PathKArgs does not mirror the kernarg packing");
static_assert(offsetof(PathKArgs, C) == zdr_align_up(offsetof(PathKArgs, R) + sizeof(RenderCfg), alignof(SamplerCfg)), "PathKArgs does not mirror the kernarg packing");
static_assert(offsetof(PathKArgs, io) == zdr_align_up(offsetof(PathKArgs, C) + sizeof(SamplerCfg), alignof(KernelIO)), "PathKArgs does not mirror the kernarg packing");
static_assert(alignof(DScene) <= 8 && alignof(RenderCfg) <= 8 && alignof(SamplerCfg) <= 8 && alignof(KernelIO) <= 8, "an over-aligned member would change the kernarg packing");
typedef __attribute__((address_space(4))) const char karg_byte_t;
ZD const PathKArgs &kargs_fresh() {
    karg_byte_t *p = (karg_byte_t *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const PathKArgs *)p;
}
#define ZDR_KARGS_BEGIN const PathKArgs *ka = &kargs_fresh();
#define ZDR_KARGS_REFRESH ka = &kargs_fresh();
// Timing-only ablations of the backward kernels exist in measurement builds alone (-DZDR_MEASURE, tools/): the product reads no
// ZDR_DEBUG_NO_SCATTER and every `ZDR_ABLATE == n` below folds to false.
#ifdef ZDR_MEASURE
#define ZDR_ABLATE (R.debug_no_scatter)
#else
#define ZDR_ABLATE 0
#endif

template <int SK, class A, bool STATS, bool ENV>
__global__ __launch_bounds__(WAVE, ENV ? A::kMinWavesFwdEnv : A::kMinWavesFwd) void k_path(ZDR_PATH_KERNEL_PARAMS) {
    ZDR_KARGS_BEGIN
#define S (ka->S)
#define R (ka->R)
#define C (ka->C)
#define io (ka->io)
    extern __shared__ int lds[];        // BvhAccel: traversal stacks (sized at launch); unused otherwise
    __shared__ uint32_t lds_perm[2 * WAVE];
    __shared__ int lds_origin[4];
    __shared__ float lds_sum[2 * 3 * WAVE];
    const int lane = threadIdx.x;
    Counters cnt;
#pragma unroll
    for (int i = 0; i < 8; i++) cnt.c[i] = 0;
    PathVertex pv; f3 term_Li = mk3(0.0f);
    ItemBanks ib; ib.logical[0] = ib.logical[1] = -1; ib.inflight[0] = ib.inflight[1] = 0;
    int bank = 1;                                           // bank of the current item (the first fetch flips it to 0)
    bool more_items = true;
    WorkItem w = decode_item(R, -1);
    uint32_t next_sample = 0, s_end = 0, perm_seed = 0;
    unsigned long long cam_mask = 0ull;
    f3 sum = mk3(0.0f);                                     // current item, this lane's pixel: paths that ended at the camera ray
    PrimaryQueue q = queue_init(io);
    bool alive = false; int pix = 0;                        // pix: bank * 64 + pixel of the running path
    PathState ps; Interaction it;
    ps.o = mk3(0.0f); ps.d = mk3(0.0f, 0.0f, 1.0f); ps.beta = mk3(1.0f); ps.L = mk3(0.0f); ps.pdf_bsdf = 1e30f; ps.depth = 0;
    ps.smp = sampler_make<SK>(C, 0, 0, 0, 0);
    it.p = mk3(0.0f); it.uv.x = 0.0f; it.uv.y = 0.0f; it.ns = mk3(0.0f, 0.0f, 1.0f); it.ng = it.ns; it.inst = 0; it.prim = 0;
    int stall = 0;
    for (;;) {
        ZDR_KARGS_REFRESH
        bool progress = false;
        if (q.tail - q.head < (uint32_t)__popcll(__ballot(!alive))) {          // the FIFO cannot serve every idle lane
            if (next_sample < s_end) {
                progress = true;
                const uint32_t t0 = q.tail;
                primary_refill<SK, A, false, STATS, ENV>(S, R, C, lds, w.x, w.y, w.valid, cam_mask, perm_seed, bank, next_sample, s_end, q, sum, cnt);
                ib.inflight[bank] += q.tail - t0;
            } else if (more_items && ib.logical[bank ^ 1] < 0) {                // next item, into the free bank
                if (ib.logical[bank] >= 0) {                                    // park the register part of the old item's sums
                    atomicAdd(&lds_sum[(bank * 3 + 0) * WAVE + lane], sum.x); atomicAdd(&lds_sum[(bank * 3 + 1) * WAVE + lane], sum.y);
                    atomicAdd(&lds_sum[(bank * 3 + 2) * WAVE + lane], sum.z);
                    sum = mk3(0.0f);
                }
                const int nxt = fetch_item(R, io.work_counters);
                if (nxt < 0) more_items = false;
                else {
                    bank ^= 1;
                    ib.logical[bank] = nxt;
                    w = decode_item(R, nxt);
                    perm_seed = (SK == 0) ? xxhash32_4((uint32_t)w.x, (uint32_t)w.y, C.seed, 0u) : 0u;
                    cam_mask = camera_mask(S, io, w);
                    lds_perm[bank * WAVE + lane] = perm_seed;
                    if (lane == 0) { lds_origin[bank * 2] = w.x; lds_origin[bank * 2 + 1] = w.y; }
                    lds_sum[(bank * 3 + 0) * WAVE + lane] = 0.0f; lds_sum[(bank * 3 + 1) * WAVE + lane] = 0.0f; lds_sum[(bank * 3 + 2) * WAVE + lane] = 0.0f;
                    __syncthreads();
                    next_sample = w.s_begin; s_end = w.s_end;
                }
                stall = 0;
                continue;
            }
        }
        const int took = primary_pop<SK>(S, C, !alive, lds_perm, lds_origin, q, ps, it);
        if (took >= 0) { alive = true; pix = took; }
        if (__ballot(alive) != 0ull) {
            progress = true;
            bool done = false;
            if (alive) {
                Hit h;
                done = path_shade<SK, A, false, STATS, ENV>(S, R, C, io, lds, ps, it, pv, h, cnt);
                if (!done) { path_continue<A, STATS>(S, lds, ps, h, cnt); done = path_arrive<false, STATS, ENV>(S, ps, h, it, term_Li, cnt); }
                if (done) {
                    alive = false;
                    if (!any_nan(ps.L)) {                   // integrator.py:27-28
                        f3 c = clamp_radiance(ps.L);
                        const int bk = pix >> 6, px = pix & 63;
                        atomicAdd(&lds_sum[(bk * 3 + 0) * WAVE + px], c.x); atomicAdd(&lds_sum[(bk * 3 + 1) * WAVE + px], c.y);
                        atomicAdd(&lds_sum[(bk * 3 + 2) * WAVE + px], c.z);
                    } else COUNT(C_NAN);
                }
            }
            ib.inflight[0] -= (uint32_t)__popcll(__ballot(done && (pix >> 6) == 0));
            ib.inflight[1] -= (uint32_t)__popcll(__ballot(done && (pix >> 6) == 1));
        }
        // retire finished items: every camera sample generated and no path left
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if (ib.logical[b] >= 0 && ib.inflight[b] == 0 && (b != bank || next_sample >= s_end)) {
                __syncthreads();
                const WorkItem wb = decode_item(R, ib.logical[b]);
                f3 tot = mk3(lds_sum[(b * 3 + 0) * WAVE + lane], lds_sum[(b * 3 + 1) * WAVE + lane], lds_sum[(b * 3 + 2) * WAVE + lane]);
                if (b == bank) { tot = tot + sum; sum = mk3(0.0f); }
                if (!STATS) store_pixel(R, C, io, wb, tot);         // the stats variant owns no image
                ib.logical[b] = -1;
            }
        }
        if (__ballot(alive) == 0ull && q.tail == q.head && next_sample >= s_end && !more_items) break;   // both banks are retired by now
        stall = progress ? 0 : stall + 1;
        if (stall > 4) { raise_device_error(S, ZDR_DEVERR_STALL); break; }   // cannot happen (every branch above makes progress); never spin on the GPU, never end silently
    }
    flush_counters<STATS>(io, cnt);
#undef S
#undef R
#undef C
#undef io
}

// PRB backward with ONE traversal.  Each trip a live lane shades one vertex of its path (same trip
// order, primary queue, pixel-free lanes, persistent waves and item banks as k_path) and appends it to its record list.  The
// records (5 float4 + a link to the path's previous one) live in a per-wave LDS POOL, see below; the few that find no slot go to
// per-lane scratch.  When a path ends, a short wave-uniform loop sweeps its records last to first and queues the
// gradients; all queue traffic happens at reconverged points so the whole wave takes part in a flush.
// Keeping the records out of scratch is what matters: 2.3 KB of scratch per lane thrashed L2 (113 GB of
// fabric traffic per launch, 13 of 37 ms, round 1); the last 17 % cost 1.4 ms of 13.6 (profiles/r3_bwd_records_and_atomics.txt).
// Neither the CMJ seeds nor the pixel cotangents of the two item banks are kept in LDS (a popped path hashes its seed again and
// loads its cotangent from the image): that leaves room for the pool in 8 of gfx950's 1,280-byte LDS blocks, 16 waves per CU.
// (The lane-owned record rows, the 12-wave layout with seeds and cotangents in LDS and the LDS padding experiment that preceded
// this are kept as profiles/r4_pruned_experiment_branches.patch.)
typedef unsigned int zdr_u4 __attribute__((ext_vector_type(4)));
template <int SK, class A, bool ENV>
__global__ __launch_bounds__(WAVE, A::kMinWavesBwd) void k_path_bwd(ZDR_PATH_KERNEL_PARAMS) {
    ZDR_KARGS_BEGIN
#define S (ka->S)
#define R (ka->R)
#define C (ka->C)
#define io (ka->io)
    extern __shared__ int lds[];        // BvhAccel: traversal stacks (sized at launch); unused otherwise
    __shared__ __attribute__((aligned(16))) float lds_q[ZDR_SCATTER_LDS_FLOATS];   // the queue writes g as one float4
    // The records of all paths of the wave share ONE pool of NS slots (80 bytes + a link each): a lane takes its home slot
    // (slot == lane) when that is free and otherwise the lowest free one, and gives the slot back when the sweep has read it.
    constexpr int NS = A::kPoolSlots;
    static_assert(NS >= 16 && NS <= 128, "the free mask is two 64-bit words");
    __shared__ float4 lds_pool[5 * NS];                     // [float4 f][slot]
    __shared__ unsigned char lds_link[NS];                  // slot of the path's previous record (255: in scratch; a path's first record links to nothing and its link is never followed).  One byte: three more slots fit
    // bit s set: slot s is free.  The only state lanes share: lane 0 stores the mask after an allocation, the sweep's lanes OR freed
    // slots in, every lane reads it before the next allocation.  All three accesses are volatile or atomic and stand between
    // wavefront-scope fences, so the protocol does not rest on what the optimiser happens to do with plain LDS accesses.
    __shared__ __attribute__((aligned(16))) unsigned int lds_free[4];
    __shared__ int lds_origin[4];                           // first pixel of each item bank's tile
    const int lane = threadIdx.x;
    constexpr unsigned long long all_lo = (NS >= 64) ? ~0ull : ((1ull << (NS & 63)) - 1ull);
    constexpr unsigned long long all_hi = (NS > 64) ? ((NS >= 128) ? ~0ull : ((1ull << ((NS - 64) & 63)) - 1ull)) : 0ull;
    if (lane < 4) {
        const unsigned long long w = (lane < 2) ? all_lo : all_hi;
        lds_free[lane] = (unsigned int)((lane & 1) ? (w >> 32) : w);
    }
    __syncthreads();
    int last = -1;                                          // where the running path's most recent record lives: slot, 255 = scratch, -1 = none yet
    int deep_link[ZDR_MAX_RECORDED_DEPTH];
    Counters cnt;
    ItemBanks ib; ib.logical[0] = ib.logical[1] = -1; ib.inflight[0] = ib.inflight[1] = 0;
    int bank = 1;
    bool more_items = true;
    WorkItem w = decode_item(R, -1);
    uint32_t next_sample = 0, s_end = 0, perm_seed = 0;
    unsigned long long cam_mask = 0ull;
    f3 le_grad = mk3(0.0f);                                 // cotangent of the running path's pixel
    ScatterQueue q = scatter_queue_init(lds_q, R.tex_h, R.tex_w, R.cell_copies);
    PackedVertex deep[ZDR_MAX_RECORDED_DEPTH];
    int nrec = 0;
    PrimaryQueue pq = queue_init(io);
    f3 unused_sum = mk3(0.0f);
    bool alive = false; int pix = 0;
    PathState ps; Interaction it;
    ps.o = mk3(0.0f); ps.d = mk3(0.0f, 0.0f, 1.0f); ps.beta = mk3(1.0f); ps.L = mk3(0.0f); ps.pdf_bsdf = 1e30f; ps.depth = 0;
    ps.smp = sampler_make<SK>(C, 0, 0, 0, 0);
    it.p = mk3(0.0f); it.uv.x = 0.0f; it.uv.y = 0.0f; it.ns = mk3(0.0f, 0.0f, 1.0f); it.ng = it.ns; it.inst = 0; it.prim = 0;
    int stall = 0;
#ifdef ZDR_MEASURE_STATS
    unsigned long long st_trips = 0, st_shaded = 0, st_fin = 0, st_iters = 0, st_steps = 0;
#endif
    for (;;) {
        ZDR_KARGS_REFRESH
        bool progress = false;
        if (pq.tail - pq.head < (uint32_t)__popcll(__ballot(!alive))) {
            if (next_sample < s_end) {
                const uint32_t t0 = pq.tail;
                primary_refill<SK, A, true, false, ENV>(S, R, C, lds, w.x, w.y, w.valid, cam_mask, perm_seed, bank, next_sample, s_end, pq, unused_sum, cnt);
                ib.inflight[bank] += pq.tail - t0;
                progress = true;
            } else if (more_items && ib.logical[bank ^ 1] < 0) {
                const int nxt = fetch_item(R, io.work_counters);
                if (nxt < 0) more_items = false;
                else {
                    bank ^= 1;
                    ib.logical[bank] = nxt;
                    w = decode_item(R, nxt);
                    perm_seed = (SK == 0) ? xxhash32_4((uint32_t)w.x, (uint32_t)w.y, C.seed, 0u) : 0u;
                    cam_mask = camera_mask(S, io, w);
                    if (lane == 0) { lds_origin[bank * 2] = w.x; lds_origin[bank * 2 + 1] = w.y; }
                    __syncthreads();
                    next_sample = w.s_begin; s_end = w.s_end;
                }
                stall = 0;
                continue;
            }
        }
        const int took = primary_pop<SK>(S, C, !alive, nullptr, lds_origin, pq, ps, it);
        if (took >= 0) {
            {   // the pixel's cotangent / spp, straight from the image (load_le_grad; a popped path is inside the shard)
                const float4 gi = io.d_image[ps.smp.px + ps.smp.py * (uint32_t)R.width];
                if (C.spp_pow2) le_grad = mk3(gi.x * C.inv_spp, gi.y * C.inv_spp, gi.z * C.inv_spp);   // x / 2^k == x * 2^-k exactly: three IEEE divisions (~30 VALU per trip) less
                else { const float fs = (float)C.spp; le_grad = mk3(__fdiv_rn(gi.x, fs), __fdiv_rn(gi.y, fs), __fdiv_rn(gi.z, fs)); }
                if (any_nan(le_grad)) le_grad = mk3(0.0f);
            }
            nrec = 0;
            last = -1;
            alive = true; pix = took;
        }
        if (__ballot(alive) != 0ull) {
            progress = true;
            bool done = false;
            // sweep state: set when a path ends and used up before the trip is over — local to the trip, so that it
            // holds no registers while the vertex is shaded
            f3 term_Li = mk3(0.0f);
            PackedVertex plast;                             // the vertex shaded this trip, as recorded
            plast.a = plast.b = plast.c = plast.d = plast.e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int sw_k = -1;                                  // next vertex the sweep consumes
            bool want_store = false, mute = false;
            SweepState sw; sw.A = mk3(0.0f); sw.Lv = mk3(0.0f); sw.s = 0.0f; sw.Z = 0.0f; sw.tw = 0.0f;
            if (alive) {
                PathVertex pv; float term_plfrac = 0.0f;
                Hit h;
                done = path_shade<SK, A, true, false, ENV>(S, R, C, io, lds, ps, it, pv, h, cnt);
                plast = pack_vertex(pv, le_grad, R.prb_mode);
                if (!done) { path_continue<A, false>(S, lds, ps, h, cnt); done = path_arrive<true, false, ENV>(S, ps, h, it, term_Li, cnt, &term_plfrac); }
                // Only a vertex whose path goes on is put away: when the path ends here (52 % of the vertices) the sweep below starts
                // from plast and nothing would read the record.  (5 LDS or scratch stores per vertex: 16.4 -> 15.5 ms for skipping
                // the vertices that stop at the shading step alone.)
                want_store = !done && ZDR_ABLATE != 3;      // (ablation 3, no sweep: nothing is kept, so no slot leaks)
                nrec++;
                if (done) {
                    alive = false;
                    // a NaN path (prb.py:100: contributes nothing) is swept all the same, muted: the sweep is what returns its slots
                    if (nrec > 0 && ZDR_ABLATE != 3) {
                        mute = any_nan(ps.L);
                        sw_k = nrec - 1;
                        sw.A = le_grad * term_Li; sw.Lv = sw.A; sw.s = 0.0f; sw.Z = 0.0f;
                        sw.tw = (R.prb_mode != ZDR_PRB_EXPECTATION) ? 0.0f : term_plfrac * dot(ps.beta, sw.A);   // emitter hit: d w_bsdf/dr = w_bsdf pl/(pb+pl) dln(pb)/dr
                    }
                }
            }
            // Reconverged: hand out slots, all requests of the trip at once.  Home slot first (slot == lane: conflict-free LDS access);
            // the lanes whose home is taken — a path's second and later records, or a home another lane borrowed — are ranked, the free
            // slots are ranked (a lane speaks for slot `lane`, then for slot 64 + lane), and request r takes free slot r: one
            // ds_permute sends every free slot's number to the lane of its rank, one ds_bpermute lets a request read the number at
            // its own rank.  Slots above 63 go first so that homes stay free; no slot left: the record goes to scratch.
            {
                const unsigned long long req = __ballot(want_store);
                int slot = -1;
                if (req != 0ull) {
                    // acquire: the ds_or of the previous trips' sweeps (any lane) are visible to this read
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const zdr_u4 fw = *(volatile const zdr_u4 *)lds_free;       // same address in every lane: a broadcast read
                    unsigned long long free_lo = ((unsigned long long)__builtin_amdgcn_readfirstlane(fw.y) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane(fw.x);
                    unsigned long long free_hi = ((unsigned long long)__builtin_amdgcn_readfirstlane(fw.w) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane(fw.z);
                    const bool home = want_store && lane < NS && ((free_lo >> lane) & 1ull) != 0ull;
                    const unsigned long long took = __ballot(home);
                    free_lo &= ~took;
                    if (home) slot = lane;
                    const unsigned long long rest = req & ~took;
                    if (rest != 0ull && (free_lo | free_hi) != 0ull) {
                        const int nrest = __popcll(rest);
                        const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(rest >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rest, 0u));   // rank of this lane's request
                        const bool mine = want_store && !home;
                        const int cnt_hi = __popcll(free_hi), cnt_lo = __popcll(free_lo);
                        // free slots 64 + lane
                        const bool fh = ((free_hi >> lane) & 1ull) != 0ull;
                        const int jh = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(free_hi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)free_hi, 0u));
                        const int at_hi = __builtin_amdgcn_ds_permute((fh ? jh : cnt_hi + (lane - jh)) << 2, lane);   // lane j < cnt_hi now holds the j-th free slot (minus 64)
                        const int pick_hi = __builtin_amdgcn_ds_bpermute(r << 2, at_hi);
                        if (mine && r < cnt_hi) slot = 64 + pick_hi;
                        free_hi &= ~__ballot(fh && jh < nrest);
                        // free slots `lane`, for the requests the upper slots did not serve
                        const int r2 = r - cnt_hi, nrest2 = nrest - cnt_hi;
                        const bool fl = ((free_lo >> lane) & 1ull) != 0ull;
                        const int jl = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(free_lo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)free_lo, 0u));
                        const int at_lo = __builtin_amdgcn_ds_permute((fl ? jl : cnt_lo + (lane - jl)) << 2, lane);
                        const int pick_lo = __builtin_amdgcn_ds_bpermute((r2 & 63) << 2, at_lo);
                        if (mine && r2 >= 0 && r2 < cnt_lo) slot = pick_lo;
                        free_lo &= ~__ballot(fl && jl < nrest2);
                    }
                    if (lane == 0) {
                        const zdr_u4 nw = {(unsigned int)free_lo, (unsigned int)(free_lo >> 32), (unsigned int)free_hi, (unsigned int)(free_hi >> 32)};
                        *(volatile zdr_u4 *)lds_free = nw;
                    }
                    // release: the new mask is in LDS before any lane's ds_or of the sweep below (in-order LDS, one wave)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (want_store) {
                    if (slot >= 0) {
                        float4 *r = lds_pool + slot;
                        r[0] = plast.a; r[NS] = plast.b; r[2 * NS] = plast.c; r[3 * NS] = plast.d; r[4 * NS] = plast.e;
                        lds_link[slot] = (unsigned char)last;
                        last = slot;
                    } else {
                        deep[nrec - 1] = plast; deep_link[nrec - 1] = last;
                        last = 255;
                    }
                }
            }
            ib.inflight[0] -= (uint32_t)__popcll(__ballot(done && (pix >> 6) == 0));
            ib.inflight[1] -= (uint32_t)__popcll(__ballot(done && (pix >> 6) == 1));
            // wave-uniform: sweep every finished path to its first vertex.  The sweep starts from the vertex packed this
            // trip (still in registers).
            PackedVertex cur = plast;
            int loc = last;                                 // where the record of the sweep's next step lives
#ifdef ZDR_MEASURE_STATS   // measurement build (tools/bwd_stats.sh): how full are the trips and the sweep iterations
            st_trips++; st_shaded += (unsigned long long)__popcll(__ballot(alive || done));
            st_fin += (unsigned long long)__popcll(__ballot(sw_k >= 0));
#endif
            int sweep_cap = (ZDR_ABLATE == 4) ? 2 : ((ZDR_ABLATE == 5) ? 1 : 64);   // timing-only ablations 4 / 5: the sweep loop cut after 2 / 1 iterations
            // One step consumes `cur`, then fetches the record of the NEXT step into the same registers and only then queues the
            // gradient: the fetch (LDS, or scratch beyond the LDS records) is under way while the push runs, and no record is copied
            // (the loop runs 5.45 times per trip at 18 % of the lanes, profiles/r3_bwd_sweep_ablation.txt; fetching one step ahead
            // into a second register set cost 24 v_mov per iteration, and unrolling by two
            // with the sets swapping roles was slower still: profiles/r3_bwd_sweep_ablation.txt section 3).
            while (__ballot(sw_k >= 0) != 0ull && sweep_cap-- > 0) {
                const bool swp = sw_k >= 0;
#ifdef ZDR_MEASURE_STATS
                st_iters++; st_steps += (unsigned long long)__popcll(__ballot(swp));
#endif
                float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                f2 guv; guv.x = 0.0f; guv.y = 0.0f;
                if (swp) { g = sweep_vertex(cur, sw, guv, R.prb_mode); sw_k--; }
                // everything that reads `cur` is finished here, before the fetch below overwrites it (left alone the compiler sinks
                // part of the step below the fetch, loads into a second register set and copies — with a wait in front of the copies)
                asm volatile("" : "+v"(g.x), "+v"(g.y), "+v"(g.z), "+v"(g.w), "+v"(guv.x), "+v"(guv.y), "+v"(sw.A.x), "+v"(sw.A.y), "+v"(sw.A.z),
                             "+v"(sw.Lv.x), "+v"(sw.Lv.y), "+v"(sw.Lv.z), "+v"(sw.s), "+v"(sw.Z), "+v"(sw.tw) : : "memory");
                const bool fetch = swp && sw_k >= 0;
                const bool pooled = fetch && loc != 255;
                int nloc = -1;
                if (pooled) {
                    const float4 *r = lds_pool + loc;
                    cur.a = r[0]; cur.b = r[NS]; cur.c = r[2 * NS]; cur.d = r[3 * NS]; cur.e = r[4 * NS];
                    nloc = (int)lds_link[loc];
                }
                // LDS first: both kinds of fetch write the same registers (for different lanes), and the second kind waits for the
                // first to land — an LDS read is back in ~100 cycles, a scratch read in ~500 and behind the flush's atomics
                asm volatile("" ::: "memory");
                if (fetch && loc == 255) { cur = deep[sw_k]; nloc = deep_link[sw_k]; }
                // the slots just read are free again: a wavefront-scope RELEASE or, so the reads of the record above are ordered before it
                // (and this wave's LDS operations execute in order anyway: a later write cannot overtake the read)
                if (pooled) __hip_atomic_fetch_or(&lds_free[loc >> 5], 1u << (loc & 31), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (fetch) loc = nloc;
                scatter_push(q, io.cells, swp && !mute && any_nonzero4(g) && !any_nan4(g), guv, g, R.tex_h, R.tex_w, ZDR_ABLATE);   // prb.py:178-187
            }
        }
#pragma unroll
        for (int b = 0; b < 2; b++)                         // an item whose samples are all generated and whose paths have ended frees its bank
            if (ib.logical[b] >= 0 && ib.inflight[b] == 0 && (b != bank || next_sample >= s_end)) ib.logical[b] = -1;
        if (__ballot(alive) == 0ull && pq.tail == pq.head && next_sample >= s_end && !more_items) break;
        stall = progress ? 0 : stall + 1;
        if (stall > 4) { raise_device_error(S, ZDR_DEVERR_STALL); break; }   // cannot happen (every branch above makes progress); never spin on the GPU, never end silently
    }
    scatter_finish(q, io.cells, R.tex_h, R.tex_w, ZDR_ABLATE);
    {   // every path has been swept, so every slot must be back: a leaked or doubly allocated slot is a protocol error, said aloud
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const zdr_u4 fw = *(volatile const zdr_u4 *)lds_free;
        const bool whole = fw.x == (unsigned int)all_lo && fw.y == (unsigned int)(all_lo >> 32) && fw.z == (unsigned int)all_hi && fw.w == (unsigned int)(all_hi >> 32);
        if (!whole && lane == 0 && stall <= 4 && ZDR_ABLATE == 0) raise_device_error(S, ZDR_DEVERR_POOL);
    }
#ifdef ZDR_MEASURE_STATS
    if (lane == 0) {
        atomicAdd(io.counters + 0, st_trips); atomicAdd(io.counters + 1, st_shaded); atomicAdd(io.counters + 2, st_fin);
        atomicAdd(io.counters + 3, st_iters); atomicAdd(io.counters + 4, st_steps);
        atomicAdd(io.counters + 5, q.st_flushes); atomicAdd(io.counters + 6, q.st_entries); atomicAdd(io.counters + 7, q.st_dups);
    }
#endif
#undef S
#undef R
#undef C
#undef io
}

// ---------------------------------------------------------------------- direct / collocated
template <int INTEG, int SK, class A, bool BWD, bool STATS, bool ENV>
#ifndef ZDR_MIN_WAVES_DIRECT
#define ZDR_MIN_WAVES_DIRECT 4   // brute-force direct kernels, cbox 512^2 spp 64: 153 VGPRs (3 waves per SIMD) 1.175 / 1.331 ms, 128 VGPRs (6 spilled) 1.110 / 1.267 ms
#endif
// (the environment instantiations take 170 VGPRs: bounded to 3 waves per SIMD, which costs no spill, instead of the 2 the compiler settles for)
__global__ __launch_bounds__(WAVE, (INTEG == ZDR_DIRECT && !A::kNeedsLds) ? (ENV ? 3 : ZDR_MIN_WAVES_DIRECT) : 1) void k_simple(ZDR_PATH_KERNEL_PARAMS) {
    ZDR_KARGS_BEGIN
#define S (ka->S)
#define R (ka->R)
#define C (ka->C)
#define io (ka->io)
    extern __shared__ int lds[];        // BvhAccel: traversal stacks (sized at launch); unused otherwise
    __shared__ __attribute__((aligned(16))) float lds_q[BWD ? ZDR_SCATTER_LDS_FLOATS : 4];
    const WorkItem w = decode_block(R);
    const uint32_t perm_seed = (SK == 0) ? xxhash32_4((uint32_t)w.x, (uint32_t)w.y, C.seed, 0u) : 0u;
    Counters cnt;
#pragma unroll
    for (int i = 0; i < 8; i++) cnt.c[i] = 0;
    f3 le_grad = mk3(0.0f);
    if (BWD) le_grad = load_le_grad(C, io, w);
    ScatterQueue q = scatter_queue_init(lds_q, R.tex_h, R.tex_w, R.cell_copies);
    const unsigned long long cam_mask = camera_mask(S, io, w);
    f3 sum = mk3(0.0f);
    for (uint32_t it = w.s_begin; it < w.s_end; it++) {     // integrator.py:15 (wave-uniform trip count)
        ZDR_KARGS_REFRESH
        float4 grad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        f2 guv; guv.x = 0.0f; guv.y = 0.0f;
        if (w.valid) {
            Sampler smp = sampler_make<SK>(C, (uint32_t)w.x, (uint32_t)w.y, perm_seed, it);
            f3 o, d;
            pixel_ray<SK, true>(R, C, smp, w.x, w.y, o, d);
            COUNT(C_SAMPLES);
            f3 rad;
            if (INTEG == ZDR_COLLOCATED) rad = collocated_sample<A, BWD, STATS>(S, R, io, lds, o, d, cam_mask, le_grad, cnt, guv, grad);
            else rad = direct_sample<SK, A, BWD, STATS, ENV>(S, R, C, io, lds, smp, o, d, cam_mask, le_grad, cnt, guv, grad);
            if (!any_nan(rad)) sum = sum + clamp_radiance(rad); else COUNT(C_NAN);
        }
        if (BWD) scatter_push(q, io.cells, w.valid && any_nonzero4(grad) && !any_nan4(grad), guv, grad, R.tex_h, R.tex_w, ZDR_ABLATE);
    }
    if (BWD) scatter_finish(q, io.cells, R.tex_h, R.tex_w, ZDR_ABLATE);
    if (!BWD && !STATS) store_pixel(R, C, io, w, sum);
    flush_counters<STATS>(io, cnt);
#undef S
#undef R
#undef C
#undef io
}

// Folds the staging cells into the gradient texture: texel (x, y) receives corner (dx, dy) of every
// cell (ix, iy) with clamp(ix + dx) == x and clamp(iy + dy) == y — the adjoint of read_bsdf's CLAMP
// bilinear lookup (interaction.py:47-60, 73-89).  Deterministic summation order.
__global__ void k_cells_to_grad(const float4 *__restrict__ cells, float4 *__restrict__ dmat, int tex_h, int tex_w, int copies) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= tex_w) return;
    int ixs[3], dxs[3], nx = 0, iys[3], dys[3], ny = 0;
    ixs[nx] = x - 1; dxs[nx++] = 1; ixs[nx] = x; dxs[nx++] = 0;
    if (x == 0) { ixs[nx] = -1; dxs[nx++] = 0; }
    if (x == tex_w - 1) { ixs[nx] = tex_w - 1; dxs[nx++] = 1; }
    iys[ny] = y - 1; dys[ny++] = 1; iys[ny] = y; dys[ny++] = 0;
    if (y == 0) { iys[ny] = -1; dys[ny++] = 0; }
    if (y == tex_h - 1) { iys[ny] = tex_h - 1; dys[ny++] = 1; }
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const size_t ncells = (size_t)(tex_h + 1) * (tex_w + 1);
    if (copies == 1) {
        for (int b = 0; b < ny; b++)
            for (int a = 0; a < nx; a++) {
                size_t cell = (size_t)(ixs[a] + 1) + (size_t)(tex_w + 1) * (iys[b] + 1);
                float4 c = cells[4 * cell + 2 * dxs[a] + dys[b]];
                acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
            }
    } else {   // few texels: the waves added into `copies` replicas of the cell array (scene.h); sum them in float64
        double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
        for (int k = 0; k < copies; k++)
            for (int b = 0; b < ny; b++)
                for (int a = 0; a < nx; a++) {
                    size_t cell = (size_t)k * ncells + (size_t)(ixs[a] + 1) + (size_t)(tex_w + 1) * (iys[b] + 1);
                    float4 c = cells[4 * cell + 2 * dxs[a] + dys[b]];
                    sx += c.x; sy += c.y; sz += c.z; sw += c.w;
                }
        acc = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
    }
    float4 d = dmat[(size_t)x + (size_t)tex_w * y];
    dmat[(size_t)x + (size_t)tex_w * y] = make_float4(d.x + acc.x, d.y + acc.y, d.z + acc.z, d.w + acc.w);
}

// render_duvdxy (uvgrad.py:76-98): mean over the samples of the screen->texture Jacobian, NaNs dropped
template <int SK, class A>
__global__ __launch_bounds__(WAVE) void k_uvgrad(DScene S, RenderCfg R, SamplerCfg C, KernelIO io) {
    extern __shared__ int lds[];
    const WorkItem w = decode_block(R);
    const uint32_t perm_seed = (SK == 0) ? xxhash32_4((uint32_t)w.x, (uint32_t)w.y, C.seed, 0u) : 0u;
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t it = w.s_begin; it < w.s_end; it++) {
        if (!w.valid) continue;
        Sampler smp = sampler_make<SK>(C, (uint32_t)w.x, (uint32_t)w.y, perm_seed, it);
        f2 off = sampler_next2<SK>(C, smp);
        if (R.use_tent) { off.x = tent_warp1(off.x) + 0.5f; off.y = tent_warp1(off.y) + 0.5f; }
        const float fx = (float)w.x + off.x, fy = (float)w.y + off.y;
        f3 o = ld3(R.cam_o), d[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {          // rays through (x, y), (x + 1, y), (x, y + 1)
            float px = R.two_over_w * (fx + (k == 1 ? 1.0f : 0.0f)) - 1.0f;
            float py = (R.two_over_h * (fy + (k == 2 ? 1.0f : 0.0f)) - 1.0f) * R.aspect;
            d[k] = normalize((ld3(R.cam_right) * (px * R.cam_tan) - ld3(R.cam_upp) * (py * R.cam_tan)) + ld3(R.cam_fwd));
        }
        float4 g = uvgrad_sample<A>(S, lds, o, d[0], o, d[1], o, d[2]);
        if (!any_nan4(g)) { sum.x += g.x; sum.y += g.y; sum.z += g.z; sum.w += g.w; }
    }
    if (w.valid) {
        float fs = (float)C.spp;
        io.image[w.pix] = make_float4(__fdiv_rn(sum.x, fs), __fdiv_rn(sum.y, fs), __fdiv_rn(sum.z, fs), __fdiv_rn(sum.w, fs));
    }
}

// sums the per-chunk partial images in chunk order (deterministic), integrator.py:29
__global__ void k_reduce_chunks(RenderCfg R, uint32_t spp, const float4 *partial, float4 *image) {
    int x = R.x0 + blockIdx.x * blockDim.x + threadIdx.x, y = R.y0 + blockIdx.y;
    if (x >= R.x1 || y >= R.y1) return;
    const size_t pix = (size_t)x + (size_t)y * R.width;
    const int lx = x - R.x0, ly = y - R.y0, ty = ly >> 3;
    int c = (lx >> 3) - (ty * R.shard_skew) % R.tiles_x;     // the tile's number in its row (decode_item)
    if (c < 0) c += R.tiles_x;
    const int tile = ty * R.tiles_x + c;
    if (tile % R.shard_count != R.shard_index) return;      // another shard's pixel: untouched
    const size_t slot = (size_t)(tile / R.shard_count) * 64 + (size_t)((ly & 7) * 8 + (lx & 7));
    f3 s = mk3(0.0f);
    for (int c = 0; c < R.nchunks; c++) { float4 p = partial[(size_t)c * R.ntiles * 64 + slot]; s = s + mk3(p.x, p.y, p.z); }
    float fs = (float)spp;
    image[pix] = make_float4(__fdiv_rn(s.x, fs), __fdiv_rn(s.y, fs), __fdiv_rn(s.z, fs), R.alpha);
}

// Zero-fill of the per-call workspaces (staging cells, work counters) by a KERNEL, not hipMemsetAsync: captured in a HIP graph, the
// runtime's memset node of the 64 MiB cell array was observed to run out of order with the kernels around it from the second replay on
// (the gradient came back zero); a kernel node is ordered like every other launch.  n16 = number of 16-byte words.
__global__ void k_zero(uint4 *__restrict__ p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int zdr_launch_zero(void *p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    const size_t n16 = (bytes + 15) / 16;                    // every workspace is allocated in multiples of 16 bytes
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, st, (uint4 *)p, n16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ----------------------------------------------------------------------------------- launch
// dynamic LDS of a wave that traverses the BVH: the first entries of the per-lane stacks (accel.h).  Sets S.lds_stack.
// path_kernels: the launch runs BvhAccel::shadow_and_closest (k_path, k_path_bwd, k_path_dump), whose walk_steal keeps 5 x 64 ints behind the stack.
static size_t bvh_dyn_lds(DScene &S, bool backward, bool path_kernels) {
    S.lds_stack = std::min<int>(S.stack_entries, backward ? ZDR_BVH_LDS_STACK_BWD : ZDR_BVH_LDS_STACK);
    return (size_t)S.lds_stack * WAVE * sizeof(int) + ((ZDR_BVH_STEAL && path_kernels) ? 5 * WAVE * sizeof(int) : 0);
}
// Persistent grid of the path kernels: as many single-wave workgroups as the chip holds at once (never more
// than there are items).  A workgroup that is not resident at first simply starts later and draws what is left.
#define ZDR_LDS_BLOCK_BYTES 1280
#define ZDR_LDS_BLOCKS_PER_CU 128
template <class K>
static dim3 persistent_grid(K kernel, size_t dyn, int nitems) {
    int per_cu = 0, dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    {   // the occupancy query costs tens of microseconds: remember it per (kernel, dynamic LDS, device)
        static std::mutex mu;
        static std::map<std::tuple<const void *, size_t, int>, std::pair<int, int>> cache;
        std::lock_guard<std::mutex> lock(mu);
        auto key = std::make_tuple((const void *)kernel, dyn, dev);
        auto it = cache.find(key);
        if (it == cache.end()) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, dyn) != hipSuccess || per_cu < 1) per_cu = 8;
            // gfx950 hands out its 160 KiB of LDS in 128 blocks of 1,280 bytes and the query above does not round up: with 13,264
            // bytes per wave it answered 12, 11 were resident, and the twelfth wave of every CU started when the launch was over
            // (profiles/r3_bwd_records_and_atomics.txt).  A persistent grid must not be larger than what is resident at once.
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, (const void *)kernel) == hipSuccess) {
                const size_t blocks = (fa.sharedSizeBytes + dyn + ZDR_LDS_BLOCK_BYTES - 1) / ZDR_LDS_BLOCK_BYTES;
                if (blocks > 0) per_cu = std::max(1, std::min<int>(per_cu, (int)(ZDR_LDS_BLOCKS_PER_CU / blocks)));
            }
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            it = cache.emplace(key, std::make_pair(per_cu, cus)).first;
        }
        per_cu = it->second.first; cus = it->second.second;
    }
    if (const char *e = getenv("ZDR_PERSISTENT_WAVES_PER_CU")) per_cu = std::max(1, atoi(e));
    long g = std::min<long>((long)per_cu * cus, (long)ZDR_MAX_PERSISTENT_BLOCKS);
    return dim3((unsigned)std::max<long>(1, std::min<long>(g, nitems)));
}

template <int SK, class A>
static void launch_path(int nitems, size_t dyn, hipStream_t st, const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int backward, int stats) {
    // the environment-light code is a separate instantiation: inside the default kernels it cost 16 % (cbox forward)
#define ZDR_LAUNCH_PERSISTENT(K) hipLaunchKernelGGL((K), persistent_grid((K), dyn, nitems), dim3(WAVE), dyn, st, S, R, C, io)
    if (S.env_count > 0) {
        if (backward) ZDR_LAUNCH_PERSISTENT((k_path_bwd<SK, A, true>));
        else if (stats) ZDR_LAUNCH_PERSISTENT((k_path<SK, A, true, true>));
        else ZDR_LAUNCH_PERSISTENT((k_path<SK, A, false, true>));
    } else {
        if (backward) ZDR_LAUNCH_PERSISTENT((k_path_bwd<SK, A, false>));
        else if (stats) ZDR_LAUNCH_PERSISTENT((k_path<SK, A, true, false>));
        else ZDR_LAUNCH_PERSISTENT((k_path<SK, A, false, false>));
    }
#undef ZDR_LAUNCH_PERSISTENT
}

template <int INTEG, int SK, class A>
static void launch_simple(dim3 grid, size_t dyn, hipStream_t st, const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int backward, int stats) {
    if (INTEG == ZDR_DIRECT && S.env_count > 0) {
        if (backward) hipLaunchKernelGGL((k_simple<INTEG, SK, A, true, false, true>), grid, dim3(WAVE), dyn, st, S, R, C, io);
        else if (stats) hipLaunchKernelGGL((k_simple<INTEG, SK, A, false, true, true>), grid, dim3(WAVE), dyn, st, S, R, C, io);
        else hipLaunchKernelGGL((k_simple<INTEG, SK, A, false, false, true>), grid, dim3(WAVE), dyn, st, S, R, C, io);
    } else {
        if (backward) hipLaunchKernelGGL((k_simple<INTEG, SK, A, true, false, false>), grid, dim3(WAVE), dyn, st, S, R, C, io);
        else if (stats) hipLaunchKernelGGL((k_simple<INTEG, SK, A, false, true, false>), grid, dim3(WAVE), dyn, st, S, R, C, io);
        else hipLaunchKernelGGL((k_simple<INTEG, SK, A, false, false, false>), grid, dim3(WAVE), dyn, st, S, R, C, io);
    }
}
template <int SK, class A>
static void launch_integ(int integrator, dim3 grid, size_t dyn, hipStream_t st, const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int backward, int stats) {
    if (integrator == ZDR_UVGRAD) hipLaunchKernelGGL((k_uvgrad<SK, A>), grid, dim3(WAVE), dyn, st, S, R, C, io);
    else if (integrator == ZDR_PATH) launch_path<SK, A>(R.ntiles * R.nchunks, dyn, st, S, R, C, io, backward, stats);
    else if (integrator == ZDR_DIRECT) launch_simple<ZDR_DIRECT, SK, A>(grid, dyn, st, S, R, C, io, backward, stats);
    else launch_simple<ZDR_COLLOCATED, SK, A>(grid, dyn, st, S, R, C, io, backward, stats);
}

int zdr_launch_render(const DScene &S_in, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io,
                      int integrator, int accel_is_bvh, int backward, int stats, hipStream_t st) {
    DScene S = S_in;
    int nblocks = R.ntiles * R.nchunks;
    if (nblocks <= 0) return 0;
    dim3 grid(((nblocks + 7) >> 3) << 3);                   // multiple of 8 for the XCD remap
    const size_t dyn = accel_is_bvh ? bvh_dyn_lds(S, backward != 0, integrator == ZDR_PATH) : 0;
    if (io.tile_masks && !io.tile_masks_valid)
        hipLaunchKernelGGL(k_tile_masks, dim3(R.tiles_x * R.tiles_y), dim3(WAVE), 0, st, S, R, (unsigned long long *)io.tile_masks);
    if (C.kind == ZDR_SAMPLER_CMJ) {
        if (accel_is_bvh) launch_integ<0, BvhAccel>(integrator, grid, dyn, st, S, R, C, io, backward, stats);
        else launch_integ<0, BruteAccel>(integrator, grid, dyn, st, S, R, C, io, backward, stats);
    } else {
        if (accel_is_bvh) launch_integ<1, BvhAccel>(integrator, grid, dyn, st, S, R, C, io, backward, stats);
        else launch_integ<1, BruteAccel>(integrator, grid, dyn, st, S, R, C, io, backward, stats);
    }
    if (backward) {   // fold the staging cells into d_material (+=)
        dim3 g((R.tex_w + 63) / 64, R.tex_h);
        hipLaunchKernelGGL(k_cells_to_grad, g, dim3(64), 0, st, (const float4 *)io.cells, (float4 *)io.d_material, R.tex_h, R.tex_w, R.cell_copies);
    }
    if (!backward && !stats && R.nchunks > 1) {
        dim3 g((R.x1 - R.x0 + 63) / 64, R.y1 - R.y0);
        hipLaunchKernelGGL(k_reduce_chunks, g, dim3(64), 0, st, R, C.spp, (const float4 *)io.partial, io.image);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------- ray batch queries
template <class A, bool ANY>
__global__ __launch_bounds__(WAVE) void k_trace(DScene S, const float4 *rays, uint32_t n, int32_t *out_i, float *out_f) {
    extern __shared__ int lds[];        // BvhAccel: traversal stacks (sized at launch); unused otherwise
    uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    bool valid = i < n;
    float4 a = valid ? rays[2 * (size_t)i] : make_float4(0, 0, 0, 0), b = valid ? rays[2 * (size_t)i + 1] : make_float4(0, 0, 1, 0);
    if (ANY) {
        bool occ = A::any(S, lds, xyz(a), xyz(b), a.w, b.w);
        if (valid) out_i[i] = occ ? 1 : 0;
    } else {
        Hit h = A::closest(S, lds, xyz(a), xyz(b), a.w, b.w);
        if (valid) {
            int inst = -1, prim = -1;
            if (h.slot >= 0) { inst = __float_as_int(S.shade[8 * (size_t)h.slot + 6].w); prim = __float_as_int(S.shade[8 * (size_t)h.slot + 7].y); }
            out_i[2 * (size_t)i] = inst; out_i[2 * (size_t)i + 1] = prim;
            float hu = h.u, hv = h.v;
            if (h.slot >= 0) {   // the slot's corners may be a rotation of the input triangle's (quads, zdr_api.cpp): report the barycentrics of the INPUT corners 1 and 2
                const int ro = __float_as_int(S.shade[8 * (size_t)h.slot + 7].z);
                const float hw = 1.0f - h.u - h.v;
                if (ro == 1) { hu = hw; hv = h.u; } else if (ro == 2) { hu = h.v; hv = hw; }
            }
            out_f[3 * (size_t)i] = hu; out_f[3 * (size_t)i + 1] = hv; out_f[3 * (size_t)i + 2] = (h.slot >= 0) ? h.t : b.w;
        }
    }
}

int zdr_launch_trace(const DScene &S_in, int accel_is_bvh, int any, const float *rays, uint32_t n, int32_t *out_i, float *out_f, hipStream_t st) {
    DScene S = S_in;
    if (n == 0) return 0;
    dim3 grid((n + WAVE - 1) / WAVE);
    const float4 *r = (const float4 *)rays;
    if (accel_is_bvh) {
        const size_t dyn = bvh_dyn_lds(S, false, false);
        if (any) hipLaunchKernelGGL((k_trace<BvhAccel, true>), grid, dim3(WAVE), dyn, st, S, r, n, out_i, out_f);
        else hipLaunchKernelGGL((k_trace<BvhAccel, false>), grid, dim3(WAVE), dyn, st, S, r, n, out_i, out_f);
    } else {
        if (any) hipLaunchKernelGGL((k_trace<BruteAccel, true>), grid, dim3(WAVE), 0, st, S, r, n, out_i, out_f);
        else hipLaunchKernelGGL((k_trace<BruteAccel, false>), grid, dim3(WAVE), 0, st, S, r, n, out_i, out_f);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------- path dump
// Test hook (zdr_path_dump, include/zdr.h): lane = one queried camera sample, walked vertex by vertex with the SAME
// device functions the path kernels run (path_arrive, path_shade, path_continue, pack_vertex, sweep_vertex), records kept
// per lane.  What it writes per path is what k_path adds to the pixel and what k_path_bwd queues for the scatter.
template <int SK, class A, bool ENV>
__global__ __launch_bounds__(WAVE) void k_path_dump(DScene S, RenderCfg R, SamplerCfg C, KernelIO io, const int32_t *queries, uint32_t n, int maxv, float *out) {
    extern __shared__ int lds[];
    const uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;                                     // no wave-level operation below: lanes are independent
    const int stride = 8 + 24 * maxv;
    float *o = out + (size_t)i * stride;
    for (int k = 0; k < stride; k++) o[k] = 0.0f;
    const int px = queries[3 * i], py = queries[3 * i + 1];
    const uint32_t idx = (uint32_t)queries[3 * i + 2];
    if (px < 0 || py < 0 || px >= R.width || py >= R.height || idx >= C.spp) return;   // a query outside the image / sample range reads as all zeros
    Counters cnt;
    const uint32_t perm_seed = (SK == 0) ? xxhash32_4((uint32_t)px, (uint32_t)py, C.seed, 0u) : 0u;
    PathState ps;
    ps.smp = sampler_make<SK>(C, (uint32_t)px, (uint32_t)py, perm_seed, idx);
    pixel_ray<SK>(R, C, ps.smp, px, py, ps.o, ps.d);
    ps.beta = mk3(1.0f); ps.L = mk3(0.0f); ps.pdf_bsdf = 1e30f; ps.depth = 0;
    f3 le_grad = mk3(R.inv_spp);
    if (io.d_image) {                                       // integrator.py:38-40
        const float4 gi = io.d_image[(size_t)px + (size_t)py * R.width];
        const float fs = (float)C.spp;
        le_grad = mk3(__fdiv_rn(gi.x, fs), __fdiv_rn(gi.y, fs), __fdiv_rn(gi.z, fs));
        if (any_nan(le_grad)) le_grad = mk3(0.0f);
    }
    PackedVertex recs[ZDR_MAX_RECORDED_DEPTH];
    Hit h = A::closest(S, lds, ps.o, ps.d, 0.0f, 1e30f);
    Interaction it; f3 term_Li = mk3(0.0f); float plfrac = 0.0f;
    bool done = path_arrive<true, false, ENV>(S, ps, h, it, term_Li, cnt, &plfrac);
    int nv = 0;
    while (!done && nv < ZDR_MAX_RECORDED_DEPTH) {
        PathVertex pv; Hit h2; h2.slot = -1; h2.u = h2.v = h2.t = 0.0f;
        const f3 L0 = ps.L;
        const int slot = h.slot;
        term_Li = mk3(0.0f); plfrac = 0.0f;
        bool stop = path_shade<SK, A, true, false, ENV>(S, R, C, io, lds, ps, it, pv, h2, cnt);
        recs[nv] = pack_vertex(pv, le_grad, R.prb_mode);
        if (nv < maxv) {
            float *q = o + 8 + 24 * nv;
            const int went_on = pv.c != 0.0f ? 1 : 0;       // the BSDF sample was kept (prb.py:73-87 passed)
            q[0] = __int_as_float(it.inst); q[1] = S.shade[8 * (size_t)slot + 7].y; q[2] = it.uv.x; q[3] = it.uv.y;
            q[4] = __int_as_float((pv.cL != 0.0f ? 1 : 0) | (went_on << 1) | (pv.rr << 2));
            if (went_on) { q[5] = ps.pdf_bsdf; q[6] = ps.d.x; q[7] = ps.d.y; q[8] = ps.d.z; q[9] = ps.beta.x; q[10] = ps.beta.y; q[11] = ps.beta.z; }
            const f3 ln = ps.L - L0;
            q[16] = ln.x; q[17] = ln.y; q[18] = ln.z;
        }
        nv++;
        if (!stop) { path_continue<A, false>(S, lds, ps, h2, cnt); h = h2; stop = path_arrive<true, false, ENV>(S, ps, h, it, term_Li, cnt, &plfrac); }
        done = stop;
    }
    o[0] = __int_as_float(nv); o[1] = ps.L.x; o[2] = ps.L.y; o[3] = ps.L.z; o[5] = term_Li.x; o[6] = term_Li.y; o[7] = term_Li.z;
    if (!any_nan(ps.L) && nv > 0) {                         // prb.py:100; the sweep of k_path_bwd
        SweepState sw; sw.A = le_grad * term_Li; sw.Lv = sw.A; sw.s = 0.0f; sw.Z = 0.0f; sw.tw = (R.prb_mode != ZDR_PRB_EXPECTATION) ? 0.0f : plfrac * dot(ps.beta, sw.A);
        for (int k = nv - 1; k >= 0; k--) {
            f2 guv; const float4 g = sweep_vertex(recs[k], sw, guv, R.prb_mode);
            if (k < maxv) { float *q = o + 8 + 24 * k; q[12] = g.x; q[13] = g.y; q[14] = g.z; q[15] = g.w; }
        }
    }
}

int zdr_launch_path_dump(const DScene &S_in, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int accel_is_bvh,
                         const int32_t *queries, uint32_t n, int32_t maxv, float *out, hipStream_t st) {
    DScene S = S_in;
    if (n == 0) return 0;
    dim3 grid((n + WAVE - 1) / WAVE);
    const size_t dyn = accel_is_bvh ? bvh_dyn_lds(S, true, true) : 0;
#define ZDR_DUMP(SKV, ACC, ENVV) hipLaunchKernelGGL((k_path_dump<SKV, ACC, ENVV>), grid, dim3(WAVE), dyn, st, S, R, C, io, queries, n, maxv, out)
    const bool env = S.env_count > 0, cmj = C.kind == ZDR_SAMPLER_CMJ;
    if (accel_is_bvh) { if (cmj) { if (env) ZDR_DUMP(0, BvhAccel, true); else ZDR_DUMP(0, BvhAccel, false); } else { if (env) ZDR_DUMP(1, BvhAccel, true); else ZDR_DUMP(1, BvhAccel, false); } }
    else { if (cmj) { if (env) ZDR_DUMP(0, BruteAccel, true); else ZDR_DUMP(0, BruteAccel, false); } else { if (env) ZDR_DUMP(1, BruteAccel, true); else ZDR_DUMP(1, BruteAccel, false); } }
#undef ZDR_DUMP
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------- sampler dump
template <int SK>
__global__ void k_sampler_dump(SamplerCfg C, const int32_t *q, uint32_t n, int nvert, int rr_depth, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t px = (uint32_t)q[3 * i], py = (uint32_t)q[3 * i + 1], idx = (uint32_t)q[3 * i + 2];
    uint32_t perm_seed = (SK == 0) ? xxhash32_4(px, py, C.seed, 0u) : 0u;
    Sampler s = sampler_make<SK>(C, px, py, perm_seed, idx);
    const int stride = 2 + 8 * nvert;
    float *o = out + (size_t)i * stride;
    int k = 0;
    f2 u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
    for (int v = 0; v < nvert; v++) {
        o[k++] = sampler_next<SK>(C, s); o[k++] = sampler_next<SK>(C, s);
        u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
        o[k++] = sampler_next<SK>(C, s);
        u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
        if (v >= rr_depth) o[k++] = sampler_next<SK>(C, s);
    }
    for (; k < stride; k++) o[k] = 0.0f;
}

// The same numbers drawn THE WAY THE PATH KERNELS DRAW THEM (shade_ctx / sample_bsdf, integrators.h): the pixel's next2f, then per
// vertex all seven numbers at once through cmj_vertex_samples (two Kensler permutations per register) and the roulette draw through
// cmj_next_with_index — whenever cmj_can_batch(C), which is every BASELINE configuration — else the calls one by one.
template <int SK>
__global__ void k_vertex_sampler_dump(SamplerCfg C, const int32_t *q, uint32_t n, int nvert, int rr_depth, int direct, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t px = (uint32_t)q[3 * i], py = (uint32_t)q[3 * i + 1], idx = (uint32_t)q[3 * i + 2];
    uint32_t perm_seed = (SK == 0) ? xxhash32_4(px, py, C.seed, 0u) : 0u;
    Sampler s = sampler_make<SK>(C, px, py, perm_seed, idx);
    const int stride = 2 + 8 * nvert;
    float *o = out + (size_t)i * stride;
    int k = 0;
    f2 u = direct ? sampler_pixel_offset<SK>(C, s) : sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;       // pixel_ray's draw: packed in the direct kernels only
    const bool pre = (SK == 0) && cmj_can_batch(C);             // shade_ctx's x.pre (without an environment light)
    for (int v = 0; v < nvert; v++) {
        if (pre && direct) {                                    // direct_sample (integrators.h): after the packed pixel draw, the calls one by one
            o[k++] = sampler_next<SK>(C, s); o[k++] = sampler_next<SK>(C, s);
            u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
            o[k++] = sampler_next<SK>(C, s);
            u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
            if (v >= rr_depth) o[k++] = sampler_next<SK>(C, s);
        } else if (pre) {
            const VertexSamples vs = cmj_vertex_samples(C, s);
            o[k++] = vs.u_pick; o[k++] = vs.u_prim; o[k++] = vs.u_pt.x; o[k++] = vs.u_pt.y; o[k++] = vs.u_lobe; o[k++] = vs.u_dir.x; o[k++] = vs.u_dir.y;
            if (v >= rr_depth) o[k++] = cmj_next_with_index(C, s, vs.i_rr);
        } else {
            o[k++] = sampler_next<SK>(C, s); o[k++] = sampler_next<SK>(C, s);
            u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
            o[k++] = sampler_next<SK>(C, s);
            u = sampler_next2<SK>(C, s); o[k++] = u.x; o[k++] = u.y;
            if (v >= rr_depth) o[k++] = sampler_next<SK>(C, s);
        }
    }
    for (; k < stride; k++) o[k] = 0.0f;
}

// as_kernels: 0 = the calls one by one, 1 = as the path kernels draw, 2 = as the direct kernels draw
int zdr_launch_sampler_dump(const SamplerCfg &C, const int32_t *queries, uint32_t n, int32_t nvert, int32_t rr_depth, float *out, int as_path_kernels, int *batched, hipStream_t st) {
    const int direct = as_path_kernels == 2;
    if (batched) *batched = (as_path_kernels && C.kind == ZDR_SAMPLER_CMJ && cmj_can_batch(C)) ? 1 : 0;
    if (n == 0) return 0;
    dim3 grid((n + 63) / 64);
    if (as_path_kernels) {
        if (C.kind == ZDR_SAMPLER_CMJ) hipLaunchKernelGGL(k_vertex_sampler_dump<0>, grid, dim3(64), 0, st, C, queries, n, nvert, rr_depth, direct, out);
        else hipLaunchKernelGGL(k_vertex_sampler_dump<1>, grid, dim3(64), 0, st, C, queries, n, nvert, rr_depth, direct, out);
    } else if (C.kind == ZDR_SAMPLER_CMJ) hipLaunchKernelGGL(k_sampler_dump<0>, grid, dim3(64), 0, st, C, queries, n, nvert, rr_depth, out);
    else hipLaunchKernelGGL(k_sampler_dump<1>, grid, dim3(64), 0, st, C, queries, n, nvert, rr_depth, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
