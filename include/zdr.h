/* zdr.h — C-ABI of libzdr_hip.so, the MI355X (gfx950) back end of the zdr render()/PRB hot path.
 *
 * This is the drop-in boundary: everything the reference's
 * `Scene.render_forward` / `Scene.render_backward` (/root/reference/render.py:159-199) obtain
 * from LuisaCompute — BVH build, ray traversal, the fused integrator kernels, in-kernel
 * autodiff and the float atomic scatter — sits behind these entry points.  Plain pointers and
 * sizes only; no torch types.  Device pointers are BORROWED for the duration of a call; work
 * is enqueued on `stream` (a hipStream_t, NULL = the default stream) and NOT synchronised:
 * the caller decides when to wait (the reference synchronises on both sides, render.py:165,172).
 *
 * All functions return 0 on success or a negative ZDR_E_* code; zdr_last_error() gives the
 * thread-local message.  One in-flight call per scene handle (render.py:216-222: the
 * reference scene is not re-entrant either) — the handle owns per-call workspaces (staging cells, chunk
 * partials, work counters, the parked-vertex FIFOs), so two renders of ONE scene must not overlap, not even on
 * different streams: enqueue them on one stream, or use one scene handle per stream.
 *
 * Stream capture (hipStreamBeginCapture / torch.cuda.graph).  The render calls only enqueue, so a call made while its stream is
 * capturing is recorded into the graph — provided one eager call of the same kind, resolution and spp has sized the handle's
 * workspaces before (a call that would have to allocate while capturing returns ZDR_E_UNSUPPORTED).  A graph names the handle's
 * workspaces and everything of the scene at capture time (camera, lights, environment, sampler tables, seed are frozen into the
 * kernel arguments).  From the first captured call on the handle therefore (a) never frees a device buffer before
 * zdr_scene_destroy — one that must grow is replaced and the old one kept alive for the graphs that name it — and (b) rebuilds
 * the camera-ray tile masks in every call, captured or eager, because a replay rewrites them for its own view.  Eager calls of
 * any view or size and any number of captures may thus be interleaved with replays on the handle's stream; what still holds is
 * the rule above: one call (or replay) in flight per handle.  Destroy the graphs before the handle.
 */
#ifndef ZDR_H
#define ZDR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZDR_VERSION_STRING "zdr-mi355x 0.3 (gfx950)"
/* Bumped whenever a struct of this header changes size or meaning (2: tile shard + prb_mode fields; 3: struct_size).
 * A binding asserts zdr_abi_version() == ZDR_ABI_VERSION of the header it was written against. */
#define ZDR_ABI_VERSION 3

enum { ZDR_OK = 0, ZDR_E_INVALID = -1, ZDR_E_HIP = -2, ZDR_E_UNSUPPORTED = -3, ZDR_E_NOMEM = -4 };

/* integrators = render.py:65-69; ZDR_UVGRAD = render_duvdxy's kernel (uvgrad.py:76-98, forward only:
 * the image receives (dudx, dvdx, dudy, dvdy) per pixel) */
enum { ZDR_COLLOCATED = 0, ZDR_DIRECT = 1, ZDR_PATH = 2, ZDR_UVGRAD = 3 };
/* samplers = integrator.py:16-17 (corrmj.py is self-contained; pmj02bn.py needs tables) */
enum { ZDR_SAMPLER_CMJ = 0, ZDR_SAMPLER_PMJ02BN = 1 };
/* acceleration structure used for LuisaCompute's Accel (render.py:74,109,127) */
enum { ZDR_ACCEL_AUTO = 0, ZDR_ACCEL_BRUTE = 1, ZDR_ACCEL_BVH = 2 };
enum { ZDR_PRB_EXPECTATION = 0, ZDR_PRB_DETACHED = 1, ZDR_PRB_LITERAL = 2 };

typedef struct zdr_scene zdr_scene;

/* render.py:28 — Camera = StructType(fov, origin, target, up); fov = full horizontal angle (rad) */
typedef struct {
    float fov;
    float origin[3], target[3], up[3];
} zdr_camera;

/* Arguments of one kernel dispatch (integrator.py:10-11, render.py:168-171,193-196) plus the
 * shard this call covers: a pixel rectangle and a sample-index range. */
typedef struct {
    uint32_t struct_size;              /* = sizeof(zdr_render_params) of the caller's header; a mismatch is ZDR_E_INVALID */
    int32_t integrator;                /* ZDR_COLLOCATED | ZDR_DIRECT | ZDR_PATH */
    int32_t sampler;                   /* ZDR_SAMPLER_* */
    int32_t width, height;             /* res = (W, H); image tensor is (H, W, 4) */
    uint32_t spp, seed;                /* backward: the caller passes seed + 1 (render.py:196) */
    int32_t use_tent;                  /* scene.use_tent_filter (render.py:71) */
    int32_t x0, y0, x1, y1;            /* pixels [x0,x1) x [y0,y1) are rendered, others untouched */
    uint32_t sample_begin, sample_end; /* sample indices [begin,end) of [0,spp) are evaluated */
    int32_t max_depth, rr_depth;       /* prb.py:15-16: 16, 2 */
    zdr_camera camera;
    int32_t tex_h, tex_w;              /* material tensor is (tex_h, tex_w, 4) float32 */
    /* Interleaved pixel-tile shard (BASELINE configs[3] "pixel-tiled across 8 GPUs"): the rectangle is cut
     * into 8x8 tiles numbered row by row from its own corner, row r starting at column r (mod the row length) — so that
     * the tiles of one shard run along diagonals, not down the columns of the image — and this call renders the tiles
     * whose number is congruent to tile_shard_index modulo tile_shard_count: one launch per rank, every rank sees every
     * part of the image (load balance).  The shards of one count partition the rectangle.  tile_shard_count <= 1: the
     * whole rectangle. */
    int32_t tile_shard_index, tile_shard_count;
    /* Form of the PRB adjoint (backward of the path integrator only).  ZDR_PRB_EXPECTATION (0, default): the derivative of
     * the forward's expectation, which is what finite differences of render() measure (BASELINE.json's gradient bar) —
     * Russian roulette without an upper clamp (prb.py:83) makes the expectation depend on the roulette probabilities and
     * on the MIS weights, and this form differentiates through them.  ZDR_PRB_DETACHED (1): every roulette factor and MIS
     * weight held constant, which is what the reference's autodiff blocks compute (prb.py:138-146, 157-163, with the
     * corrected BSDF weight of SURVEY App. B-3).  ZDR_PRB_LITERAL (2): as DETACHED but with the BSDF-sample adjoint seeded
     * exactly as prb.py:157-163 writes it, backward(bsdf, beta / pdf_bsdf * Le * le_grad) with Le the remaining path
     * radiance — which already contains beta * bsdf / pdf, so this is NOT the derivative of the forward (19 % off finite
     * differences, tests/test_oracle_render.py); it exists so that the one output the reference defines and the other two
     * modes cannot produce is available for comparison. */
    int32_t prb_mode;
} zdr_render_params;

typedef struct {
    uint32_t ntris, nverts, ninst, light_count;
    int32_t accel;                     /* ZDR_ACCEL_BRUTE or ZDR_ACCEL_BVH actually in use */
    uint32_t bvh_nodes, bvh_max_depth, bvh_stack_entries;   /* BVH4 nodes, depth of the binary SAH tree, traversal-stack entries per lane the tree can need (the first 6-12 in LDS, the rest in scratch) */
    int32_t device;
    uint64_t device_bytes;             /* HBM held by the scene */
} zdr_scene_info_t;

const char *zdr_version(void);
int zdr_abi_version(void);             /* ZDR_ABI_VERSION the library was built with */
const char *zdr_last_error(void);

/* Replaces Scene.load_geometry (render.py:73-128): luisa.Buffer uploads, accel.add(vb, tb,
 * transform), heap.emplace(...), accel.update().  HOST inputs:
 *   verts8          nverts x 8 float32  {v[3], vt[2], vn[3]} in object space (vertex.py:4)
 *   tris            ntris  x 3 int32    indices into verts8
 *   inst_tri_begin  ninst + 1 int32     triangles [begin[i], begin[i+1]) belong to instance i
 *   inst_xform      ninst x 16 float32  row-major object->world 4x4 (NULL = identity)
 *   inst_emission   ninst x 3 float32   (render.py:85-91; a light is any emission component > 0)
 * Only instance 0 is textured; every other instance is an emitter or a blocker (prb.py:45). */
int zdr_scene_create(const float *verts8, uint32_t nverts, const int32_t *tris, uint32_t ntris,
                     const int32_t *inst_tri_begin, const float *inst_xform, const float *inst_emission,
                     uint32_t ninst, int device, int accel, zdr_scene **out);
int zdr_scene_destroy(zdr_scene *scene);
int zdr_scene_info(const zdr_scene *scene, zdr_scene_info_t *info);

/* Replaces Scene.update_lights (render.py:130-148). HOST input ninst x 3; rebuilds the light list. */
int zdr_scene_set_emissions(zdr_scene *scene, const float *inst_emission, void *stream);

/* Replaces Scene.add_envmap / load_envmap (render.py:150-156, envmap.py:116-203): a lat-long environment
 * light.  HOST inputs, copied to the device: tex (tex_h x tex_w x 4 float32, already made square as
 * envmap.py:123-128 does) and the importance-sampling tables the host builds from it (zdr_amd/envmap.py:
 * alias_prob / alias_idx hold the marginal p(y) table (map_h entries) followed by map_h conditional p(x|y)
 * tables of map_w entries; pdf is map_h x map_w).  tex == NULL removes the environment (env_count = 0). */
int zdr_scene_set_envmap(zdr_scene *scene, const float *tex, uint32_t tex_h, uint32_t tex_w, const float *alias_prob,
                         const int32_t *alias_idx, const float *pdf, uint32_t map_w, uint32_t map_h);

/* Tables of the PMJ02bn sampler (pmj02bn.py:9-18; the reference's own are absent,
 * .MISSING_LARGE_BLOBS).  HOST inputs, copied to the device: pmj [nsets][nsamples][2] uint32
 * (value / 2^32), bn [ntex][res][res] uint16 (value / 2^16). */
int zdr_scene_set_pmj02bn_tables(zdr_scene *scene, const uint32_t *pmj, uint32_t nsets, uint32_t nsamples,
                                 const uint16_t *bn, uint32_t ntex, uint32_t bnres);

/* Replaces Scene.render_forward (render.py:159-173) = render_{path,direct,collocated}_kernel
 * (integrator.py:9-30).  material: DEVICE (tex_h, tex_w, 4) float32; image: DEVICE (H, W, 4)
 * float32, pixels of the shard are overwritten with (sum over the sample range / spp,
 * (sample_end - sample_begin) / spp). */
int zdr_render_forward(zdr_scene *scene, const zdr_render_params *params, const float *material,
                       float *image, void *stream);

/* Replaces Scene.render_backward (render.py:176-199) = render_*_backward_kernel
 * (integrator.py:33-53): d_material (DEVICE, tex_h x tex_w x 4) is ACCUMULATED into (+=), as the
 * reference's atomic_fetch_add does (interaction.py:63-70); the caller zeroes it (render.py:220).
 * d_image: DEVICE (H, W, 4) cotangent of the image. */
int zdr_render_backward(zdr_scene *scene, const zdr_render_params *params, const float *d_image,
                        const float *material, float *d_material, void *stream);

/* Path statistics of one forward pass over the shard: counters[8] (HOST, written
 * after an internal synchronise) = camera samples, closest-hit rays, closest rays that hit,
 * shadow rays (one per shaded vertex, prb.py:59), shaded vertices, emitter hits via BSDF sampling, NaN-dropped samples,
 * shadow rays actually traced (the BVH kernels skip those whose light sample carries nothing). */
int zdr_render_stats(zdr_scene *scene, const zdr_render_params *params, const float *material,
                     uint64_t counters[8], void *stream);

/* The kernels carry watchdogs that can end work early instead of spinning on the GPU (a persistent wave that makes
 * no progress; a BVH walk past its iteration budget).  They never trip on valid inputs; if one does it sets a sticky
 * bit in a device error word, and the images / gradients produced since the last check are INCOMPLETE.
 * zdr_scene_check synchronises `stream`, returns ZDR_E_HIP (message in zdr_last_error) if the word is set, and
 * clears it.  zdr_render_stats checks by itself; with the environment variable ZDR_CHECK=1 every render call does
 * (one synchronise per call).  The reference has no counterpart: LuisaCompute raises from luisa.synchronize()
 * (render.py:172,198). */
int zdr_scene_check(zdr_scene *scene, void *stream);

/* LuisaCompute Accel.trace_closest / trace_any (prb.py:25,59) as batch queries, for testing the
 * acceleration structure.  DEVICE rays: n x 8 {o[3], tmin, d[3], tmax}.
 * inst_prim: n x 2 int32 (-1,-1 on miss); bary_t: n x 3 float32 {u, v, t}; occluded: n int32. */
int zdr_trace_closest(zdr_scene *scene, const float *rays, uint32_t n, int32_t *inst_prim, float *bary_t, void *stream);
int zdr_trace_any(zdr_scene *scene, const float *rays, uint32_t n, int32_t *occluded, void *stream);

/* Sampler values as the kernels draw them, for bit-exact comparison with the oracle
 * (BASELINE.json: "sample indices bit-exact").  For each of n queries {px, py, sample_index}
 * (DEVICE int32 n x 3) writes 2 + 8*nvert float32 to out (DEVICE, stride 2 + 8*nvert): next2f,
 * then per vertex next,next,next2f,next,next2f and — for vertices k >= rr_depth — next
 * (SURVEY App. A.8); unused slots are 0. */
int zdr_sampler_dump(zdr_scene *scene, int32_t sampler, uint32_t seed, uint32_t spp, const int32_t *queries,
                     uint32_t n, int32_t nvert, int32_t rr_depth, float *out, void *stream);

/* The same draws, produced THE WAY THE PATH KERNELS PRODUCE THEM: on every BASELINE configuration (CMJ, power-of-two spp
 * <= 65536 and strata grid) the path kernels do not call next() / next2f() one by one (corrmj.py:95-117) but draw a
 * vertex's seven numbers at once with two Kensler permutations per register (csrc/sampler.h, cmj_vertex_samples) and the
 * Russian-roulette number from an index permuted alongside (cmj_next_with_index).  This entry point runs exactly that code
 * — and the one-by-one calls where the kernels fall back to them — so that "sample indices bit-exact" is asserted on the
 * instructions the renders execute.  `integrator` = ZDR_PATH or ZDR_DIRECT: the direct kernels pack the PIXEL's 2-D draw
 * (an index pass + one packed pass for its two strata, csrc/sampler.h cmj_next2_packed) and draw the vertex's numbers one
 * by one; the path kernels draw the pixel's numbers one by one and pack the vertex's (each kernel keeps the form that
 * its register budget pays for: csrc/integrators.h, pixel_ray).  Other arguments and the output
 * layout as zdr_sampler_dump; *batched (HOST, may be NULL) receives 1 when the packed route was taken, 0 for the fallback. */
int zdr_vertex_sampler_dump(zdr_scene *scene, int32_t integrator, int32_t sampler, uint32_t seed, uint32_t spp,
                            const int32_t *queries, uint32_t n, int32_t nvert, int32_t rr_depth, float *out,
                            int32_t *batched, void *stream);

/* Per-path traces of the path integrator, for comparing the kernels with the oracle PATH BY PATH (glossy materials
 * amplify last-ulp differences, so whole-image statistics alone cannot tell a flipped branch from wrong arithmetic).
 * For each of n queries {px, py, sample_index} (DEVICE int32 n x 3) the camera sample is walked with the device
 * functions of the path kernels (prb.py:19-88) and swept like the backward kernel (prb.py:92-187); `params` as for
 * zdr_render_backward (its seed is used as it is), d_image the cotangent (DEVICE, or NULL = ones).
 * out (DEVICE float32, n x (8 + 24 maxv), 1 <= maxv <= 16):
 *   header  {bits(nvert), L.rgb — the sample's radiance before the clamp of integrator.py:26 —, 0, Li.rgb of the emitter that ended it}
 *   vertex  {bits(inst), bits(prim), uv.xy, bits(flags), pdf_bsdf, wi.xyz (world), beta.rgb leaving the vertex,
 *            grad.rgba (what the backward pass scatters at uv for this vertex), NEE radiance.rgb, 0 x 5}
 *   flags = light sample accepted | path went on << 1 | Russian roulette kind << 2 (0 none, 1 stochastic, 2 renormalising).
 * A query whose pixel lies outside the image or whose sample_index >= spp yields an all-zero row. */
int zdr_path_dump(zdr_scene *scene, const zdr_render_params *params, const float *material, const float *d_image,
                  const int32_t *queries, uint32_t n, int32_t maxv, float *out, void *stream);

/* Host-only: builds the acceleration structure exactly as zdr_scene_create does and returns it,
 * without touching a GPU, so that the CPU test-suite can run an emulation of the device traversal
 * on the very data the kernels read (tests/test_bvh_emulation.py).  tri_xyz: ntris x 9 world-space
 * corners.  nodes_out: nodes_cap x 16 floats (BVH4 node = 64 bytes, layout in csrc/scene.h);
 * order_out[slot] = input triangle; isect_out: ntris x 12 floats (plane-form records, slot order).
 * Brute force (ZDR_ACCEL_BRUTE, or AUTO with <= 64 triangles): there are no nodes; *nnodes receives the number Q of
 * planar convex quads the walk merged out of coplanar triangle pairs — slots 2q and 2q + 1 for q < Q, the other
 * triangles after them — and the records of a quad's two triangles start at the corner OPPOSITE the shared edge;
 * *stack_entries receives how many of the quads are parallelograms (they come first; the walk tests them in pairs with
 * the first triangle's u, v and 1 - u, 1 - v) (tests/test_brute_quads.py).  (zdr_scene_create additionally orders the
 * primitives WITHIN each of these groups so that those a shadow segment can meet come first — zdr_debug_never_occluders —
 * which needs the lights and is not reproduced here.) */
int zdr_debug_build_accel(const float *tri_xyz, uint32_t ntris, int accel, float *nodes_out, uint32_t nodes_cap,
                          uint32_t *nnodes, uint32_t *stack_entries, int32_t *order_out, float *isect_out);

/* Host-only: which triangles can never lie between a surface point of the scene and a point of a light, i.e. can be left out of
 * the shadow-segment walk of next-event estimation (prb.py:57-59; csrc/zdr_api.cpp, never_occluders: the triangle's plane supports
 * the whole scene and the lights keep a distance from it that makes a hit inside (tmin, tmax) impossible).  The brute-force accel
 * skips the pairs of the pair walk whose primitives all qualify.  tri_xyz: ntris x 9 world-space corners; is_light_tri: ntris flags;
 * never_out: ntris flags (tests/test_brute_quads.py). */
int zdr_debug_never_occluders(const float *tri_xyz, uint32_t ntris, const uint8_t *is_light_tri, uint8_t *never_out);

#ifdef __cplusplus
}
#endif
#endif
