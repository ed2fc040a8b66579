/* zdr_oracle.h — CPU oracle for the zdr render()/PRB hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under zdr_amd/ may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / reported baseline.
 *
 * PARITY UNPINNED at the LuisaCompute boundary: the reference (111116/zdr)
 * cannot be imported here (no `luisa`, needs CUDA), ships no golden vectors and
 * no asserting tests.  This file is a plain-C restatement
 * of the reference's Python/LuisaCompute-DSL source, function by function, with
 * file:line citations into /root/reference.  It is pinned instead by closed-form
 * known-answer tests, float64 NumPy re-evaluations and finite differences
 * (tests/test_oracle_*.py).  (The one reference module that does run here, load_obj.py, has no counterpart
 * in this file: it pins the product's OBJ ingest directly, tests/golden/obj_fixtures.npz.)
 */
#ifndef ZDR_ORACLE_H
#define ZDR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ZDRO_COLLOCATED = 0, ZDRO_DIRECT = 1, ZDRO_PATH = 2, ZDRO_UVGRAD = 3 /* render_duvdxy, uvgrad.py */ };
enum { ZDRO_SAMPLER_CMJ = 0, ZDRO_SAMPLER_PMJ02BN = 1 };
/* PRB adjoint form: CORRECT = derivative of the forward's expectation (SURVEY App. A.7 plus the
 * Russian-roulette renormalisation terms, see path_backward); LITERAL = the weight of prb.py:162;
 * DETACHED = App. A.7 with every RR factor held constant.  The last two only document deviations. */
enum { ZDRO_PRB_CORRECT = 0, ZDRO_PRB_LITERAL = 1, ZDRO_PRB_DETACHED = 2 };

typedef struct zdro_scene zdro_scene;

/* Scene = list of mesh instances, like Scene.load_geometry (render.py:73-128).
 * verts: nverts x 8 floats {v[3], vt[2], vn[3]} in OBJECT space (vertex.py:4).
 * tris: ntris x 3 vertex indices (global into verts).
 * inst_tri_begin[i] .. inst_tri_begin[i+1]: triangles of instance i (ninst+1 entries).
 * inst_xform: ninst x 16, row-major 4x4 object->world.
 * inst_emission: ninst x 3. */
zdro_scene *zdro_scene_create(const float *verts, int nverts, const int32_t *tris, int ntris,
                              const int32_t *inst_tri_begin, const float *inst_xform,
                              const float *inst_emission, int ninst);
void zdro_scene_destroy(zdro_scene *);
void zdro_scene_set_emissions(zdro_scene *, const float *inst_emission); /* render.py:130-148 */
/* Scene.add_envmap (render.py:150-156): texture (tex_h x tex_w x 4) + the tables of zdr_amd/envmap.py; tex NULL = none */
void zdro_scene_set_envmap(zdro_scene *, const float *tex, int tex_h, int tex_w, const float *alias_prob,
                           const int32_t *alias_idx, int n_alias, const float *pdf, int map_w, int map_h);

/* Optional pbrt-v4 style tables for the PMJ02bn sampler (pmj02bn.py:9-18).
 * pmj: [nsets][nsamples][2] uint32 (value / 2^32); bn: [ntex][res][res] uint16 (/2^16).
 * The reference's own tables are absent (.MISSING_LARGE_BLOBS). Pointers are borrowed. */
void zdro_set_pmj02bn_tables(const uint32_t *pmj, int nsets, int nsamples,
                             const uint16_t *bn, int ntex, int bnres);

typedef struct {
    int32_t integrator, sampler;
    int32_t width, height;
    uint32_t spp, seed;
    int32_t use_tent;
    int32_t x0, y0, x1, y1;            /* pixel rect, [x0,x1) x [y0,y1) */
    uint32_t sample_begin, sample_end; /* sample index range inside [0,spp) */
    int32_t max_depth, rr_depth;       /* prb.py:15-16 (16, 2) */
    int32_t prb_mode;
    float cam_fov, cam_origin[3], cam_target[3], cam_up[3]; /* render.py:28 */
    int32_t tex_h, tex_w;
    int32_t nthreads;                  /* OpenMP threads; <=0 = default */
} zdro_params;

/* counters[0..7]: camera samples, closest-hit rays, closest rays that hit,
 * shadow rays, shaded vertices, emitter hits via BSDF sampling (depth>0),
 * NaN-dropped samples, gradient scatters. */
int zdro_render_forward(const zdro_scene *, const zdro_params *, const float *material,
                        float *image /* H x W x 4 */, uint64_t *counters /* 8 or NULL */);
/* d_material (tex_h x tex_w x 4) is ACCUMULATED into (+=), float64 internally. */
int zdro_render_backward(const zdro_scene *, const zdro_params *, const float *d_image,
                         const float *material, float *d_material, uint64_t *counters);

/* Per-path trace of the path integrator (twin of zdr_path_dump, include/zdr.h; layout in zdr_oracle.c). */
int zdro_path_dump(const zdro_scene *, const zdro_params *, const float *material, const float *d_image /* or NULL */,
                   const int32_t *queries /* n x 3 */, int n, int maxv, float *out /* n x (8 + 24 maxv) */);

/* Batch ray queries (LuisaCompute Accel.trace_closest / trace_any).
 * rays: n x 8 {o[3], tmin, d[3], tmax}; hits: n x 4 {inst, prim, as-float u, v}, t in tout. */
/* The oracle searches scenes above 256 triangles through its own binary BVH (same triangle test, same answer as the loop over
 * every triangle: smallest t, then smallest triangle index); 1 = loop over every triangle instead (tests compare the two). */
void zdro_debug_force_brute(int on);
void zdro_trace_closest(const zdro_scene *, const float *rays, int n, int32_t *inst_prim /* n x 2 */,
                        float *bary_t /* n x 3: u, v, t */);
void zdro_trace_any(const zdro_scene *, const float *rays, int n, int32_t *occluded);

/* Known-answer hooks for the unit tests. */
uint32_t zdro_xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w);
uint32_t zdro_permutation_element(uint32_t i, uint32_t l, uint32_t w, uint32_t p);
void zdro_cmj_grid(uint32_t spp, uint32_t *resx, uint32_t *resy);
/* Draw the sampler sequence a path of `nvert` shaded vertices consumes
 * (SURVEY App. A.8): next2f, then per vertex next,next,next2f,next,next2f,(next if k>=rr_depth).
 * out receives the floats in draw order; returns count. */
int zdro_sampler_dump(int sampler, int px, int py, uint32_t seed, uint32_t spp, uint32_t sample_index,
                      int nvert, int rr_depth, float *out);
void zdro_ggx_brdf(const float wo[3], const float wi[3], const float diffuse[3], float roughness, float out[3]);
float zdro_ggx_sample_pdf(const float wo[3], const float wi[3], float roughness);
void zdro_ggx_sample(const float wo[3], float roughness, float u_lobe, const float u2[2], float out[3]);
/* reverse-mode derivative of ggx_brdf w.r.t. (diffuse rgb, roughness) for cotangent g */
void zdro_ggx_brdf_grad(const float wo[3], const float wi[3], const float diffuse[3], float roughness,
                        const float g[3], float out[4]);
float zdro_ggx_dlnpdf_dr(const float wo[3], const float wi[3], float r);
void zdro_generate_ray(const zdro_params *, float px, float py, float o[3], float d[3]);
void zdro_offset_ray_origin(const float p[3], const float n[3], float out[3]);
void zdro_read_bsdf(const float *material, int tex_h, int tex_w, float u, float v, float out[4]);

#ifdef __cplusplus
}
#endif
#endif
