/* zdr_oracle.c — CPU oracle for the zdr render()/PRB hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see zdr_oracle.h).  PARITY UNPINNED at the
 * LuisaCompute boundary: the reference cannot run here and holds no golden
 * vectors; this restatement is pinned by tests/test_oracle_*.py instead.
 *
 * Every function cites the reference file:line (under /root/reference) it
 * restates.  Arithmetic is float32 with IEEE semantics (compile with
 * -ffp-contract=off, no fast-math), integers are uint32 with wrap-around
 * (SURVEY.md App. A.9, App. B-6/B-7).
 *
 * Third-party pieces the reference gets from LuisaCompute (luisa-python,
 * unpinned, environment.yml:11-12) and that are restated from their published
 * behaviour: ray/triangle intersection (two-sided, tmin < t < tmax, evaluated
 * in plane form — see tri_planes below), Hit::interpolate ((1-u-v)a + ub + vc), offset_ray_origin
 * (Waechter & Binder, Ray Tracing Gems ch. 6), reverse-mode autodiff of
 * ggx_brdf (hand-written tape below), float atomic add (here: float64 sums).
 */
#include "zdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
static inline int vany_nan(v3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }

#define PI_F 3.14159265358979323846f

/* ------------------------------------------------------------------- scene */
struct zdro_scene {
    int nverts, ntris, ninst;
    float *verts;       /* nverts x 8, object space (vertex.py:4) */
    int32_t *tris;      /* ntris x 3 */
    int32_t *tri_begin; /* ninst + 1 */
    float *xform;       /* ninst x 16 row-major */
    float *nmat;        /* ninst x 9: inverse(transpose(M3x3)) (interaction.py:28) */
    float *emission;    /* ninst x 3 (heap slot 23333, render.py:120-123) */
    int32_t *light_insts; /* heap slot 23334 (render.py:121) */
    int light_count;
    int32_t *tri_inst;  /* instance of each triangle */
    v3 *wp;             /* ntris x 3 world-space corner positions */
    float *planes;      /* ntris x 12: {n, n.p0} {nu, du} {nv, dv} (tri_planes) */
    /* environment light (envmap.py): heap slots 23330 (alias tables), 23331 (pdf), 23332 (texture) */
    int env_count, env_h, env_w, map_w, map_h;
    float *env_tex;     /* env_h x env_w x 4 */
    float *alias_prob; int32_t *alias_idx;   /* [map_h] marginal p(y), then map_h tables of map_w: p(x|y) */
    float *env_pdf;     /* map_h x map_w */
    /* Search structure of the ORACLE's own ray queries (scenes above ZDRO_BVH_MIN_TRIS triangles): a plain binary BVH
     * over the same plane records.  It changes WHICH triangles are tested, never the test or the answer: trace_closest
     * returns the hit of smallest t and, among equal t, of smallest triangle index — exactly what the brute-force loop
     * returns (tests/test_oracle_render.py compares the two bit for bit).  It exists so that the 1 M-triangle scene of
     * BASELINE configs[4] can be checked at useful sizes and timed as a CPU baseline; it mirrors nothing of the
     * reference (LuisaCompute's Accel is third-party) and nothing of the product's BVH4 (zdr_api.cpp). */
    int nbvh; struct zdro_bnode *bvh; int32_t *bvh_tri;   /* nodes; triangle indices in leaf order */
};
struct zdro_bnode { float lo[3], hi[3]; int32_t left, count; };   /* count > 0: leaf over bvh_tri[left .. left + count); else children left, left + 1 */
#define ZDRO_BVH_MIN_TRIS 256
static int g_force_brute = 0;
void zdro_debug_force_brute(int on) { g_force_brute = on; }

static void tri_planes(const v3 *p, float *out);
static void build_bvh(zdro_scene *s);

static v3 xform_point(const float *m, v3 v) {
    /* (transform * float4(v, 1)).xyz — interaction.py:19-21 */
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3],
              m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7],
              m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]);
}

static void normal_matrix(const float *m, float *n) {
    /* inverse(transpose(A)) = cofactor(A) / det(A), A = upper 3x3 (interaction.py:28) */
    float a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
    float c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
    float det = a * c00 + b * c01 + c * c02;
    float inv = 1.0f / det;
    n[0] = c00 * inv; n[1] = c01 * inv; n[2] = c02 * inv;
    n[3] = c10 * inv; n[4] = c11 * inv; n[5] = c12 * inv;
    n[6] = c20 * inv; n[7] = c21 * inv; n[8] = c22 * inv;
}

static void rebuild_lights(zdro_scene *s) {
    /* render.py:89-90,118-121 and update_lights 146-148: a light is any
     * instance with an emission component > 0, in instance order */
    s->light_count = 0;
    for (int i = 0; i < s->ninst; i++) {
        const float *e = s->emission + 3 * i;
        if (e[0] > 0 || e[1] > 0 || e[2] > 0) s->light_insts[s->light_count++] = i;
    }
    for (int i = s->light_count; i < s->ninst; i++) s->light_insts[i] = 0;
}

zdro_scene *zdro_scene_create(const float *verts, int nverts, const int32_t *tris, int ntris,
                              const int32_t *inst_tri_begin, const float *inst_xform,
                              const float *inst_emission, int ninst) {
    zdro_scene *s = (zdro_scene *)calloc(1, sizeof(*s));
    s->nverts = nverts; s->ntris = ntris; s->ninst = ninst;
    s->verts = (float *)malloc(sizeof(float) * 8 * (size_t)nverts);
    memcpy(s->verts, verts, sizeof(float) * 8 * (size_t)nverts);
    s->tris = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)ntris);
    memcpy(s->tris, tris, sizeof(int32_t) * 3 * (size_t)ntris);
    s->tri_begin = (int32_t *)malloc(sizeof(int32_t) * (ninst + 1));
    memcpy(s->tri_begin, inst_tri_begin, sizeof(int32_t) * (ninst + 1));
    s->xform = (float *)malloc(sizeof(float) * 16 * ninst);
    memcpy(s->xform, inst_xform, sizeof(float) * 16 * ninst);
    s->nmat = (float *)malloc(sizeof(float) * 9 * ninst);
    s->emission = (float *)malloc(sizeof(float) * 3 * ninst);
    memcpy(s->emission, inst_emission, sizeof(float) * 3 * ninst);
    s->light_insts = (int32_t *)malloc(sizeof(int32_t) * ninst);
    s->tri_inst = (int32_t *)malloc(sizeof(int32_t) * (size_t)ntris);
    s->wp = (v3 *)malloc(sizeof(v3) * 3 * (size_t)ntris);
    s->planes = (float *)malloc(sizeof(float) * 12 * (size_t)ntris);
    for (int i = 0; i < ninst; i++) {
        normal_matrix(s->xform + 16 * i, s->nmat + 9 * i);
        for (int t = s->tri_begin[i]; t < s->tri_begin[i + 1]; t++) {
            s->tri_inst[t] = i;
            for (int k = 0; k < 3; k++) {
                const float *v = s->verts + 8 * (size_t)s->tris[3 * (size_t)t + k];
                s->wp[3 * (size_t)t + k] = xform_point(s->xform + 16 * i, V3(v[0], v[1], v[2]));
            }
            tri_planes(s->wp + 3 * (size_t)t, s->planes + 12 * (size_t)t);
        }
    }
    rebuild_lights(s);
    if (ntris > ZDRO_BVH_MIN_TRIS) build_bvh(s);
    return s;
}

void zdro_scene_destroy(zdro_scene *s) {
    if (!s) return;
    free(s->verts); free(s->tris); free(s->tri_begin); free(s->xform); free(s->nmat);
    free(s->emission); free(s->light_insts); free(s->tri_inst); free(s->wp); free(s->planes);
    free(s->env_tex); free(s->alias_prob); free(s->alias_idx); free(s->env_pdf); free(s->bvh); free(s->bvh_tri); free(s);
}

/* Scene.add_envmap / load_envmap (render.py:150-156, envmap.py:116-203): the tables come from the host
 * (zdr_amd/envmap.py); tex == NULL removes the environment. */
void zdro_scene_set_envmap(zdro_scene *s, const float *tex, int tex_h, int tex_w, const float *alias_prob,
                           const int32_t *alias_idx, int n_alias, const float *pdf, int map_w, int map_h) {
    free(s->env_tex); free(s->alias_prob); free(s->alias_idx); free(s->env_pdf);
    s->env_tex = 0; s->alias_prob = 0; s->alias_idx = 0; s->env_pdf = 0; s->env_count = 0;
    if (!tex) return;
    s->env_h = tex_h; s->env_w = tex_w; s->map_w = map_w; s->map_h = map_h;
    s->env_tex = (float *)malloc(sizeof(float) * 4 * (size_t)tex_h * tex_w); memcpy(s->env_tex, tex, sizeof(float) * 4 * (size_t)tex_h * tex_w);
    s->alias_prob = (float *)malloc(sizeof(float) * n_alias); memcpy(s->alias_prob, alias_prob, sizeof(float) * n_alias);
    s->alias_idx = (int32_t *)malloc(sizeof(int32_t) * n_alias); memcpy(s->alias_idx, alias_idx, sizeof(int32_t) * n_alias);
    s->env_pdf = (float *)malloc(sizeof(float) * (size_t)map_w * map_h); memcpy(s->env_pdf, pdf, sizeof(float) * (size_t)map_w * map_h);
    s->env_count = 1;
}

void zdro_scene_set_emissions(zdro_scene *s, const float *e) {
    memcpy(s->emission, e, sizeof(float) * 3 * s->ninst);
    rebuild_lights(s);
}

/* ------------------------------------------------- ray / triangle (LC Accel) */
typedef struct { v3 o; float tmin; v3 d; float tmax; } ray_t;
typedef struct { int inst, prim; float u, v, t; } hit_t; /* inst < 0: miss */

/* Ray/triangle test in plane form (Havel & Herout 2010 style).  Per triangle, from the float32
 * world-space corners, in float64 and rounded once to float32:
 *   n = e1 x e2,  N = (n, n.p0)            t = (N.w - n.o) / (n.d)
 *   nu = (e2 x n) / |n|^2, du = -nu.p0     u = nu.p + du      with p = o + t d
 *   nv = (n x e1) / |n|^2, dv = -nv.p0     v = nv.p + dv
 * Two-sided; a hit needs tmin < t < tmax, u >= 0, v >= 0, u + v <= 1.  LuisaCompute's own
 * intersector (OptiX on its cuda backend) is third-party and unpinned; this is its restatement. */
static void tri_planes(const v3 *p, float *out) {
    double p0[3] = {p[0].x, p[0].y, p[0].z};
    double e1[3] = {(double)p[1].x - p0[0], (double)p[1].y - p0[1], (double)p[1].z - p0[2]};
    double e2[3] = {(double)p[2].x - p0[0], (double)p[2].y - p0[1], (double)p[2].z - p0[2]};
    double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    double nu[3] = {(e2[1] * n[2] - e2[2] * n[1]) / nn, (e2[2] * n[0] - e2[0] * n[2]) / nn, (e2[0] * n[1] - e2[1] * n[0]) / nn};
    double nv[3] = {(n[1] * e1[2] - n[2] * e1[1]) / nn, (n[2] * e1[0] - n[0] * e1[2]) / nn, (n[0] * e1[1] - n[1] * e1[0]) / nn};
    out[0] = (float)n[0]; out[1] = (float)n[1]; out[2] = (float)n[2]; out[3] = (float)(n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
    out[4] = (float)nu[0]; out[5] = (float)nu[1]; out[6] = (float)nu[2]; out[7] = (float)-(nu[0] * p0[0] + nu[1] * p0[1] + nu[2] * p0[2]);
    out[8] = (float)nv[0]; out[9] = (float)nv[1]; out[10] = (float)nv[2]; out[11] = (float)-(nv[0] * p0[0] + nv[1] * p0[1] + nv[2] * p0[2]);
}

static inline int tri_intersect(const float *q, const ray_t *r, float tmax, float *t, float *u, float *v) {
    float nd = q[0] * r->d.x + q[1] * r->d.y + q[2] * r->d.z;
    float tn = q[3] - (q[0] * r->o.x + q[1] * r->o.y + q[2] * r->o.z);
    float tt = tn / nd;
    if (!(tt > r->tmin && tt < tmax)) return 0;
    v3 p = vadd(r->o, vscale(r->d, tt));
    float uu = q[4] * p.x + q[5] * p.y + q[6] * p.z + q[7];
    float vv = q[8] * p.x + q[9] * p.y + q[10] * p.z + q[11];
    if (!(uu >= 0.0f && vv >= 0.0f && uu + vv <= 1.0f)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

/* ---- the oracle's own search structure (see struct zdro_scene) ---- */
typedef struct { float c[3]; int32_t tri; } bprim_t;
static __thread int g_sort_axis;   /* per thread: ctypes releases the GIL, two scenes may be built at once */
static int cmp_prim(const void *a, const void *b) {
    float x = ((const bprim_t *)a)->c[g_sort_axis], y = ((const bprim_t *)b)->c[g_sort_axis];
    return (x > y) - (x < y);
}
static void tri_bounds(const zdro_scene *s, int t, float *lo, float *hi) {
    for (int a = 0; a < 3; a++) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
    for (int k = 0; k < 3; k++) {
        const v3 p = s->wp[3 * (size_t)t + k]; const float c[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], c[a]); hi[a] = fmaxf(hi[a], c[a]); }
    }
}
/* median split along the longest axis of the centroid bounds; leaves of <= 4 triangles; iterative (explicit work list) */
static void build_bvh(zdro_scene *s) {
    const int n = s->ntris;
    bprim_t *pr = (bprim_t *)malloc(sizeof(bprim_t) * (size_t)n);
    for (int t = 0; t < n; t++) {
        float lo[3], hi[3]; tri_bounds(s, t, lo, hi);
        for (int a = 0; a < 3; a++) pr[t].c[a] = 0.5f * (lo[a] + hi[a]);
        pr[t].tri = t;
    }
    s->bvh = (struct zdro_bnode *)malloc(sizeof(struct zdro_bnode) * (size_t)(2 * n));
    int (*work)[3] = (int (*)[3])malloc(sizeof(int[3]) * 128);   /* node, first, count */
    int nw = 0, nn = 1;
    work[nw][0] = 0; work[nw][1] = 0; work[nw][2] = n; nw++;
    while (nw) {
        nw--; const int node = work[nw][0], first = work[nw][1], count = work[nw][2];
        struct zdro_bnode *b = &s->bvh[node];
        if (count <= 4) { b->left = first; b->count = count; continue; }
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int i = first; i < first + count; i++)
            for (int a = 0; a < 3; a++) { clo[a] = fminf(clo[a], pr[i].c[a]); chi[a] = fmaxf(chi[a], pr[i].c[a]); }
        int ax = 0; if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1; if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        g_sort_axis = ax;
        qsort(pr + first, (size_t)count, sizeof(bprim_t), cmp_prim);
        const int half = count / 2;
        b->left = nn; b->count = 0;
        work[nw][0] = nn; work[nw][1] = first; work[nw][2] = half; nw++;
        work[nw][0] = nn + 1; work[nw][1] = first + half; work[nw][2] = count - half; nw++;
        nn += 2;
    }
    free(work);
    s->nbvh = nn;
    s->bvh_tri = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n; i++) s->bvh_tri[i] = pr[i].tri;
    free(pr);
    /* boxes bottom-up (children always have larger indices than their parent), padded so that the slab test below can
     * never reject a ray that the plane-form triangle test accepts: the accepted hit point lies within rounding of the
     * triangle, the pad is four orders of magnitude above float32 rounding of the scene's coordinates */
    float ext = 0.0f;
    for (size_t i = 0; i < 3 * (size_t)n; i++) ext = fmaxf(ext, fmaxf(fabsf(s->wp[i].x), fmaxf(fabsf(s->wp[i].y), fabsf(s->wp[i].z))));
    const float pad = 1e-4f * (ext > 0.0f ? ext : 1.0f);
    for (int node = nn - 1; node >= 0; node--) {
        struct zdro_bnode *b = &s->bvh[node];
        for (int a = 0; a < 3; a++) { b->lo[a] = 3.0e38f; b->hi[a] = -3.0e38f; }
        if (b->count > 0) {
            for (int i = b->left; i < b->left + b->count; i++) {
                float lo[3], hi[3]; tri_bounds(s, s->bvh_tri[i], lo, hi);
                for (int a = 0; a < 3; a++) { b->lo[a] = fminf(b->lo[a], lo[a] - pad); b->hi[a] = fmaxf(b->hi[a], hi[a] + pad); }
            }
        } else {
            for (int c = 0; c < 2; c++)
                for (int a = 0; a < 3; a++) { b->lo[a] = fminf(b->lo[a], s->bvh[b->left + c].lo[a]); b->hi[a] = fmaxf(b->hi[a], s->bvh[b->left + c].hi[a]); }
        }
    }
}
/* conservative slab test: the ray's [tmin, tmax] against the padded box; a NaN (0 * inf) never rejects */
static inline int box_hit(const struct zdro_bnode *b, const ray_t *r, const float *inv, float tmax) {
    float t0 = r->tmin, t1 = tmax;
    const float o[3] = {r->o.x, r->o.y, r->o.z};
    for (int a = 0; a < 3; a++) {
        float ta = (b->lo[a] - o[a]) * inv[a], tb = (b->hi[a] - o[a]) * inv[a];
        if (ta > tb) { float x = ta; ta = tb; tb = x; }
        if (ta > t0) t0 = ta;                       /* comparisons with NaN are false: the axis is ignored */
        if (tb * 1.0000005f < t1) t1 = tb * 1.0000005f;
    }
    return !(t0 > t1);
}

static hit_t trace_closest(const zdro_scene *s, const ray_t *r) {
    hit_t h; h.inst = -1; h.prim = -1; h.u = h.v = 0; h.t = r->tmax;
    if (!s->bvh || g_force_brute) {
        for (int t = 0; t < s->ntris; t++) {
            float tt, u, v;
            if (tri_intersect(s->planes + 12 * (size_t)t, r, h.t, &tt, &u, &v)) {
                h.t = tt; h.u = u; h.v = v; h.inst = s->tri_inst[t]; h.prim = t - s->tri_begin[h.inst];
            }
        }
        return h;
    }
    /* same answer through the search structure: smallest t, and among equal t the smallest triangle index (= the first
     * one the loop above would have kept) */
    const float inv[3] = {1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z};
    int stack[128], sp = 0, best = -1;
    stack[sp++] = 0;
    while (sp) {
        const struct zdro_bnode *b = &s->bvh[stack[--sp]];
        if (!box_hit(b, r, inv, h.t)) continue;
        if (b->count > 0) {
            for (int i = b->left; i < b->left + b->count; i++) {
                const int t = s->bvh_tri[i];
                float tt, u, v;
                /* tri_intersect wants tt < tmax: ask with the next float above the best t so that ties reach the index test */
                if (tri_intersect(s->planes + 12 * (size_t)t, r, best < 0 ? h.t : nextafterf(h.t, 3.0e38f), &tt, &u, &v) &&
                    (best < 0 || tt < h.t || t < best)) {
                    h.t = tt; h.u = u; h.v = v; best = t;
                }
            }
        } else if (sp + 2 <= 128) { stack[sp++] = b->left; stack[sp++] = b->left + 1; }
    }
    if (best >= 0) { h.inst = s->tri_inst[best]; h.prim = best - s->tri_begin[h.inst]; }
    return h;
}

static int trace_any(const zdro_scene *s, const ray_t *r) {
    if (!s->bvh || g_force_brute) {
        for (int t = 0; t < s->ntris; t++) {
            float tt, u, v;
            if (tri_intersect(s->planes + 12 * (size_t)t, r, r->tmax, &tt, &u, &v)) return 1;
        }
        return 0;
    }
    const float inv[3] = {1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z};
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const struct zdro_bnode *b = &s->bvh[stack[--sp]];
        if (!box_hit(b, r, inv, r->tmax)) continue;
        if (b->count > 0) {
            for (int i = b->left; i < b->left + b->count; i++) {
                float tt, u, v;
                if (tri_intersect(s->planes + 12 * (size_t)s->bvh_tri[i], r, r->tmax, &tt, &u, &v)) return 1;
            }
        } else if (sp + 2 <= 128) { stack[sp++] = b->left; stack[sp++] = b->left + 1; }
    }
    return 0;
}

void zdro_trace_closest(const zdro_scene *s, const float *rays, int n, int32_t *inst_prim, float *bary_t) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        const float *q = rays + 8 * (size_t)i;
        ray_t r; r.o = V3(q[0], q[1], q[2]); r.tmin = q[3]; r.d = V3(q[4], q[5], q[6]); r.tmax = q[7];
        hit_t h = trace_closest(s, &r);
        inst_prim[2 * i] = h.inst; inst_prim[2 * i + 1] = h.prim;
        bary_t[3 * i] = h.u; bary_t[3 * i + 1] = h.v; bary_t[3 * i + 2] = h.inst < 0 ? r.tmax : h.t;
    }
}

void zdro_trace_any(const zdro_scene *s, const float *rays, int n, int32_t *occ) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        const float *q = rays + 8 * (size_t)i;
        ray_t r; r.o = V3(q[0], q[1], q[2]); r.tmin = q[3]; r.d = V3(q[4], q[5], q[6]); r.tmax = q[7];
        occ[i] = trace_any(s, &r);
    }
}

/* LuisaCompute offset_ray_origin (Waechter & Binder, RT Gems ch.6): recalled
 * from upstream LC, unpinned. prb.py:75, direct.py:64 */
static v3 offset_ray_origin(v3 p, v3 n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    float pc[3] = {p.x, p.y, p.z}, nc[3] = {n.x, n.y, n.z}, out[3];
    for (int k = 0; k < 3; k++) {
        int32_t of_i = (int32_t)(int_scale * nc[k]);
        int32_t pi; memcpy(&pi, &pc[k], 4);
        pi += pc[k] < 0.0f ? -of_i : of_i;
        float p_i; memcpy(&p_i, &pi, 4);
        out[k] = fabsf(pc[k]) < origin ? pc[k] + float_scale * nc[k] : p_i;
    }
    return V3(out[0], out[1], out[2]);
}

void zdro_offset_ray_origin(const float p[3], const float n[3], float out[3]) {
    v3 r = offset_ray_origin(V3(p[0], p[1], p[2]), V3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ---------------------------------------------------------------- samplers */
/* pmj02bn.py:60-74 (unsigned form; corrmj.py:31-44 is the same hash, App. B-7) */
uint32_t zdro_xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    const uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    uint32_t h = w + P5 + x * P3;
    h = P4 * ((h << 17) | (h >> 15));
    h += y * P3;
    h = P4 * ((h << 17) | (h >> 15));
    h += z * P3;
    h = P4 * ((h << 17) | (h >> 15));
    h = P2 * (h ^ (h >> 15));
    h = P3 * (h ^ (h >> 13));
    return h ^ (h >> 16);
}

/* corrmj.py:6-28 / pmj02bn.py:33-57 (Kensler's permute; all-uint32, App. B-6) */
uint32_t zdro_permutation_element(uint32_t i, uint32_t l, uint32_t w, uint32_t p) {
    /* The hash is a bijection of [0, w]; started inside [0, l) the cycle walk re-enters [0, l)
     * after at most w + 1 - l rejected values.  The reference loops `while True` and would spin
     * forever on a start value >= l whose cycle stays outside [0, l) (corrmj.py:109-112 produces such
     * values when spp is not a perfect square); the walk is bounded here by that exact maximum. */
    uint32_t budget = w - l + 2u;
    do {
        i ^= p; i *= 0xe170893du; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8;
        i *= 0x0929eb3fu; i ^= p >> 23; i ^= (i & w) >> 1; i *= 1u | p >> 27;
        i *= 0x6935fa69u; i ^= (i & w) >> 11; i *= 0x74dcb303u; i ^= (i & w) >> 2;
        i *= 0x9e501cc3u; i ^= (i & w) >> 2; i *= 0xc860a3dfu; i &= w; i ^= i >> 5;
    } while (i >= l && --budget);
    return (i + p) % l;
}

static const float ONE_MINUS_EPS = 0x1.fffffep-1f; /* corrmj.py:46 */

static const uint32_t *g_pmj; static int g_pmj_nsets, g_pmj_nsamples;
static const uint16_t *g_bn; static int g_bn_ntex, g_bn_res;
void zdro_set_pmj02bn_tables(const uint32_t *pmj, int nsets, int nsamples,
                             const uint16_t *bn, int ntex, int bnres) {
    g_pmj = pmj; g_pmj_nsets = nsets; g_pmj_nsamples = nsamples;
    g_bn = bn; g_bn_ntex = ntex; g_bn_res = bnres;
}

typedef struct {
    int kind;
    uint32_t px, py, sample_index, dimension, seed, spp, w; /* pmj02bn.py:78-85 */
    uint32_t permutation_seed, state;                         /* corrmj.py:48-57 */
    uint32_t resx, resy, reswx, reswy;                         /* 2-D strata grid, resx * resy >= spp */
} sampler_t;

static uint32_t smear(uint32_t w) { w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16; return w; }

/* corrmj.py:67 uses res = int(sqrt(spp + 0.4)) for both axes, which is only meaningful when
 * spp == res * res: otherwise y = index // res reaches values >= res (out of the permutation's
 * domain, u.y > 1).  Generalisation used here and in the HIP kernels: a resx x resy grid with
 * resx * resy >= spp — identical to the reference whenever spp is a perfect square. */
void zdro_cmj_grid(uint32_t spp, uint32_t *resx, uint32_t *resy) {
    uint32_t res = (uint32_t)(int)sqrtf((float)spp + 0.4f);
    if (res < 1) res = 1;
    if (res * res == spp) { *resx = res; *resy = res; return; }
    if ((spp & (spp - 1)) == 0) { /* 2^(2k+1): 2^(k+1) x 2^k tiles the samples exactly */
        uint32_t lg = 0; while ((1u << lg) < spp) lg++;
        *resx = 1u << ((lg + 1) / 2); *resy = spp / *resx; return;
    }
    uint32_t m = res; while (m * m < spp) m++;
    *resx = m; *resy = (spp + m - 1) / m;
}

static sampler_t make_sampler(int kind, int px, int py, uint32_t seed, uint32_t spp, uint32_t sample_index) {
    sampler_t t; memset(&t, 0, sizeof t);
    t.kind = kind; t.px = (uint32_t)px; t.py = (uint32_t)py; t.sample_index = sample_index;
    t.dimension = 0; t.seed = seed; t.spp = spp;
    t.w = smear(spp - 1); /* corrmj.py:61-66, pmj02bn.py:89-94 */
    if (kind == ZDRO_SAMPLER_CMJ) { /* corrmj.py:60-84 */
        zdro_cmj_grid(spp, &t.resx, &t.resy);
        t.reswx = smear(t.resx - 1); t.reswy = smear(t.resy - 1);
        t.permutation_seed = zdro_xxhash32_4(t.px, t.py, seed, 0);
        t.state = zdro_xxhash32_4(t.px, t.py, seed, sample_index);
    }
    return t;
}

static float next_lcg(sampler_t *s) { /* corrmj.py:88-92 */
    s->state = 1664525u * s->state + 1013904223u;
    return (float)(s->state & 0x00ffffffu) * (1.0f / 16777216.0f);
}

static float blue_noise(uint32_t tex, uint32_t x, uint32_t y) {
    /* pmj02bn.py:20-24, with the pbrt-v4 layout (tex*Res + x)*Res + y (App. B-5) */
    uint32_t ti = tex % (uint32_t)g_bn_ntex, cx = x % (uint32_t)g_bn_res, cy = y % (uint32_t)g_bn_res;
    return (float)g_bn[((size_t)ti * g_bn_res + cx) * g_bn_res + cy] * (1.0f / 65536.0f);
}

static float sampler_next(sampler_t *s) {
    if (s->kind == ZDRO_SAMPLER_CMJ) { /* corrmj.py:95-102 */
        uint32_t ps = s->permutation_seed + s->dimension;
        uint32_t index = zdro_permutation_element(s->sample_index, s->spp, s->w, (ps * 0x45fbe943u) & 0x70ffffffu);
        float delta = next_lcg(s);
        float u = ((float)index + delta) / (float)s->spp;
        s->dimension += 1;
        return clampf(u, 0.0f, ONE_MINUS_EPS);
    } else { /* pmj02bn.py:105-112 */
        uint32_t h = zdro_xxhash32_4(s->px, s->py, s->dimension, s->seed);
        uint32_t index = zdro_permutation_element(s->sample_index, s->spp, s->w, h);
        float delta = blue_noise(s->dimension, s->px ^ s->seed, s->py ^ s->seed);
        float u = ((float)index + delta) / (float)s->spp;
        s->dimension += 1;
        return clampf(u, 0.0f, ONE_MINUS_EPS);
    }
}

static v2 sampler_next2(sampler_t *s) {
    v2 u;
    if (s->kind == ZDRO_SAMPLER_CMJ) { /* corrmj.py:105-117 */
        uint32_t ps = s->permutation_seed + s->dimension;
        uint32_t index = zdro_permutation_element(s->sample_index, s->spp, s->w, (ps * 0x51633e2du) & 0x70ffffffu);
        uint32_t y = index / s->resx, x = index % s->resx;
        uint32_t sx = zdro_permutation_element(x, s->resx, s->reswx, (ps * 0x68bc21ebu) & 0x70ffffffu);
        uint32_t sy = zdro_permutation_element(y, s->resy, s->reswy, (ps * 0x02e5be93u) & 0x70ffffffu);
        float dx = next_lcg(s), dy = next_lcg(s);
        float frx = (float)s->resx, fry = (float)s->resy;
        u.x = ((float)x + ((float)sy + dx) / fry) / frx;
        u.y = ((float)y + ((float)sx + dy) / frx) / fry;
        s->dimension += 2;
        u.x = clampf(u.x, 0.0f, ONE_MINUS_EPS); u.y = clampf(u.y, 0.0f, ONE_MINUS_EPS);
    } else { /* pmj02bn.py:115-126 */
        uint32_t index = s->sample_index;
        uint32_t inst = s->dimension / 2;
        if (inst >= (uint32_t)g_pmj_nsets) {
            uint32_t h = zdro_xxhash32_4(s->px, s->py, s->dimension, s->seed);
            index = zdro_permutation_element(s->sample_index, s->spp, s->w, h);
        }
        size_t i = (size_t)(inst % (uint32_t)g_pmj_nsets) * g_pmj_nsamples + index; /* pmj02bn.py:27-30 */
        /* table / 2**32 converted to float32 (pmj02bn.py:9) */
        float tx = (float)((double)g_pmj[2 * i] / 4294967296.0), ty = (float)((double)g_pmj[2 * i + 1] / 4294967296.0);
        float ux = tx + blue_noise(s->dimension, s->px ^ s->seed, s->py ^ s->seed);
        float uy = ty + blue_noise(s->dimension + 1, s->px ^ s->seed, s->py ^ s->seed);
        s->dimension += 2;
        u.x = ux - floorf(ux); u.y = uy - floorf(uy); /* fract */
    }
    return u;
}

int zdro_sampler_dump(int kind, int px, int py, uint32_t seed, uint32_t spp, uint32_t sample_index,
                      int nvert, int rr_depth, float *out) {
    sampler_t s = make_sampler(kind, px, py, seed, spp, sample_index);
    int n = 0; v2 u = sampler_next2(&s); out[n++] = u.x; out[n++] = u.y;
    for (int k = 0; k < nvert; k++) {
        out[n++] = sampler_next(&s); out[n++] = sampler_next(&s);
        u = sampler_next2(&s); out[n++] = u.x; out[n++] = u.y;
        out[n++] = sampler_next(&s);
        u = sampler_next2(&s); out[n++] = u.x; out[n++] = u.y;
        if (k >= rr_depth) out[n++] = sampler_next(&s);
    }
    return n;
}

/* ------------------------------------------------------------------ camera */
static float tent_warp1(float u, float radius) { /* camera.py:20-31 */
    return u < 0.5f ? radius * (sqrtf(2.0f * u) - 1.0f) : radius * (1.0f - sqrtf(2.0f - 2.0f * u));
}

static ray_t generate_ray(const zdro_params *P, float px, float py) { /* camera.py:5-17 */
    v3 origin = V3(P->cam_origin[0], P->cam_origin[1], P->cam_origin[2]);
    v3 target = V3(P->cam_target[0], P->cam_target[1], P->cam_target[2]);
    v3 up = V3(P->cam_up[0], P->cam_up[1], P->cam_up[2]);
    v3 forward = vnormalize(vsub(target, origin));
    v3 right = vnormalize(vcross(forward, up));
    v3 up_perp = vcross(right, forward);
    float tn = tanf(0.5f * P->cam_fov);
    px *= tn; py *= tn;
    v3 dir = vnormalize(vadd(vsub(vscale(right, px), vscale(up_perp, py)), forward));
    ray_t r; r.o = origin; r.d = dir; r.tmin = 0.0f; r.tmax = 1e30f;
    return r;
}

void zdro_generate_ray(const zdro_params *P, float px, float py, float o[3], float d[3]) {
    ray_t r = generate_ray(P, px, py);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}

/* --------------------------------------------------------------------- onb */
typedef struct { v3 tangent, binormal, normal; } onb_t;
static onb_t make_onb(v3 n) { /* onb.py:21-28 */
    onb_t o;
    o.binormal = vnormalize(fabsf(n.x) > fabsf(n.z) ? V3(-n.y, n.x, 0.0f) : V3(0.0f, -n.z, n.y));
    o.tangent = vnormalize(vcross(o.binormal, n));
    o.normal = n;
    return o;
}
static v3 to_world(const onb_t *o, v3 v) { /* onb.py:10-11 */
    return vadd(vadd(vscale(o->tangent, v.x), vscale(o->binormal, v.y)), vscale(o->normal, v.z));
}
static v3 to_local(const onb_t *o, v3 v) { /* onb.py:14-15 */
    return V3(vdot(v, o->tangent), vdot(v, o->binormal), vdot(v, o->normal));
}

/* -------------------------------------------------------------- microfacet */
static float ggx_distribution(v3 h, float alpha) { /* microfacet.py:7-11 */
    float alpha2 = alpha * alpha;
    float nh = fmaxf(0.00001f, h.z);
    float t = nh * nh * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (PI_F * (t * t));
}
static float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
static float fresnel_schlick(float c, float specular) { /* microfacet.py:14-15 */
    return specular + (1.0f - specular) * pow5(1.0f - c);
}
static float smith_geometry(v3 v, float alpha) { /* microfacet.py:18-21 */
    float alpha2 = alpha * alpha;
    float nv = fmaxf(0.00001f, v.z);
    return 2.0f / (1.0f + sqrtf(1.0f + alpha2 * (1.0f - nv * nv) / (nv * nv)));
}
static v3 ggx_brdf(v3 wo, v3 wi, v3 diffuse, float specular, float roughness) { /* microfacet.py:24-30 */
    float alpha = roughness * roughness;
    v3 h = vnormalize(vadd(wi, wo));
    float d = ggx_distribution(h, alpha);
    float f = fresnel_schlick(clampf(vdot(wo, h), 0.00001f, 1.0f), specular);
    float g = smith_geometry(wi, alpha) * smith_geometry(wo, alpha);
    float s = (d * f * g) / (4.0f * fmaxf(0.00001f, wi.z) * fmaxf(0.00001f, wo.z));
    return vscale(vadd(V3(s, s, s), vdivs(diffuse, PI_F)), wi.z);
}
static v3 cosine_sample_hemisphere(v2 u) { /* microfacet.py:34-37 */
    float r = sqrtf(u.x), phi = 2.0f * PI_F * u.y;
    return V3(r * cosf(phi), r * sinf(phi), sqrtf(1.0f - u.x));
}
static float pdf_wm(v3 w, v3 wm, float alpha) { /* microfacet.py:68-69 */
    return smith_geometry(w, alpha) / fabsf(w.z) * ggx_distribution(wm, alpha) * fabsf(vdot(w, wm));
}
static v3 sample_wm(v3 w, float alpha, v2 u) { /* microfacet.py:72-92 (pbrt-v4 VNDF) */
    v3 wh = vnormalize(V3(alpha * w.x, alpha * w.y, w.z));
    if (wh.z < 0) wh = vneg(wh);
    v3 T1 = wh.z < 0.99999f ? vnormalize(vcross(V3(0, 0, 1), wh)) : V3(1, 0, 0);
    v3 T2 = vcross(wh, T1);
    float r = sqrtf(u.x), theta = 2.0f * PI_F * u.y; /* SampleUniformDiskPolar 61-65 */
    float px = r * cosf(theta), py = r * sinf(theta);
    float h = sqrtf(1.0f - px * px);
    py = lerpf(h, py, (1.0f + wh.z) / 2.0f);
    float pz = sqrtf(fmaxf(0.0f, 1.0f - (px * px + py * py)));
    v3 nh = vadd(vadd(vscale(T1, px), vscale(T2, py)), vscale(wh, pz));
    return vnormalize(V3(alpha * nh.x, alpha * nh.y, fmaxf(1e-6f, nh.z)));
}
static v3 ggx_sample_u(v3 wo, float roughness, float u_lobe, v2 u2) { /* microfacet.py:41-49 */
    if (u_lobe < 0.5f) return cosine_sample_hemisphere(u2);
    float alpha = roughness * roughness;
    v3 wm = sample_wm(wo, alpha, u2);
    /* reflect(-wo, wm) = -wo - 2 dot(wm, -wo) wm */
    v3 i = vneg(wo);
    return vsub(i, vscale(wm, 2.0f * vdot(wm, i)));
}
static v3 ggx_sample(v3 wo, float roughness, sampler_t *s) {
    float ul = sampler_next(s);
    v2 u2 = sampler_next2(s); /* both branches draw next2f() after next() */
    return ggx_sample_u(wo, roughness, ul, u2);
}
static float ggx_sample_pdf(v3 wo, v3 wi, float roughness) { /* microfacet.py:52-58 */
    float alpha = roughness * roughness;
    v3 wm = vnormalize(vadd(wi, wo));
    float diffuse_pdf = wi.z / PI_F;
    float glossy_pdf = pdf_wm(wo, wm, alpha) / (4.0f * fabsf(vdot(wo, wm)));
    return 0.5f * diffuse_pdf + 0.5f * glossy_pdf;
}

/* Reverse-mode derivative of ggx_brdf w.r.t. mat = (diffuse.rgb, roughness) for a
 * float3 cotangent g — what the reference obtains from luisa.autodiff
 * (prb.py:138-146,157-163; direct.py:126-131,160-165; collocated.py:44-56).
 * Hand-written tape over the operations of microfacet.py:7-30; max/clamp guards and
 * the Fresnel term do not depend on mat. */
static v4 ggx_brdf_grad(v3 wo, v3 wi, v3 diffuse, float specular, float roughness, v3 g) {
    (void)diffuse;
    float alpha = roughness * roughness, alpha2 = alpha * alpha;
    v3 h = vnormalize(vadd(wi, wo));
    float nh = fmaxf(0.00001f, h.z);
    float t = nh * nh * (alpha2 - 1.0f) + 1.0f;
    float D = alpha2 / (PI_F * (t * t));
    float F = fresnel_schlick(clampf(vdot(wo, h), 0.00001f, 1.0f), specular);
    float nvi = fmaxf(0.00001f, wi.z), nvo = fmaxf(0.00001f, wo.z);
    float ki = (1.0f - nvi * nvi) / (nvi * nvi), ko = (1.0f - nvo * nvo) / (nvo * nvo);
    float si = sqrtf(1.0f + alpha2 * ki), so = sqrtf(1.0f + alpha2 * ko);
    float G1i = 2.0f / (1.0f + si), G1o = 2.0f / (1.0f + so);
    float denom = 4.0f * nvi * nvo;
    /* out_c = (D*F*G1i*G1o/denom + diffuse_c/pi) * wi.z */
    float dS = (g.x + g.y + g.z) * wi.z;          /* adjoint of the specular scalar */
    float dD = dS * F * (G1i * G1o) / denom;
    float dG = dS * D * F / denom;
    float dG1i = dG * G1o, dG1o = dG * G1i;
    /* G1 = 2/(1+s), s = sqrt(1 + alpha2*k) */
    float dsi = dG1i * (-2.0f / ((1.0f + si) * (1.0f + si)));
    float dso = dG1o * (-2.0f / ((1.0f + so) * (1.0f + so)));
    float dalpha2 = dsi * ki / (2.0f * si) + dso * ko / (2.0f * so);
    /* D = alpha2/(pi t^2), t = nh^2 (alpha2-1) + 1 */
    dalpha2 += dD * (1.0f / (PI_F * t * t) - 2.0f * alpha2 * nh * nh / (PI_F * t * t * t));
    float dalpha = dalpha2 * 2.0f * alpha;
    float dr = dalpha * 2.0f * roughness;
    v4 out = {g.x * wi.z / PI_F, g.y * wi.z / PI_F, g.z * wi.z / PI_F, dr};
    return out;
}

/* d ln(ggx_sample_pdf) / d roughness (microfacet.py:52-58): only the glossy half depends on it,
 * glossy = G1(wo) D(wm) / (4 |wo.z|). */
static float ggx_dlnpdf_dr(v3 wo, v3 wi, float roughness) {
    float alpha = roughness * roughness, a2 = alpha * alpha;
    v3 wm = vnormalize(vadd(wi, wo));
    float nh = fmaxf(0.00001f, wm.z), nh2 = nh * nh;
    float tt = nh2 * (a2 - 1.0f) + 1.0f;
    float D = a2 / (PI_F * tt * tt);
    float dD = (1.0f - nh2 * (1.0f + a2)) / (PI_F * tt * tt * tt);
    float nvo = fmaxf(0.00001f, wo.z);
    float ko = (1.0f - nvo * nvo) / (nvo * nvo), so = sqrtf(1.0f + a2 * ko);
    float G1o = 2.0f / (1.0f + so), dG1o = -ko / (so * (1.0f + so) * (1.0f + so));
    float dglossy = (dG1o * D + G1o * dD) / (4.0f * fabsf(wo.z));
    float dp = 0.5f * dglossy * 4.0f * roughness * roughness * roughness;
    return dp / ggx_sample_pdf(wo, wi, roughness);
}

void zdro_ggx_brdf(const float wo[3], const float wi[3], const float d[3], float r, float out[3]) {
    v3 f = ggx_brdf(V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), V3(d[0], d[1], d[2]), 0.04f, r);
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
}
float zdro_ggx_sample_pdf(const float wo[3], const float wi[3], float r) {
    return ggx_sample_pdf(V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), r);
}
void zdro_ggx_sample(const float wo[3], float r, float ul, const float u2[2], float out[3]) {
    v2 u = {u2[0], u2[1]};
    v3 w = ggx_sample_u(V3(wo[0], wo[1], wo[2]), r, ul, u);
    out[0] = w.x; out[1] = w.y; out[2] = w.z;
}
void zdro_ggx_brdf_grad(const float wo[3], const float wi[3], const float d[3], float r, const float g[3], float out[4]) {
    v4 q = ggx_brdf_grad(V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), V3(d[0], d[1], d[2]), 0.04f, r, V3(g[0], g[1], g[2]));
    out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
}

/* ------------------------------------------------------------- interaction */
typedef struct { v3 p; v2 uv; v3 ns, ng; } interaction_t; /* interaction.py:6 */

static interaction_t surface_interact(const zdro_scene *s, const hit_t *h) { /* interaction.py:9-30 */
    int t = s->tri_begin[h->inst] + h->prim;
    const float *a = s->verts + 8 * (size_t)s->tris[3 * (size_t)t];
    const float *b = s->verts + 8 * (size_t)s->tris[3 * (size_t)t + 1];
    const float *c = s->verts + 8 * (size_t)s->tris[3 * (size_t)t + 2];
    v3 p0 = s->wp[3 * (size_t)t], p1 = s->wp[3 * (size_t)t + 1], p2 = s->wp[3 * (size_t)t + 2];
    float w0 = 1.0f - h->u - h->v, w1 = h->u, w2 = h->v; /* Hit::interpolate */
    interaction_t it;
    it.p = vadd(vadd(vscale(p0, w0), vscale(p1, w1)), vscale(p2, w2));
    it.uv.x = w0 * a[3] + w1 * b[3] + w2 * c[3];
    it.uv.y = w0 * a[4] + w1 * b[4] + w2 * c[4];
    v3 ns0 = V3(w0 * a[5] + w1 * b[5] + w2 * c[5], w0 * a[6] + w1 * b[6] + w2 * c[6], w0 * a[7] + w1 * b[7] + w2 * c[7]);
    const float *n = s->nmat + 9 * h->inst;
    it.ns = vnormalize(V3(n[0] * ns0.x + n[1] * ns0.y + n[2] * ns0.z,
                          n[3] * ns0.x + n[4] * ns0.y + n[5] * ns0.z,
                          n[6] * ns0.x + n[7] * ns0.y + n[8] * ns0.z));
    it.ng = vnormalize(vcross(vsub(p1, p0), vsub(p2, p0)));
    return it;
}

static v4 read_single_bsdf(const float *m, int tex_h, int tex_w, int x, int y) { /* interaction.py:36-44 */
    x = clampi(x, 0, tex_w - 1); y = clampi(y, 0, tex_h - 1);
    size_t idx = (size_t)x + (size_t)tex_w * y;
    v4 r = {m[idx * 4], m[idx * 4 + 1], m[idx * 4 + 2], m[idx * 4 + 3]};
    return r;
}
static v4 read_bsdf(const float *m, int tex_h, int tex_w, v2 uv) { /* interaction.py:47-60 (bilinear) */
    float px = uv.x * (float)(tex_w - 1), py = (1.0f - uv.y) * (float)(tex_h - 1);
    int ix = (int)px, iy = (int)py;
    float ox = px - (float)ix, oy = py - (float)iy;
    v4 c00 = read_single_bsdf(m, tex_h, tex_w, ix, iy), c01 = read_single_bsdf(m, tex_h, tex_w, ix, iy + 1);
    v4 c10 = read_single_bsdf(m, tex_h, tex_w, ix + 1, iy), c11 = read_single_bsdf(m, tex_h, tex_w, ix + 1, iy + 1);
    v4 r;
    r.x = lerpf(lerpf(c00.x, c01.x, oy), lerpf(c10.x, c11.x, oy), ox);
    r.y = lerpf(lerpf(c00.y, c01.y, oy), lerpf(c10.y, c11.y, oy), ox);
    r.z = lerpf(lerpf(c00.z, c01.z, oy), lerpf(c10.z, c11.z, oy), ox);
    r.w = lerpf(lerpf(c00.w, c01.w, oy), lerpf(c10.w, c11.w, oy), ox);
    return r;
}
void zdro_read_bsdf(const float *m, int tex_h, int tex_w, float u, float v, float out[4]) {
    v2 uv = {u, v}; v4 r = read_bsdf(m, tex_h, tex_w, uv);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

static int g_atomic_scatter = 0;
static void write_single_bsdf_grad(double *dm, int tex_h, int tex_w, int x, int y, float k, v4 g) { /* interaction.py:63-70 */
    x = clampi(x, 0, tex_w - 1); y = clampi(y, 0, tex_h - 1);
    size_t idx = (size_t)x + (size_t)tex_w * y;
    const double a[4] = {(double)(k * g.x), (double)(k * g.y), (double)(k * g.z), (double)(k * g.w)};
    for (int c = 0; c < 4; c++) {
        if (g_atomic_scatter) {
#pragma omp atomic update
            dm[idx * 4 + c] += a[c];
        } else dm[idx * 4 + c] += a[c];
    }
}
static void write_bsdf_grad(double *dm, int tex_h, int tex_w, v2 uv, v4 g) { /* interaction.py:73-89 */
    float px = uv.x * (float)(tex_w - 1), py = (1.0f - uv.y) * (float)(tex_h - 1);
    int ix = (int)px, iy = (int)py;
    float ox = px - (float)ix, oy = py - (float)iy;
    write_single_bsdf_grad(dm, tex_h, tex_w, ix, iy, (1 - ox) * (1 - oy), g);
    write_single_bsdf_grad(dm, tex_h, tex_w, ix, iy + 1, (1 - ox) * oy, g);
    write_single_bsdf_grad(dm, tex_h, tex_w, ix + 1, iy, ox * (1 - oy), g);
    write_single_bsdf_grad(dm, tex_h, tex_w, ix + 1, iy + 1, ox * oy, g);
}
static int v4_any_nan(v4 g) { return isnan(g.x) || isnan(g.y) || isnan(g.z) || isnan(g.w); }
static int v4_any_nonzero(v4 g) { return g.x != 0 || g.y != 0 || g.z != 0 || g.w != 0; }

/* ------------------------------------------------------------------ lights */
typedef struct { v3 wi; float dist, pdf; v3 eval; } light_sample_t; /* light.py:11 */

/* heap.texture2d_sample(23332, uv) — LuisaCompute, "default filter & address mode" (envmap.py:130),
 * unpinned: bilinear between texel centres, clamp to edge (same in zdr_amd/envmap.py and the kernels) */
static v3 env_lookup(const zdro_scene *s, v2 uv) {
    float x = uv.x * (float)s->env_w - 0.5f, y = uv.y * (float)s->env_h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    int x0 = clampi((int)x0f, 0, s->env_w - 1), x1 = clampi((int)x0f + 1, 0, s->env_w - 1);
    int y0 = clampi((int)y0f, 0, s->env_h - 1), y1 = clampi((int)y0f + 1, 0, s->env_h - 1);
    const float *c00 = s->env_tex + 4 * ((size_t)y0 * s->env_w + x0), *c10 = s->env_tex + 4 * ((size_t)y0 * s->env_w + x1);
    const float *c01 = s->env_tex + 4 * ((size_t)y1 * s->env_w + x0), *c11 = s->env_tex + 4 * ((size_t)y1 * s->env_w + x1);
    float r[3];
    for (int k = 0; k < 3; k++) {
        float top = c00[k] + (c10[k] - c00[k]) * fx, bot = c01[k] + (c11[k] - c01[k]) * fx;
        r[k] = top + (bot - top) * fy;
    }
    return V3(r[0], r[1], r[2]);
}
static v3 uv_to_direction(v2 uv) { /* envmap.py:206-214 */
    float phi = 2.0f * PI_F * (1.0f - uv.x), theta = PI_F * uv.y;
    float y = cosf(theta), st = sinf(theta);
    return vnormalize(V3(sinf(phi) * st, y, cosf(phi) * st));
}
static v2 direction_to_uv(v3 d) { /* envmap.py:216-220 */
    v2 uv; uv.x = 1.0f - atan2f(d.x, d.z) / (2.0f * PI_F); uv.y = acosf(d.y) / PI_F;
    return uv;
}
static void sample_alias_table(const zdro_scene *s, int n, float u_in, int offset, int *index, float *uu) { /* envmap.py:85-106 */
    float u = u_in * (float)n;
    int i = clampi((int)u, 0, n - 1);
    float ur = u - floorf(u);
    float prob = s->alias_prob[i + offset];
    if (ur < prob) { *index = i; *uu = ur / prob; }
    else { *index = s->alias_idx[i + offset]; *uu = (ur - prob) / (1.0f - prob); }
}
/* n = number of lights the light sampler chooses from: the reference forgets the 1/n for the environment
 * (envmap.py:236) and for sample_light_pdf (light.py:89) — SURVEY App. B-9 "fix when envmap lands" */
static light_sample_t sample_envmap(const zdro_scene *s, v2 u, int n) { /* envmap.py:223-238 */
    int iy, ix; float uy, ux;
    sample_alias_table(s, s->map_h, u.y, 0, &iy, &uy);
    sample_alias_table(s, s->map_w, u.x, s->map_h + iy * s->map_w, &ix, &ux);
    v2 uv; uv.x = ((float)ix + ux) / (float)s->map_w; uv.y = ((float)iy + uy) / (float)s->map_h;
    float pdf = s->env_pdf[iy * s->map_w + ix];
    light_sample_t L;
    L.wi = uv_to_direction(uv); L.dist = 1e30f;
    float sn = sinf(PI_F * uv.y), inv_s = sn > 0 ? 1.0f / sn : 0.0f;
    L.pdf = pdf * inv_s / (2.0f * PI_F * PI_F) / (float)n;
    L.eval = env_lookup(s, uv);
    return L;
}
static float env_sampled_light_pdf(const zdro_scene *s, v3 dir, int n) { /* envmap.py:240-248 */
    v2 uv = direction_to_uv(dir);
    int index = clampi((int)(uv.y * (float)s->map_h), 0, s->map_h - 1) * s->map_w + clampi((int)(uv.x * (float)s->map_w), 0, s->map_w - 1);
    float sn = sinf(PI_F * uv.y), inv_s = sn > 0 ? 1.0f / sn : 0.0f;
    return s->env_pdf[index] * inv_s / (2.0f * PI_F * PI_F) / (float)n;
}

static v3 sample_uniform_triangle(v2 u) { /* light.py:16-20 */
    v2 uv;
    if (u.x < u.y) { uv.x = 0.5f * u.x; uv.y = -0.5f * u.x + u.y; }
    else { uv.x = -0.5f * u.y + u.x; uv.y = 0.5f * u.y; }
    return V3(uv.x, uv.y, 1.0f - uv.x - uv.y);
}

static float light_pdf_core(const zdro_scene *s, int n, int inst, int prim, v3 origin, v3 p, float *cos_light_out, v3 *wi_out, float *sqr_out) {
    int trig_count = s->tri_begin[inst + 1] - s->tri_begin[inst];
    int t = s->tri_begin[inst] + prim;
    v3 p0 = s->wp[3 * (size_t)t], p1 = s->wp[3 * (size_t)t + 1], p2 = s->wp[3 * (size_t)t + 2];
    v3 wi = vnormalize(vsub(p, origin));
    v3 c = vcross(vsub(p1, p0), vsub(p2, p0));
    v3 ln = vnormalize(c);
    float cos_light = -vdot(ln, wi);
    v3 dp = vsub(p, origin);
    float sqr_dist = vdot(dp, dp);
    float area = sqrtf(vdot(c, c)) / 2.0f;
    if (cos_light_out) *cos_light_out = cos_light;
    if (wi_out) *wi_out = wi;
    if (sqr_out) *sqr_out = sqr_dist;
    return sqr_dist / ((float)(n * trig_count) * area * cos_light);
}

static light_sample_t sample_light(const zdro_scene *s, v3 origin, sampler_t *smp) { /* light.py:23-81, mesh lights only */
    light_sample_t L; memset(&L, 0, sizeof L);
    float u = sampler_next(smp);
    int n = s->env_count + s->light_count; /* point_light_count = 0 (light.py:7) */
    if (n <= 0) { /* reference would index out of bounds; consume the same dimensions, contribute nothing */
        (void)sampler_next(smp); (void)sampler_next2(smp);
        L.wi = V3(0, 0, 1); L.dist = 0; L.pdf = 1.0f; L.eval = V3(0, 0, 0);
        return L;
    }
    int idx = clampi((int)(u * (float)n), 0, n - 1);
    if (idx < s->env_count) return sample_envmap(s, sampler_next2(smp), n); /* light.py:29-31: only next2f() is drawn */
    idx -= s->env_count;
    int inst = s->light_insts[idx];
    int trig_count = s->tri_begin[inst + 1] - s->tri_begin[inst];
    int prim = clampi((int)(sampler_next(smp) * (float)trig_count), 0, trig_count - 1);
    int t = s->tri_begin[inst] + prim;
    v3 p0 = s->wp[3 * (size_t)t], p1 = s->wp[3 * (size_t)t + 1], p2 = s->wp[3 * (size_t)t + 2];
    v3 abc = sample_uniform_triangle(sampler_next2(smp));
    v3 p = vadd(vadd(vscale(p0, abc.x), vscale(p1, abc.y)), vscale(p2, abc.z));
    float cos_light, sqr_dist; v3 wi;
    float pdf = light_pdf_core(s, n, inst, prim, origin, p, &cos_light, &wi, &sqr_dist);
    const float *e = s->emission + 3 * inst;
    L.wi = wi; L.dist = 0.9999f * sqrtf(sqr_dist); L.pdf = pdf;
    L.eval = cos_light > 1e-4f ? V3(e[0], e[1], e[2]) : V3(0, 0, 0);
    return L;
}

static float sample_light_pdf(const zdro_scene *s, v3 origin, int inst, int prim, v3 p) { /* light.py:84-111 */
    return light_pdf_core(s, s->env_count + s->light_count, inst, prim, origin, p, 0, 0, 0);
}

static float balanced_heuristic(float a, float b) { return a / fmaxf(a + b, 1e-4f); } /* prb.py:12-13 */

/* ---------------------------------------------------------------- counters */
typedef struct { uint64_t c[8]; } counters_t;
enum { C_SAMPLES, C_CLOSEST, C_HITS, C_SHADOW, C_SHADED, C_EMIT_BSDF, C_NAN, C_SCATTER };

/* ------------------------------------------------------------- collocated */
static v3 collocated_estimator(const zdro_scene *s, const zdro_params *P, const float *mat, ray_t ray, counters_t *C) { /* collocated.py:11-31 */
    C->c[C_CLOSEST]++;
    hit_t hit = trace_closest(s, &ray);
    if (hit.inst < 0) return V3(0, 0, 0);
    C->c[C_HITS]++;
    interaction_t it = surface_interact(s, &hit);
    if (vdot(vneg(ray.d), it.ng) < 1e-4f || vdot(vneg(ray.d), it.ns) < 1e-4f) return V3(0, 0, 0);
    v4 m = read_bsdf(mat, P->tex_h, P->tex_w, it.uv);
    C->c[C_SHADED]++;
    onb_t onb = make_onb(it.ns);
    v3 wo = to_local(&onb, vneg(ray.d));
    v3 beta = ggx_brdf(wo, wo, V3(m.x, m.y, m.z), 0.04f, m.w);
    float inv = 1.0f / hit.t, li = inv * inv;
    return vscale(beta, li);
}

static void collocated_backward(const zdro_scene *s, const zdro_params *P, const float *mat, double *dmat, ray_t ray, v3 le_grad, counters_t *C) { /* collocated.py:35-57 */
    hit_t hit = trace_closest(s, &ray);
    if (hit.inst < 0) return;
    interaction_t it = surface_interact(s, &hit);
    if (vdot(vneg(ray.d), it.ng) < 1e-4f || vdot(vneg(ray.d), it.ns) < 1e-4f) return;
    v4 m = read_bsdf(mat, P->tex_h, P->tex_w, it.uv);
    onb_t onb = make_onb(it.ns);
    v3 wo = to_local(&onb, vneg(ray.d));
    float inv = 1.0f / hit.t, li = inv * inv;
    v4 g = ggx_brdf_grad(wo, wo, V3(m.x, m.y, m.z), 0.04f, m.w, vscale(le_grad, li));
    if (!v4_any_nan(g)) { write_bsdf_grad(dmat, P->tex_h, P->tex_w, it.uv, g); C->c[C_SCATTER]++; }
}

/* ------------------------------------------------------------------ uvgrad */
/* uvgrad.py:6-49 (screen -> texture Jacobian of the primary hit): returns (dudx, dvdx, dudy, dvdy).
 * Positions are taken in WORLD space (the reference reads the untransformed vertex positions,
 * uvgrad.py:29-31, which only agrees with its world-space rays for identity transforms). */
static v4 uvgrad_estimator(const zdro_scene *s, ray_t ray, ray_t rdx, ray_t rdy) {
    v4 zero = {0, 0, 0, 0};
    hit_t hit = trace_closest(s, &ray);
    if (hit.inst < 0) return zero;
    int t = s->tri_begin[hit.inst] + hit.prim;
    const float *a = s->verts + 8 * (size_t)s->tris[3 * (size_t)t];
    const float *b = s->verts + 8 * (size_t)s->tris[3 * (size_t)t + 1];
    const float *c = s->verts + 8 * (size_t)s->tris[3 * (size_t)t + 2];
    v3 p0 = s->wp[3 * (size_t)t], p1 = s->wp[3 * (size_t)t + 1], p2 = s->wp[3 * (size_t)t + 2];
    float w0 = 1.0f - hit.u - hit.v;
    v3 p = vadd(vadd(vscale(p0, w0), vscale(p1, hit.u)), vscale(p2, hit.v));
    /* compute_dpduv (uvgrad.py:6-16): dpde = [e1 e2], duvde = [pt1-pt0, pt2-pt0] (columns) */
    v3 e1 = vsub(p1, p0), e2 = vsub(p2, p0);
    float m00 = b[3] - a[3], m10 = b[4] - a[4], m01 = c[3] - a[3], m11 = c[4] - a[4]; /* m[row][col] */
    float det = m00 * m11 - m01 * m10;
    float i00 = m11 / det, i01 = -m01 / det, i10 = -m10 / det, i11 = m00 / det;       /* inverse */
    v3 dpdu = vadd(vscale(e1, i00), vscale(e2, i10));
    v3 dpdv = vneg(vadd(vscale(e1, i01), vscale(e2, i11)));                           /* inverted v */
    v3 ng = vnormalize(vcross(e1, e2));
    float t_dx = vdot(vsub(p, rdx.o), ng) / vdot(rdx.d, ng);
    float t_dy = vdot(vsub(p, rdy.o), ng) / vdot(rdy.d, ng);
    v3 dpdx = vsub(vadd(rdx.o, vscale(rdx.d, t_dx)), p);
    v3 dpdy = vsub(vadd(rdy.o, vscale(rdy.d, t_dy)), p);
    /* (A^T A)^-1 A^T with A = [dpdu dpdv] (uvgrad.py:45-48) */
    float a00 = vdot(dpdu, dpdu), a01 = vdot(dpdu, dpdv), a11 = vdot(dpdv, dpdv);
    float d2 = a00 * a11 - a01 * a01;
    float j00 = a11 / d2, j01 = -a01 / d2, j11 = a00 / d2;
    float bx0 = vdot(dpdu, dpdx), bx1 = vdot(dpdv, dpdx), by0 = vdot(dpdu, dpdy), by1 = vdot(dpdv, dpdy);
    v4 r = {j00 * bx0 + j01 * bx1, j01 * bx0 + j11 * bx1, j00 * by0 + j01 * by1, j01 * by0 + j11 * by1};
    return r;
}

/* ------------------------------------------------------------------ direct */
/* Walks direct.py:21-85; when dmat != NULL also accumulates the adjoint of
 * direct.py:89-167 (gradient written at the PRIMARY uv, App. B-11). */
static v3 direct_walk(const zdro_scene *s, const zdro_params *P, const float *mat, ray_t ray, sampler_t *smp,
                      double *dmat, v3 le_grad, counters_t *C) {
    C->c[C_CLOSEST]++;
    hit_t hit = trace_closest(s, &ray);
    if (hit.inst < 0) return s->env_count > 0 ? env_lookup(s, direction_to_uv(ray.d)) : V3(0, 0, 0); /* direct.py:23-24 */
    C->c[C_HITS]++;
    interaction_t it = surface_interact(s, &hit);
    if (vdot(vneg(ray.d), it.ng) < 1e-4f || vdot(vneg(ray.d), it.ns) < 1e-4f) return V3(0, 0, 0);
    if (hit.inst > 0) { const float *e = s->emission + 3 * hit.inst; return V3(e[0], e[1], e[2]); } /* direct.py:30-32 */
    v4 m = read_bsdf(mat, P->tex_h, P->tex_w, it.uv);
    v3 diffuse = V3(m.x, m.y, m.z); float roughness = m.w; const float specular = 0.04f;
    C->c[C_SHADED]++;
    v2 uv0 = it.uv;
    v4 mat_grad = {0, 0, 0, 0};
    v3 radiance = V3(0, 0, 0);
    light_sample_t light = sample_light(s, it.p, smp);
    ray_t sh; sh.o = it.p; sh.d = light.wi; sh.tmin = 1e-4f; sh.tmax = light.dist;
    C->c[C_SHADOW]++;
    int occluded = trace_any(s, &sh);
    onb_t onb = make_onb(it.ns);
    v3 wo = to_local(&onb, vneg(ray.d));
    v3 wil = to_local(&onb, light.wi);
    if (!occluded && wil.z > 0.0f) { /* direct.py:49 */
        v3 bsdf = ggx_brdf(wo, wil, diffuse, specular, roughness);
        float pdf_bsdf = ggx_sample_pdf(wo, wil, roughness);
        float mis = balanced_heuristic(light.pdf, pdf_bsdf);
        float dn = fmaxf(light.pdf, 1e-4f);
        v3 W = vdivs(vscale(light.eval, mis), dn);
        radiance = vadd(radiance, vdivs(vmul(vscale(bsdf, mis), light.eval), dn));
        if (dmat) { v4 g = ggx_brdf_grad(wo, wil, diffuse, specular, roughness, vmul(W, le_grad));
            mat_grad.x += g.x; mat_grad.y += g.y; mat_grad.z += g.z; mat_grad.w += g.w; }
    }
    /* use_MIS = True (direct.py:14) */
    v3 wi_local = ggx_sample(wo, roughness, smp);
    v3 wi = to_world(&onb, wi_local);
    do {
        if (vdot(wi, it.ng) < 1e-4f || wi_local.z < 1e-4f) break;
        ray_t r2; r2.o = offset_ray_origin(it.p, it.ng); r2.d = wi; r2.tmin = 0.0f; r2.tmax = 1e30f;
        v3 origin = it.p;
        C->c[C_CLOSEST]++;
        hit_t h2 = trace_closest(s, &r2);
        float envc[3] = {0, 0, 0}; const float *e; float pdf_light;
        if (h2.inst < 0) { /* direct.py:68-71 */
            if (s->env_count <= 0) break;
            v3 ev = env_lookup(s, direction_to_uv(r2.d));
            envc[0] = ev.x; envc[1] = ev.y; envc[2] = ev.z; e = envc;
            pdf_light = env_sampled_light_pdf(s, r2.d, s->env_count + s->light_count);
        } else {
            C->c[C_HITS]++;
            interaction_t it2 = surface_interact(s, &h2);
            if (vdot(vneg(r2.d), it2.ng) < 1e-4f || vdot(vneg(r2.d), it2.ns) < 1e-4f) break;
            e = s->emission + 3 * h2.inst;
            pdf_light = sample_light_pdf(s, origin, h2.inst, h2.prim, it2.p);
        }
        if (e[0] > 0 || e[1] > 0 || e[2] > 0) {
            float pdf_bsdf = ggx_sample_pdf(wo, wi_local, roughness);
            float mis = balanced_heuristic(pdf_bsdf, pdf_light);
            v3 beta = vdivs(ggx_brdf(wo, wi_local, diffuse, specular, roughness), pdf_bsdf);
            v3 em = V3(e[0], e[1], e[2]);
            C->c[C_EMIT_BSDF]++;
            radiance = vadd(radiance, vmul(vscale(beta, mis), em));
            if (dmat) { v3 ct = vmul(vscale(em, mis / pdf_bsdf), le_grad);
                v4 g = ggx_brdf_grad(wo, wi_local, diffuse, specular, roughness, ct);
                mat_grad.x += g.x; mat_grad.y += g.y; mat_grad.z += g.z; mat_grad.w += g.w; }
        }
    } while (0);
    if (dmat && v4_any_nonzero(mat_grad) && !v4_any_nan(mat_grad)) {
        write_bsdf_grad(dmat, P->tex_h, P->tex_w, uv0, mat_grad); C->c[C_SCATTER]++;
    }
    return radiance;
}

/* -------------------------------------------------------------------- path */
static int g_debug_rr_clamp = 0;
void zdro_debug_rr_clamp(int on) { g_debug_rr_clamp = on; }
#define ZDRO_MAX_DEPTH 64
typedef struct {
    v2 uv; v4 mat; v3 wo;
    int has_nee; v3 wi_light, W;   /* W = mis * eval / max(pdf_light, 1e-4) */
    int has_bsdf; v3 wi; float pdf, q; /* bounce continued: f/pdf/q multiplies beta */
    v3 beta;                       /* throughput on entry */
    int rr_scaled, rr_norm;        /* RR divided beta by q = lum(beta') (>= 0.05); ... and lum(beta') >= 1 */
    float nee_pb_frac;             /* pdf_bsdf / (pdf_light + pdf_bsdf) of the accepted light sample */
    v3 beta_out;                   /* throughput leaving the vertex (unit luminance when rr_scaled) */
    int inst, prim; v3 wi_world, L_nee;   /* zdro_path_dump only: the hit, the sampled direction in world space, the NEE radiance added here */
} path_vertex_t;

/* prb.py:19-88 with the current helper signatures (App. B-1). Optionally records
 * the shaded vertices (for the adjoint sweep) and the terminal emitter term. */
static v3 path_walk(const zdro_scene *s, const zdro_params *P, const float *mat, ray_t ray, sampler_t *smp,
                    path_vertex_t *rec, int *nrec, v3 *terminal_Li, counters_t *C, float *terminal_pl_frac) {
    if (terminal_pl_frac) *terminal_pl_frac = 0.0f;
    v3 radiance = V3(0, 0, 0), beta = V3(1, 1, 1);
    float pdf_bsdf = 1e30f;
    int nr = 0;
    if (terminal_Li) *terminal_Li = V3(0, 0, 0);
    int max_depth = P->max_depth < ZDRO_MAX_DEPTH ? P->max_depth : ZDRO_MAX_DEPTH;
    for (int depth = 0; depth < max_depth; depth++) {
        C->c[C_CLOSEST]++;
        hit_t hit = trace_closest(s, &ray);
        if (hit.inst < 0) { /* prb.py:26-32; written like direct.py:70-83 (prb.py is stale and squares beta) */
            if (s->env_count > 0) {
                v3 em = env_lookup(s, direction_to_uv(ray.d));
                float pdf_light = env_sampled_light_pdf(s, ray.d, s->env_count + s->light_count);
                float mis = balanced_heuristic(pdf_bsdf, pdf_light);
                radiance = vadd(radiance, vmul(vscale(beta, mis), em));
                if (terminal_Li) *terminal_Li = vscale(em, mis);
                if (terminal_pl_frac) *terminal_pl_frac = (pdf_bsdf + pdf_light > 1e-4f) ? pdf_light / (pdf_bsdf + pdf_light) : 0.0f;
            }
            break;
        }
        C->c[C_HITS]++;
        interaction_t it = surface_interact(s, &hit);
        if (vdot(vneg(ray.d), it.ng) < 1e-4f || vdot(vneg(ray.d), it.ns) < 1e-4f) break;
        const float *e = s->emission + 3 * hit.inst;
        if (e[0] > 0 || e[1] > 0 || e[2] > 0) { /* prb.py:39-44 */
            float pdf_light = sample_light_pdf(s, ray.o, hit.inst, hit.prim, it.p);
            float mis = balanced_heuristic(pdf_bsdf, pdf_light);
            v3 em = V3(e[0], e[1], e[2]);
            radiance = vadd(radiance, vmul(vscale(beta, mis), em));
            if (terminal_Li) *terminal_Li = vscale(em, mis);
            if (terminal_pl_frac) *terminal_pl_frac = (pdf_bsdf + pdf_light > 1e-4f) ? pdf_light / (pdf_bsdf + pdf_light) : 0.0f;
            if (depth > 0) C->c[C_EMIT_BSDF]++;
            break;
        }
        if (hit.inst > 0) break; /* prb.py:45-46 */
        v4 m = read_bsdf(mat, P->tex_h, P->tex_w, it.uv);
        v3 diffuse = V3(m.x, m.y, m.z); float roughness = m.w; const float specular = 0.04f;
        C->c[C_SHADED]++;
        path_vertex_t *pv = rec ? &rec[nr] : 0;
        if (pv) { memset(pv, 0, sizeof *pv); pv->uv = it.uv; pv->mat = m; pv->beta = beta; pv->q = 1.0f; pv->inst = hit.inst; pv->prim = hit.prim; }
        nr++;
        onb_t onb = make_onb(it.ns);
        v3 wo = to_local(&onb, vneg(ray.d));
        if (pv) pv->wo = wo;
        light_sample_t light = sample_light(s, it.p, smp);
        ray_t sh; sh.o = it.p; sh.d = light.wi; sh.tmin = 1e-4f; sh.tmax = light.dist;
        C->c[C_SHADOW]++;
        int occluded = trace_any(s, &sh);
        v3 wil = to_local(&onb, light.wi);
        if (!occluded && wil.z >= 1e-4f) { /* prb.py:62-66 */
            v3 bsdf = ggx_brdf(wo, wil, diffuse, specular, roughness);
            float pb = ggx_sample_pdf(wo, wil, roughness);
            float mis = balanced_heuristic(light.pdf, pb);
            float dn = fmaxf(light.pdf, 1e-4f);
            radiance = vadd(radiance, vdivs(vmul(vscale(vmul(beta, bsdf), mis), light.eval), dn));
            if (pv) { pv->L_nee = vdivs(vmul(vscale(vmul(beta, bsdf), mis), light.eval), dn);
                      pv->has_nee = 1; pv->wi_light = wil; pv->W = vdivs(vscale(light.eval, mis), dn);
                      pv->nee_pb_frac = (light.pdf + pb > 1e-4f) ? pb / (light.pdf + pb) : 0.0f; }
        }
        v3 wi_local = ggx_sample(wo, roughness, smp);
        pdf_bsdf = ggx_sample_pdf(wo, wi_local, roughness);
        v3 wi = to_world(&onb, wi_local);
        if (vdot(wi, it.ng) < 1e-4f || wi_local.z < 1e-4f) break; /* prb.py:73-74 */
        ray.o = offset_ray_origin(it.p, it.ng); ray.d = wi; ray.tmin = 0.0f; ray.tmax = 1e30f;
        beta = vmul(beta, vdivs(ggx_brdf(wo, wi_local, diffuse, specular, roughness), pdf_bsdf));
        float q = 1.0f;
        if (depth >= P->rr_depth) { /* prb.py:79-87 */
            float l = 0.212671f * beta.x + 0.715160f * beta.y + 0.072169f * beta.z;
            if (l == 0.0f) break;
            q = fmaxf(l, 0.05f);
            if (g_debug_rr_clamp) q = fminf(q, 1.0f);   /* diagnostic only (env ZDRO_DEBUG_RR_CLAMP): unbiased RR */
            float r = sampler_next(smp);
            if (r >= q) break;
            beta = vdivs(beta, q);
            if (pv && l >= 0.05f && !(g_debug_rr_clamp && l >= 1.0f)) { pv->rr_scaled = 1; pv->rr_norm = l >= 1.0f; }
        }
        if (pv) { pv->has_bsdf = 1; pv->wi = wi_local; pv->wi_world = wi; pv->pdf = pdf_bsdf; pv->q = q; pv->beta_out = beta; }
    }
    if (nrec) *nrec = nr;
    return radiance;
}

/* PRB adjoint (prb.py:92-187). One forward walk records the shaded vertices; the
 * sweep runs last-to-first carrying Li. ZDRO_PRB_LITERAL
 * reproduces the weight of prb.py:162 (beta/pdf * Le_remaining) for comparison. */
typedef struct { path_vertex_t rec[ZDRO_MAX_DEPTH]; v4 grad[ZDRO_MAX_DEPTH]; int n; v3 L, term_Li; } path_trace_t;   /* zdro_path_dump */
static void path_backward(const zdro_scene *s, const zdro_params *P, const float *mat, double *dmat, ray_t ray,
                          sampler_t *smp, v3 le_grad, counters_t *C, path_trace_t *trace) {
    path_vertex_t rec_local[ZDRO_MAX_DEPTH];
    path_vertex_t *rec = trace ? trace->rec : rec_local;
    int n = 0; v3 Li; float term_pl_frac = 0.0f;
    v3 Le = path_walk(s, P, mat, ray, smp, rec, &n, &Li, C, &term_pl_frac);
    if (trace) { trace->n = n; trace->L = Le; trace->term_Li = Li; memset(trace->grad, 0, sizeof trace->grad); }
    if (vany_nan(Le)) return; /* prb.py:100 */
    const float specular = 0.04f;
    /* Adjoint sweep, last vertex to first, extended to the derivative of the forward's
     * EXPECTATION under the reference's Russian roulette.  prb.py:83 does not clamp q = max(lum(beta'), 0.05)
     * to 1, so there are three kinds of vertex (beta' = beta f / pdf, l = lum(beta')):
     *   P  no RR (depth < rr_depth) or l < 0.05 (q constant): plain product;
     *   T  0.05 <= l < 1: survive with probability l, divide by l — unbiased, but the realised throughput
     *      leaves with unit luminance and the survival probability S picks up the factor l;
     *   N  l >= 1: survive with certainty and STILL divide by l: the expected throughput becomes
     *      S * beta'/lum(beta') — renormalised, biased, and a function of the material through the colour
     *      ratio, through S (earlier T vertices) and through the sampling density at this vertex (the
     *      factor 1/pdf no longer survives).
     * With E = expected throughput and S = survival probability: P: E' = E u;  T: E' = E u, S' = lum(E u);
     * N: E' = S E u / lum(E u).  Reverse mode over that recursion, evaluated on the realised path
     * (A = dL/dE in units of the realised beta = arriving radiance, s = dL/dS):
     *   P: Aeff = A;                 s stays
     *   T: Aeff = A + s w;           s = 0                      (w = luminance weights, prb.py:80)
     *   N: Aeff = A - w (b . A);     s += b . A;   Z = <g, b L>  (L = plain arriving radiance; b = beta leaving)
     *   grad += d f[(beta/(pdf q)) Aeff];   A = g fL W + (f/(pdf q)) Aeff
     *   grad_r += dln(pdf)/dr Z  at EVERY vertex: renormalisation discards all scalar factors gathered since
     *   the last T vertex (whose S carries them on), so the sampling densities of all those vertices lose
     *   their 1/pdf compensation: Z is set at N, cleared at T and inherited through P vertices.
     * The renormalisation mixes the colour channels, so A is the adjoint of the SCALAR <g, L> (g = the
     * pixel's cotangent), i.e. it is carried already multiplied by g.
     * ZDRO_PRB_DETACHED keeps every RR factor constant (what the reference's autodiff blocks do). */
    const v3 wl = V3(0.212671f, 0.715160f, 0.072169f);
    float sS = 0.0f, Zs = 0.0f;
    v3 Lg = vmul(Li, le_grad);       /* g-contracted adjoint; Li stays the plain radiance for ZDRO_PRB_LITERAL */
    for (int k = n - 1; k >= 0; k--) {
        path_vertex_t *v = &rec[k];
        v3 diffuse = V3(v->mat.x, v->mat.y, v->mat.z); float r = v->mat.w;
        v4 grad = {0, 0, 0, 0};
        v3 fL = V3(0, 0, 0);
        if (v->has_nee) { /* prb.py:138-146 */
            fL = ggx_brdf(v->wo, v->wi_light, diffuse, specular, r);
            v4 g = ggx_brdf_grad(v->wo, v->wi_light, diffuse, specular, r, vmul(vmul(v->beta, v->W), le_grad));
            grad.x += g.x; grad.y += g.y; grad.z += g.z; grad.w += g.w;
            if (P->prb_mode == ZDRO_PRB_CORRECT)   /* d w_nee / dr = -w_nee pb/(pl+pb) dln(pb)/dr */
                grad.w -= vdot(vmul(vmul(v->beta, vmul(fL, v->W)), le_grad), V3(1, 1, 1)) * v->nee_pb_frac * ggx_dlnpdf_dr(v->wo, v->wi_light, r);
        }
        if (P->prb_mode == ZDRO_PRB_CORRECT && k == n - 1 && v->has_bsdf)   /* d w_bsdf / dr of the emitter hit that ended the path */
            grad.w += vdot(v->beta_out, Lg) * term_pl_frac * ggx_dlnpdf_dr(v->wo, v->wi, r);
        v3 T = V3(0, 0, 0), Y = Lg;
        float score = 0.0f;
        if (!v->has_bsdf) { sS = 0.0f; Zs = 0.0f; }   /* the path stops here: nothing downstream */
        if (v->has_bsdf) { /* prb.py:157-163, corrected */
            v3 f = ggx_brdf(v->wo, v->wi, diffuse, specular, r);
            T = vdivs(vdivs(f, v->pdf), v->q);
            v3 ct;
            if (P->prb_mode == ZDRO_PRB_LITERAL) {
                v3 Le_rem = vmul(vmul(v->beta, T), Li);     /* what prb.py keeps in Le after the subtractions */
                ct = vmul(vmul(vdivs(v->beta, v->pdf), Le_rem), le_grad);
            } else {
                v3 Aeff = Lg;
                if (P->prb_mode == ZDRO_PRB_CORRECT && v->rr_scaled) {
                    if (v->rr_norm) {                                   /* N */
                        float ba = vdot(v->beta_out, Lg);
                        Aeff = vsub(Lg, vscale(wl, ba));
                        sS += ba;
                        /* the score multiplies the downstream VALUE <g, b L>, not b . A: later renormalisations make
                         * the downstream homogeneous of degree 0 in beta, so b . (gradient) would drop them (Euler) */
                        Zs = vdot(vmul(le_grad, v->beta_out), Li);
                    } else {                                            /* T */
                        Aeff = vadd(Lg, vscale(wl, sS));
                        sS = 0.0f;
                        Zs = 0.0f;
                    }
                }
                if (P->prb_mode == ZDRO_PRB_CORRECT && Zs != 0.0f) score = ggx_dlnpdf_dr(v->wo, v->wi, r) * Zs;
                Y = Aeff;
                ct = vmul(vdivs(vdivs(v->beta, v->pdf), v->q), Aeff);
            }
            v4 g = ggx_brdf_grad(v->wo, v->wi, diffuse, specular, r, ct);
            grad.x += g.x; grad.y += g.y; grad.z += g.z; grad.w += g.w + score;
        }
        Lg = vadd(vmul(vmul(fL, v->W), le_grad), vmul(T, Y));
        Li = vadd(vmul(fL, v->W), vmul(T, Li));
        if (trace) trace->grad[k] = grad;
        if (dmat && v4_any_nonzero(grad) && !v4_any_nan(grad)) { /* prb.py:178-187 */
            write_bsdf_grad(dmat, P->tex_h, P->tex_w, v->uv, grad); C->c[C_SCATTER]++;
        }
    }
}

/* ----------------------------------------------------------------- drivers */
static ray_t pixel_ray(const zdro_params *P, int x, int y, sampler_t *smp) { /* integrator.py:19-24 */
    v2 off = sampler_next2(smp);
    if (P->use_tent) { off.x = tent_warp1(off.x, 1.0f) + 0.5f; off.y = tent_warp1(off.y, 1.0f) + 0.5f; }
    float px = 2.0f / (float)P->width * ((float)x + off.x) - 1.0f;
    float py = 2.0f / (float)P->height * ((float)y + off.y) - 1.0f;
    py *= (float)P->height / (float)P->width; /* integrator.py:23; exact 1 for square images */
    return generate_ray(P, px, py);
}

static ray_t pixel_ray_at(const zdro_params *P, float fx, float fy) {
    float px = 2.0f / (float)P->width * fx - 1.0f;
    float py = 2.0f / (float)P->height * fy - 1.0f;
    py *= (float)P->height / (float)P->width;
    return generate_ray(P, px, py);
}

static int check_params(const zdro_params *P) {
    if (P->width <= 0 || P->height <= 0 || P->spp == 0) return -1;
    if (P->x0 < 0 || P->y0 < 0 || P->x1 > P->width || P->y1 > P->height) return -1;
    if (P->sample_end > P->spp || P->sample_begin > P->sample_end) return -1;
    if (P->sampler == ZDRO_SAMPLER_PMJ02BN && (!g_pmj || !g_bn)) return -2;
    if (P->integrator < 0 || P->integrator > 3) return -3;
    return 0;
}

int zdro_render_forward(const zdro_scene *s, const zdro_params *P, const float *material, float *image, uint64_t *counters) {
    int rc = check_params(P); if (rc) return rc;
    counters_t total; memset(&total, 0, sizeof total);
    int nth = P->nthreads;
#ifdef _OPENMP
    if (nth <= 0) nth = omp_get_max_threads();
#else
    nth = 1;
#endif
#pragma omp parallel num_threads(nth)
    {
        counters_t C; memset(&C, 0, sizeof C);
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int y = P->y0; y < P->y1; y++)
            for (int x = P->x0; x < P->x1; x++) { /* integrator.py:10-29 */
                v3 sum = V3(0, 0, 0);
                if (P->integrator == ZDRO_UVGRAD) { /* uvgrad.py:76-98 (scene sampler instead of LC's RNG) */
                    float s4[4] = {0, 0, 0, 0};
                    for (uint32_t it = P->sample_begin; it < P->sample_end; it++) {
                        sampler_t smp = make_sampler(P->sampler, x, y, P->seed, P->spp, it);
                        v2 off = sampler_next2(&smp);
                        if (P->use_tent) { off.x = tent_warp1(off.x, 1.0f) + 0.5f; off.y = tent_warp1(off.y, 1.0f) + 0.5f; }
                        float fx = (float)x + off.x, fy = (float)y + off.y;
                        v4 g = uvgrad_estimator(s, pixel_ray_at(P, fx, fy), pixel_ray_at(P, fx + 1.0f, fy), pixel_ray_at(P, fx, fy + 1.0f));
                        if (!v4_any_nan(g)) { s4[0] += g.x; s4[1] += g.y; s4[2] += g.z; s4[3] += g.w; }
                    }
                    float *px4 = image + 4 * ((size_t)x + (size_t)y * P->width);
                    for (int k = 0; k < 4; k++) px4[k] = s4[k] / (float)P->spp;
                    continue;
                }
                for (uint32_t it = P->sample_begin; it < P->sample_end; it++) {
                    sampler_t smp = make_sampler(P->sampler, x, y, P->seed, P->spp, it);
                    ray_t ray = pixel_ray(P, x, y, &smp);
                    v3 rad;
                    C.c[C_SAMPLES]++;
                    if (P->integrator == ZDRO_COLLOCATED) rad = collocated_estimator(s, P, material, ray, &C);
                    else if (P->integrator == ZDRO_DIRECT) rad = direct_walk(s, P, material, ray, &smp, 0, V3(0, 0, 0), &C);
                    else rad = path_walk(s, P, material, ray, &smp, 0, 0, 0, &C, 0);
                    if (!vany_nan(rad)) { /* integrator.py:27-28 */
                        sum.x += clampf(rad.x, 0.0f, 100000.0f); sum.y += clampf(rad.y, 0.0f, 100000.0f); sum.z += clampf(rad.z, 0.0f, 100000.0f);
                    } else C.c[C_NAN]++;
                }
                float *px = image + 4 * ((size_t)x + (size_t)y * P->width);
                px[0] = sum.x / (float)P->spp; px[1] = sum.y / (float)P->spp; px[2] = sum.z / (float)P->spp;
                px[3] = (float)(P->sample_end - P->sample_begin) / (float)P->spp; /* 1.0 for a full render */
            }
#pragma omp critical
        for (int i = 0; i < 8; i++) total.c[i] += C.c[i];
    }
    if (counters) memcpy(counters, total.c, sizeof total.c);
    return 0;
}

int zdro_render_backward(const zdro_scene *s, const zdro_params *P, const float *d_image, const float *material,
                         float *d_material, uint64_t *counters) {
    int rc = check_params(P); if (rc) return rc;
    counters_t total; memset(&total, 0, sizeof total);
    size_t ntex = (size_t)P->tex_h * P->tex_w * 4;
    int nth = P->nthreads;
#ifdef _OPENMP
    if (nth <= 0) nth = omp_get_max_threads();
#else
    nth = 1;
#endif
    /* one shared float64 accumulator, updated with atomic adds: the stand-in for the reference's
     * atomic_fetch_add (interaction.py:67-70).  float64 sums make the arrival order irrelevant at
     * float32 output precision. */
    double *dm = (double *)calloc(ntex, sizeof(double));
    g_atomic_scatter = nth > 1;
#pragma omp parallel num_threads(nth)
    {
        counters_t C; memset(&C, 0, sizeof C);
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int y = P->y0; y < P->y1; y++)
            for (int x = P->x0; x < P->x1; x++) { /* integrator.py:34-52 */
                const float *g = d_image + 4 * ((size_t)x + (size_t)y * P->width);
                v3 le_grad = V3(g[0] / (float)P->spp, g[1] / (float)P->spp, g[2] / (float)P->spp);
                if (vany_nan(le_grad)) le_grad = V3(0, 0, 0);
                for (uint32_t it = P->sample_begin; it < P->sample_end; it++) {
                    sampler_t smp = make_sampler(P->sampler, x, y, P->seed, P->spp, it);
                    ray_t ray = pixel_ray(P, x, y, &smp);
                    C.c[C_SAMPLES]++;
                    if (P->integrator == ZDRO_COLLOCATED) collocated_backward(s, P, material, dm, ray, le_grad, &C);
                    else if (P->integrator == ZDRO_DIRECT) (void)direct_walk(s, P, material, ray, &smp, dm, le_grad, &C);
                    else path_backward(s, P, material, dm, ray, &smp, le_grad, &C, 0);
                }
            }
#pragma omp critical
        for (int i = 0; i < 8; i++) total.c[i] += C.c[i];
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < ntex; i++) d_material[i] = (float)((double)d_material[i] + dm[i]);
    free(dm);
    if (counters) memcpy(counters, total.c, sizeof total.c);
    return 0;
}

/* Per-path trace of the path integrator (test hook; twin of zdr_path_dump in include/zdr.h, same layout):
 * for each query {px, py, sample_index} the path of that camera sample as path_backward walks it.
 * out: n x (8 + 24 maxv) floats
 *   header  {bits(nvert), L.rgb (unclamped radiance of the sample), 0, term_Li.rgb}
 *   vertex  {bits(inst), bits(prim), uv.xy, bits(flags), pdf_bsdf, wi.xyz (world), beta_out.rgb, grad.rgba,
 *            L_nee.rgb, 0 x 5};  flags = has_nee | has_bsdf << 1 | rr_kind << 2 (rr_kind 0 none, 1 stochastic, 2 renormalising)
 * d_image NULL = a cotangent of ones.  The gradient is what the backward pass scatters for this vertex. */
int zdro_path_dump(const zdro_scene *s, const zdro_params *P, const float *material, const float *d_image,
                   const int32_t *queries, int n, int maxv, float *out) {
    int rc = check_params(P); if (rc) return rc;
    if (maxv < 1 || maxv > ZDRO_MAX_DEPTH) return -4;
    const int stride = 8 + 24 * maxv;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        const int x = queries[3 * i], y = queries[3 * i + 1]; const uint32_t it = (uint32_t)queries[3 * i + 2];
        float *o = out + (size_t)i * stride;
        memset(o, 0, sizeof(float) * stride);
        if (x < 0 || y < 0 || x >= P->width || y >= P->height || it >= P->spp) continue;   /* as the GPU twin: such a query reads as all zeros */
        v3 le_grad = V3(1.0f / (float)P->spp, 1.0f / (float)P->spp, 1.0f / (float)P->spp);
        if (d_image) {
            const float *g = d_image + 4 * ((size_t)x + (size_t)y * P->width);
            le_grad = V3(g[0] / (float)P->spp, g[1] / (float)P->spp, g[2] / (float)P->spp);
            if (vany_nan(le_grad)) le_grad = V3(0, 0, 0);
        }
        sampler_t smp = make_sampler(P->sampler, x, y, P->seed, P->spp, it);
        ray_t ray = pixel_ray(P, x, y, &smp);
        counters_t C; memset(&C, 0, sizeof C);
        path_trace_t T; memset(&T, 0, sizeof T);
        path_backward(s, P, material, 0, ray, &smp, le_grad, &C, &T);
        int32_t nv = T.n; memcpy(&o[0], &nv, 4);
        o[1] = T.L.x; o[2] = T.L.y; o[3] = T.L.z; o[5] = T.term_Li.x; o[6] = T.term_Li.y; o[7] = T.term_Li.z;
        for (int k = 0; k < T.n && k < maxv; k++) {
            const path_vertex_t *v = &T.rec[k];
            float *q = o + 8 + 24 * k;
            int32_t fl = (v->has_nee ? 1 : 0) | (v->has_bsdf ? 2 : 0) | ((v->rr_scaled ? (v->rr_norm ? 2 : 1) : 0) << 2);
            memcpy(&q[0], &v->inst, 4); memcpy(&q[1], &v->prim, 4); q[2] = v->uv.x; q[3] = v->uv.y; memcpy(&q[4], &fl, 4);
            if (v->has_bsdf) { q[5] = v->pdf; q[6] = v->wi_world.x; q[7] = v->wi_world.y; q[8] = v->wi_world.z;
                               q[9] = v->beta_out.x; q[10] = v->beta_out.y; q[11] = v->beta_out.z; }
            q[12] = T.grad[k].x; q[13] = T.grad[k].y; q[14] = T.grad[k].z; q[15] = T.grad[k].w;
            q[16] = v->L_nee.x; q[17] = v->L_nee.y; q[18] = v->L_nee.z;
        }
    }
    return 0;
}

float zdro_ggx_dlnpdf_dr(const float wo[3], const float wi[3], float r) {
    return ggx_dlnpdf_dr(V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), r);
}
