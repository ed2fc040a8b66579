// integrators.h — the three radiance estimators of the reference and their adjoints, written
// for wave64: one lane owns one pixel and walks its samples in index order (the summation order
// of integrator.py:15-29), but the path integrator runs as a flat state machine — each loop
// trip every live lane shades ONE vertex of its path and finished lanes immediately start
// their next sample, so short paths do not idle behind the longest path of the wave.
#pragma once
#include "accel.h"
#include "internal.h"
#include "microfacet.h"
#include "zdr.h"

ZD float tent_warp1(float u) {                                   // camera.py:20-31, radius 1
    return (u < 0.5f) ? (fsqrt(2.0f * u) - 1.0f) : (1.0f - fsqrt(2.0f - 2.0f * u));
}

// integrator.py:19-24 + camera.py:5-17
// PACKED: the pixel's 2-D draw in two hash passes instead of three (sampler.h, cmj_next2_packed; same values).  The direct / collocated kernels take it
// (direct forward 0.898 -> 0.873 ms); the path kernels do NOT: there it buys nothing (the camera samples are 7 % of their instructions) and, in the BVH forward
// kernel, which lives at 80 VGPRs, it moved five spill operations INTO the walk loop: 184 -> 199 ms on the 1 M-triangle scene (round 4, measured).
template <int SK, bool PACKED = false>
ZD void pixel_ray(const RenderCfg &R, const SamplerCfg &C, Sampler &smp, int x, int y, f3 &o, f3 &d) {
    f2 off = PACKED ? sampler_pixel_offset<SK>(C, smp) : sampler_next2<SK>(C, smp);
    if (R.use_tent) { off.x = tent_warp1(off.x) + 0.5f; off.y = tent_warp1(off.y) + 0.5f; }
    float px = R.two_over_w * ((float)x + off.x) - 1.0f;
    float py = R.two_over_h * ((float)y + off.y) - 1.0f;
    py *= R.aspect;
    px *= R.cam_tan; py *= R.cam_tan;
    o = ld3(R.cam_o);
    d = normalize((ld3(R.cam_right) * px - ld3(R.cam_upp) * py) + ld3(R.cam_fwd));
}

struct Counters { uint32_t c[8]; };
enum { C_SAMPLES, C_CLOSEST, C_HITS, C_SHADOW, C_SHADED, C_EMIT_BSDF, C_NAN, C_SHADOW_TRACED };
#define COUNT(i) do { if (STATS) cnt.c[i]++; } while (0)

ZD f3 clamp_radiance(f3 r) { return mk3(clampf(r.x, 0.0f, 100000.0f), clampf(r.y, 0.0f, 100000.0f), clampf(r.z, 0.0f, 100000.0f)); }
ZD bool any_nonzero4(float4 g) { return (g.x != 0.0f) | (g.y != 0.0f) | (g.z != 0.0f) | (g.w != 0.0f); }
ZD bool any_nan4(float4 g) { return (g.x != g.x) | (g.y != g.y) | (g.z != g.z) | (g.w != g.w); }
ZD float4 brdf_grad(float cz_over_pi, float dfdr, f3 ct) {     // d(f cos)[ct] w.r.t. (d.rgb, r), App. A.6
    return make_float4(ct.x * cz_over_pi, ct.y * cz_over_pi, ct.z * cz_over_pi, (ct.x + ct.y + ct.z) * dfdr);
}

// ---------------------------------------------------------------------------- collocated
// collocated.py:11-31 / 35-57: L = brdf(wo, wo) / t^2
// BWD: the vertex gradient is returned through (guv, grad) — grad stays 0 when there is nothing to add —
// and the caller queues it at a reconverged point (scene.h, ScatterQueue).
template <class A, bool BWD, bool STATS>
ZD f3 collocated_sample(const DScene &S, const RenderCfg &R, const KernelIO &io, int *lds, f3 o, f3 d, unsigned long long cam_mask, f3 le_grad, Counters &cnt, f2 &guv, float4 &grad) {
    COUNT(C_CLOSEST);
    Hit h = A::closest_camera(S, lds, o, d, cam_mask);
    if (h.slot < 0) return mk3(0.0f);
    COUNT(C_HITS);
    Interaction it = surface_interact(S, h);
    if (dot(-d, it.ng) < 1e-4f || dot(-d, it.ns) < 1e-4f) return mk3(0.0f);
    float4 m = read_bsdf(io.material, it.uv, R.tex_h, R.tex_w);
    COUNT(C_SHADED);
    Onb onb = make_onb(it.ns);
    f3 wo = to_local(onb, -d);
    GgxTerms g = ggx_terms(wo, wo, m.w);
    float inv_t = rcp(h.t), li = inv_t * inv_t;
    if (BWD) {
        float dl_; grad = brdf_grad(wo.z * ZDR_INV_PI, ggx_dfdr_from(g, wo, wo, m.w, dl_), le_grad * li);
        guv = it.uv;
    }
    return ggx_brdf_from(g, wo, mk3(m.x, m.y, m.z)) * li;
}

// -------------------------------------------------------------------------------- uvgrad
// uvgrad.py:6-49: Jacobian of the texture coordinates of the primary hit w.r.t. the pixel position,
// (dudx, dvdx, dudy, dvdy), from the hits of the rays through (x+1, y) and (x, y+1) with the hit
// triangle's plane.  World-space positions (see oracle/zdr_oracle.c uvgrad_estimator).
template <class A>
ZD float4 uvgrad_sample(const DScene &S, int *lds, f3 o, f3 d, f3 odx, f3 ddx, f3 ody, f3 ddy) {
    Hit h = A::closest(S, lds, o, d, 0.0f, 1e30f);
    if (h.slot < 0) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 *r = S.shade + 8 * (size_t)h.slot;
    float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6];
    f3 p0 = xyz(r0), p1 = xyz(r1), p2 = xyz(r2);
    float w0 = 1.0f - h.u - h.v;
    f3 p = p0 * w0 + p1 * h.u + p2 * h.v;
    f3 e1 = p1 - p0, e2 = p2 - p0;
    float m00 = r2.w - r0.w, m10 = r3.w - r1.w, m01 = r4.w - r0.w, m11 = r5.w - r1.w;   // [pt1-pt0, pt2-pt0] as columns
    float idet = rcp(m00 * m11 - m01 * m10);
    float i00 = m11 * idet, i01 = -m01 * idet, i10 = -m10 * idet, i11 = m00 * idet;
    f3 dpdu = e1 * i00 + e2 * i10;
    f3 dpdv = -(e1 * i01 + e2 * i11);                                                  // inverted v (uvgrad.py:15)
    f3 ng = xyz(r6);
    float t_dx = dot(p - odx, ng) * rcp(dot(ddx, ng));
    float t_dy = dot(p - ody, ng) * rcp(dot(ddy, ng));
    f3 dpdx = (odx + ddx * t_dx) - p, dpdy = (ody + ddy * t_dy) - p;
    float a00 = dot(dpdu, dpdu), a01 = dot(dpdu, dpdv), a11 = dot(dpdv, dpdv);
    float id2 = rcp(a00 * a11 - a01 * a01);
    float j00 = a11 * id2, j01 = -a01 * id2, j11 = a00 * id2;
    float bx0 = dot(dpdu, dpdx), bx1 = dot(dpdv, dpdx), by0 = dot(dpdu, dpdy), by1 = dot(dpdv, dpdy);
    return make_float4(j00 * bx0 + j01 * bx1, j01 * bx0 + j11 * bx1, j00 * by0 + j01 * by1, j01 * by0 + j11 * by1);
}

// -------------------------------------------------------------------------------- direct
// direct.py:21-85 (forward) / 89-167 (adjoint; gradient written once at the primary uv, App. B-11)
template <int SK, class A, bool BWD, bool STATS, bool ENV>
ZD f3 direct_sample(const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int *lds,
                    Sampler &smp, f3 o, f3 d, unsigned long long cam_mask, f3 le_grad, Counters &cnt, f2 &guv, float4 &grad) {
    COUNT(C_CLOSEST);
    Hit h = A::closest_camera(S, lds, o, d, cam_mask);
    if (h.slot < 0) return (ENV && S.env_count > 0) ? env_lookup(S, direction_to_uv(d)) : mk3(0.0f);   // direct.py:23-24
    COUNT(C_HITS);
    Interaction it = surface_interact(S, h);
    if (dot(-d, it.ng) < 1e-4f || dot(-d, it.ns) < 1e-4f) return mk3(0.0f);
    if (it.inst > 0) return xyz(S.emission4[it.inst]);                            // direct.py:30-32
    float4 m = read_bsdf(io.material, it.uv, R.tex_h, R.tex_w);
    f3 diffuse = mk3(m.x, m.y, m.z); float roughness = m.w;
    COUNT(C_SHADED);
    float4 mat_grad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 radiance = mk3(0.0f);
    // (Only the pixel's draw is packed in this kernel, pixel_ray.  Drawing the vertex's numbers in packed passes — all seven at once as the path kernels do,
    // or group by group: (light, triangle) in one pass, either 2-D draw as an index pass + one packed pass, six passes instead of nine — costs it more in
    // spills at 128 VGPRs than the passes save: 1.12 -> 1.18 ms in round 3, 0.873 / 1.05 -> 0.883 / 1.10 ms in round 4.)
    float u_pick = sampler_next<SK>(C, smp);
    LightSample light = sample_light<ENV>(S, it.p, u_pick, [&]() { return sampler_next<SK>(C, smp); }, [&]() { return sampler_next2<SK>(C, smp); });
    COUNT(C_SHADOW);
    bool occluded = A::any_shadow(S, lds, it.p, light.wi, 1e-4f, light.dist);
    Onb onb = make_onb(it.ns);
    f3 wo = to_local(onb, -d);
    f3 wil = to_local(onb, light.wi);
    if (!occluded && wil.z > 0.0f) {                                              // direct.py:49
        GgxTerms g = ggx_terms(wo, wil, roughness);
        f3 bsdf = ggx_brdf_from(g, wil, diffuse);
        float pdf_bsdf = ggx_pdf_from(g, wo, wil);
        float mis = balanced_heuristic(light.pdf, pdf_bsdf);
        float inv_dn = rcp(fmaxf(light.pdf, 1e-4f));
        radiance = radiance + ((bsdf * mis) * light.eval) * inv_dn;
        if (BWD) {
            f3 W = (light.eval * mis) * inv_dn;
            float dl_; float4 gr = brdf_grad(wil.z * ZDR_INV_PI, ggx_dfdr_from(g, wo, wil, roughness, dl_), W * le_grad);
            mat_grad.x += gr.x; mat_grad.y += gr.y; mat_grad.z += gr.z; mat_grad.w += gr.w;
        }
    }
    // use_MIS = True (direct.py:14): one BSDF sample, emitter lookup only
    float u_lobe = sampler_next<SK>(C, smp);
    f2 u_dir = sampler_next2<SK>(C, smp);
    f3 wi_local = ggx_sample(wo, roughness, u_lobe, u_dir);
    f3 wi = to_world(onb, wi_local);
    if (!(dot(wi, it.ng) < 1e-4f || wi_local.z < 1e-4f)) {
        f3 o2 = offset_ray_origin(it.p, it.ng);
        COUNT(C_CLOSEST);
        Hit h2 = A::closest(S, lds, o2, wi, 0.0f, 1e30f);
        f3 em = mk3(0.0f); float pdf_light = 0.0f; bool lit = false;
        if (h2.slot >= 0) {
            COUNT(C_HITS);
            Interaction it2 = surface_interact(S, h2);
            if (!(dot(-wi, it2.ng) < 1e-4f || dot(-wi, it2.ns) < 1e-4f)) {
                em = xyz(S.emission4[it2.inst]);
                pdf_light = sample_light_pdf<ENV>(S, it.p, it2.inst, h2.slot, it2.p);   // origin = it.p (direct.py:66)
                lit = true;
            }
        } else if (ENV && S.env_count > 0) {                                      // direct.py:68-71
            em = env_lookup(S, direction_to_uv(wi));
            pdf_light = env_sampled_light_pdf(S, wi, S.env_count + S.light_count);
            lit = true;
        }
        if (lit && (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f)) {
            GgxTerms g = ggx_terms(wo, wi_local, roughness);
            float pdf_bsdf = ggx_pdf_from(g, wo, wi_local);
            float mis = balanced_heuristic(pdf_bsdf, pdf_light);
            float inv_p = rcp(pdf_bsdf);
            f3 beta = ggx_brdf_from(g, wi_local, diffuse) * inv_p;
            COUNT(C_EMIT_BSDF);
            radiance = radiance + (beta * mis) * em;
            if (BWD) {
                float dl_; float4 gr = brdf_grad(wi_local.z * ZDR_INV_PI, ggx_dfdr_from(g, wo, wi_local, roughness, dl_), (em * (mis * inv_p)) * le_grad);
                mat_grad.x += gr.x; mat_grad.y += gr.y; mat_grad.z += gr.z; mat_grad.w += gr.w;
            }
        }
    }
    if (BWD) { grad = mat_grad; guv = it.uv; }
    return radiance;
}

// ---------------------------------------------------------------------------------- path
// One shaded vertex as the adjoint sweep needs it.  With the per-event
// derivative factors evaluated during the walk, the sweep is a handful of FMAs per vertex:
//   grad_k = d f^L_k [ bW g ] + d f_k [ bpq Li_{k+1} g ],   Li_k = fLW + T Li_{k+1}
struct PathVertex {
    f2 uv;
    f3 bW;  float cL, dfLdr;     // beta_k * W_k;  wiL.z/pi;  d(f^L cos)/dr      (0 when NEE rejected)
    f3 neeM;                     // beta_k f^L W_k * [pdf_bsdf/(pdf_light+pdf_bsdf) * dln(pdf_bsdf)/dr]: -d(MIS weight)/dr of the NEE term
    f3 bpq; float c, dfdr;       // beta_k/(p_k q_k);  wi.z/pi;  d(f cos)/dr     (0 when the path stops here)
    f3 T, fLW;                   // f_k/(p_k q_k);  f^L_k * W_k
    float dlnp;                  // dln(pdf_bsdf)/dr of the sampled direction
    int rr;                      // Russian roulette at this vertex: 0 none / q = 0.05, 1 stochastic (0.05 <= lum < 1), 2 renormalising (lum >= 1)
    f3 bnorm;                    // beta leaving the vertex when rr == 2
};

// Per-lane path state of the flat loop.
struct PathState {
    f3 o, d, beta, L;
    float pdf_bsdf;
    int depth;
    Sampler smp;
};

// One bounce of prb.py:23-87 (`for depth in range(max_depth)`) in two halves, so that the kernels can
// order a trip as  shade vertex (its two rays traced inside) -> classify the new hit  and every live lane
// enters a trip with a vertex to shade (zdr_kernels.hip).
//
// path_arrive: what the ray (ps.o, ps.d) reached (prb.py:25-46).  Returns true when the path ends here
// (miss, back face, emitter, untextured instance); otherwise `it` is the vertex to shade.  BWD: sets
// term_Li (and the MIS-weight fraction of the terminal emitter) when the path ended on a light.
template <bool BWD, bool STATS, bool ENV>
ZD bool path_arrive(const DScene &S, PathState &ps, const Hit &h, Interaction &it, f3 &term_Li, Counters &cnt, float *term_plfrac = nullptr) {
    if (h.slot < 0) {                                                             // prb.py:26-32, in the form of direct.py:70-83
        if (ENV && S.env_count > 0) {
            f3 em = env_lookup(S, direction_to_uv(ps.d));
            float pdf_light = env_sampled_light_pdf(S, ps.d, S.env_count + S.light_count);
            float mis = balanced_heuristic(ps.pdf_bsdf, pdf_light);
            ps.L = ps.L + (ps.beta * mis) * em;
            if (BWD) { term_Li = em * mis;
                       if (term_plfrac) *term_plfrac = (ps.pdf_bsdf + pdf_light > 1e-4f) ? pdf_light * rcp(ps.pdf_bsdf + pdf_light) : 0.0f; }
        }
        return true;
    }
    COUNT(C_HITS);
    it = surface_interact(S, h);
    if (dot(-ps.d, it.ng) < 1e-4f || dot(-ps.d, it.ns) < 1e-4f) return true;      // prb.py:35-36
    f3 em = xyz(S.emission4[it.inst]);
    if (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) {                              // prb.py:39-44
        float pdf_light = sample_light_pdf<ENV>(S, ps.o, it.inst, h.slot, it.p);
        float mis = balanced_heuristic(ps.pdf_bsdf, pdf_light);
        ps.L = ps.L + (ps.beta * mis) * em;
        if (BWD) { term_Li = em * mis;
                   if (term_plfrac) *term_plfrac = (ps.pdf_bsdf + pdf_light > 1e-4f) ? pdf_light * rcp(ps.pdf_bsdf + pdf_light) : 0.0f; }
        if (STATS && ps.depth > 0) cnt.c[C_EMIT_BSDF]++;
        return true;
    }
    if (it.inst > 0) return true;                                                 // prb.py:45-46
    return false;
}

// path_shade: next-event estimation, BSDF sampling, Russian roulette and the tracing of both rays of the
// vertex `it` (prb.py:47-87 and the trace_closest of the next loop trip, prb.py:25).  Returns true when the path
// stops at this vertex; otherwise (ps.o, ps.d) is the continuation ray and — after path_continue — `h` what it hit.
// The sampler draws always come in the reference's order (light pick, light point, lobe, direction, roulette).
// A::kFuseRays selects WHEN the shadow ray is traced:
//   false (brute force)  shadow ray -> NEE -> BSDF sampling, as the reference; the continuation ray is traced by
//                        path_continue, which the backward kernel calls once the vertex record has left the registers;
//   true  (BVH)          NEE arithmetic and BSDF sampling first, then both rays in ONE traversal loop (a lane starts its
//                        continuation ray as soon as its shadow ray has ended; no shadow ray at all when the light
//                        sample carries nothing), then the NEE terms are added if the shadow ray came through.
// BWD: fills pv, the record of this vertex.
struct ShadeCtx { f3 diffuse; float roughness; Onb onb; f3 wo, wil; LightSample light; float u_lobe; f2 u_dir; uint32_t i_rr; bool pre; };   // pre: the BSDF sample's numbers were drawn with the light's (cmj_vertex_samples)
// The light sample's contribution AS IF it were unoccluded (prb.py:60-66); applied once the shadow ray is known to be free.
struct NeeTerms { f3 dL, bW, fLW, neeM; float cL, dfLdr; };

// material, frame and the light sample of the vertex (prb.py:47-58); BWD: resets pv
template <int SK, bool BWD, bool STATS, bool ENV>
ZD ShadeCtx shade_ctx(const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, PathState &ps, const Interaction &it, PathVertex &pv, Counters &cnt) {
    ShadeCtx x;
    float4 m = read_bsdf(io.material, it.uv, R.tex_h, R.tex_w);
    x.diffuse = mk3(m.x, m.y, m.z); x.roughness = m.w;
    COUNT(C_SHADED);
    if (BWD) {
        pv.uv = it.uv; pv.bW = mk3(0.0f); pv.cL = 0.0f; pv.dfLdr = 0.0f; pv.bpq = mk3(0.0f); pv.c = 0.0f; pv.dfdr = 0.0f;
        pv.T = mk3(0.0f); pv.fLW = mk3(0.0f); pv.bnorm = mk3(0.0f); pv.neeM = mk3(0.0f); pv.dlnp = 0.0f; pv.rr = 0;
    }
    x.onb = make_onb(it.ns);
    x.wo = to_local(x.onb, -ps.d);
    // next-event estimation: the light sample (prb.py:57-58)
    x.pre = (SK == 0) && !ENV && cmj_can_batch(C);               // wave-uniform
    x.u_lobe = 0.0f; x.u_dir.x = 0.0f; x.u_dir.y = 0.0f; x.i_rr = 0u;
    if (x.pre) {                                                 // all seven numbers of the vertex at once, two permutations per register (sampler.h)
        const VertexSamples v = cmj_vertex_samples(C, ps.smp);
        x.u_lobe = v.u_lobe; x.u_dir = v.u_dir; x.i_rr = v.i_rr;
        x.light = sample_light<ENV>(S, it.p, v.u_pick, [&]() { return v.u_prim; }, [&]() { return v.u_pt; });
    } else {
        float u_pick = sampler_next<SK>(C, ps.smp);
        x.light = sample_light<ENV>(S, it.p, u_pick, [&]() { return sampler_next<SK>(C, ps.smp); }, [&]() { return sampler_next2<SK>(C, ps.smp); });
    }
    x.wil = to_local(x.onb, x.light.wi);
    return x;
}

template <bool BWD>
ZD NeeTerms nee_terms(const ShadeCtx &x, f3 beta_in) {                            // beta_in: throughput arriving at the vertex
    NeeTerms n; n.dL = n.bW = n.fLW = n.neeM = mk3(0.0f); n.cL = 0.0f; n.dfLdr = 0.0f;
    GgxTerms g = ggx_terms(x.wo, x.wil, x.roughness);
    f3 bsdf = ggx_brdf_from(g, x.wil, x.diffuse);
    float pb = ggx_pdf_from(g, x.wo, x.wil);
    float mis = balanced_heuristic(x.light.pdf, pb);
    float inv_dn = rcp(fmaxf(x.light.pdf, 1e-4f));
    n.dL = (((beta_in * bsdf) * mis) * x.light.eval) * inv_dn;
    if (BWD) {
        f3 W = (x.light.eval * mis) * inv_dn;
        float dlnpL;
        n.bW = beta_in * W; n.cL = x.wil.z * ZDR_INV_PI; n.dfLdr = ggx_dfdr_from(g, x.wo, x.wil, x.roughness, dlnpL);
        n.fLW = bsdf * W;
        float pbf = (x.light.pdf + pb > 1e-4f) ? pb * rcp(x.light.pdf + pb) : 0.0f;   // d w_nee/dr = -w_nee pb/(pl+pb) dln(pb)/dr
        n.neeM = ((beta_in * bsdf) * W) * (pbf * dlnpL);
    }
    return n;
}
template <bool BWD>
ZD void nee_apply(PathState &ps, PathVertex &pv, const NeeTerms &n) {
    ps.L = ps.L + n.dL;
    if (BWD) { pv.bW = n.bW; pv.cL = n.cL; pv.dfLdr = n.dfLdr; pv.fLW = n.fLW; pv.neeM = n.neeM; }
}

// BSDF sampling and Russian roulette (prb.py:69-87); true = the path stops here, else (ps.o, ps.d) is the continuation ray
template <int SK, bool BWD>
ZD bool sample_bsdf(const RenderCfg &R, const SamplerCfg &C, const ShadeCtx &x, PathState &ps, const Interaction &it, PathVertex &pv) {
    float u_lobe = x.u_lobe; f2 u_dir = x.u_dir;
    if (!x.pre) { u_lobe = sampler_next<SK>(C, ps.smp); u_dir = sampler_next2<SK>(C, ps.smp); }
    f3 wi_local = ggx_sample(x.wo, x.roughness, u_lobe, u_dir);
    GgxTerms g = ggx_terms(x.wo, wi_local, x.roughness);
    ps.pdf_bsdf = ggx_pdf_from(g, x.wo, wi_local);
    f3 wi = to_world(x.onb, wi_local);
    bool stop = (dot(wi, it.ng) < 1e-4f) || (wi_local.z < 1e-4f);             // prb.py:73-74
    const f3 beta_in = ps.beta;
    float q = 1.0f;
    int rr_kind = 0;
    if (!stop) {
        ps.o = offset_ray_origin(it.p, it.ng); ps.d = wi;
        f3 f = ggx_brdf_from(g, wi_local, x.diffuse);
        float inv_p = rcp(ps.pdf_bsdf);
        ps.beta = ps.beta * (f * inv_p);
        if (ps.depth >= R.rr_depth) {                                         // prb.py:79-87
            float l = 0.212671f * ps.beta.x + 0.715160f * ps.beta.y + 0.072169f * ps.beta.z;
            if (l == 0.0f) stop = true;
            else {
                q = fmaxf(l, 0.05f);
                float r = x.pre ? cmj_next_with_index(C, ps.smp, x.i_rr) : sampler_next<SK>(C, ps.smp);
                if (r >= q) stop = true;
                else { ps.beta = ps.beta * rcp(q); rr_kind = (l >= 1.0f) ? 2 : ((l >= 0.05f) ? 1 : 0); }
            }
        }
        if (BWD && !stop) {
            float inv_pq = inv_p * rcp(q);
            pv.bpq = beta_in * inv_pq; pv.c = wi_local.z * ZDR_INV_PI; pv.dfdr = ggx_dfdr_from(g, x.wo, wi_local, x.roughness, pv.dlnp);
            pv.T = f * inv_pq;
            // prb.py:83 has no upper clamp on q: for lum(beta') >= 1 the path survives with certainty and is
            // STILL divided by q = lum(beta'), i.e. beta leaves this vertex with unit luminance.  The forward's
            // expectation then depends on q(material); the sweep differentiates through it (sweep_vertex).
            pv.rr = rr_kind;
            if (rr_kind == 2) pv.bnorm = ps.beta;
            // ZDR_PRB_LITERAL (prb.py:157-163): the roulette fields go unused; bnorm carries beta f / pdf, the part of the literal seed
            // beta / pdf * Le that is not already in Q and in the arriving radiance (sweep_vertex)
            if (R.prb_mode == ZDR_PRB_LITERAL) pv.bnorm = beta_in * (f * inv_p);   // (pv.rr keeps the roulette kind for zdr_path_dump's flags; pack_vertex ignores it in this mode)
        }
    }
    ps.depth++;
    if (ps.depth >= R.max_depth) stop = true;
    return stop;
}

// Everything of a vertex that comes BEFORE its rays are traced, in the order the BVH kernels use: the NEE arithmetic
// first, so that only its results (not the frame, the light sample and the material) have to live through the
// traversal — and so that the shadow ray can be SKIPPED when nothing rides on it: a light sample below the horizon
// (prb.py:62) or one that carries exactly nothing even if visible (a light seen from behind or edge-on has eval = 0,
// light.py:76).  On the Cornell box that is every vertex of the ceiling.  Exact: only all-zero, NaN-free terms are dropped,
// so the image and the gradients are the reference's bit for bit.  Returns the shadow segment (if any) and whether the
// path stops at this vertex; the continuation ray is (ps.o, ps.d).
struct VertexRays { bool shadow, stop; f3 sd; float stmax; };
template <int SK, bool BWD, bool STATS, bool ENV>
ZD VertexRays path_vertex_begin(const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io,
                                PathState &ps, const Interaction &it, PathVertex &pv, NeeTerms &n, Counters &cnt) {
    const ShadeCtx x = shade_ctx<SK, BWD, STATS, ENV>(S, R, C, io, ps, it, pv, cnt);
    VertexRays vr; vr.sd = x.light.wi; vr.stmax = x.light.dist;
    n.dL = n.bW = n.fLW = n.neeM = mk3(0.0f); n.cL = 0.0f; n.dfLdr = 0.0f;
    COUNT(C_SHADOW);
    vr.shadow = x.wil.z >= 1e-4f;
    if (vr.shadow) {
        n = nee_terms<BWD>(x, ps.beta);
        const bool nothing = (n.dL.x == 0.0f) & (n.dL.y == 0.0f) & (n.dL.z == 0.0f) &&
                             (!BWD || ((n.bW.x == 0.0f) & (n.bW.y == 0.0f) & (n.bW.z == 0.0f) & (n.fLW.x == 0.0f) & (n.fLW.y == 0.0f) & (n.fLW.z == 0.0f) &
                                       (n.neeM.x == 0.0f) & (n.neeM.y == 0.0f) & (n.neeM.z == 0.0f) & (fabsf(n.dfLdr) < 3.0e38f)));
        vr.shadow = !nothing;
    }
    if (vr.shadow) COUNT(C_SHADOW_TRACED);                                          // counter 7: shadow rays actually traced
    vr.stop = sample_bsdf<SK, BWD>(R, C, x, ps, it, pv);
    if (!vr.stop) COUNT(C_CLOSEST);
    return vr;
}

template <int SK, class A, bool BWD, bool STATS, bool ENV>
ZD bool path_shade(const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int *lds,
                   PathState &ps, const Interaction &it, PathVertex &pv, Hit &h, Counters &cnt) {
    if constexpr (A::kFuseRays) {
        NeeTerms n;
        const VertexRays vr = path_vertex_begin<SK, BWD, STATS, ENV>(S, R, C, io, ps, it, pv, n, cnt);
        bool occluded;
        A::shadow_and_closest(S, lds, vr.shadow, it.p, vr.sd, 1e-4f, vr.stmax, !vr.stop, ps.o, ps.d, occluded, h);
        if (vr.shadow && !occluded) nee_apply<BWD>(ps, pv, n);
        return vr.stop;
    } else {
        const ShadeCtx x = shade_ctx<SK, BWD, STATS, ENV>(S, R, C, io, ps, it, pv, cnt);
        COUNT(C_SHADOW);
        const bool occluded = A::any_shadow(S, lds, it.p, x.light.wi, 1e-4f, x.light.dist);
        if (!occluded && x.wil.z >= 1e-4f) nee_apply<BWD>(ps, pv, nee_terms<BWD>(x, ps.beta));
        COUNT(C_SHADOW_TRACED);
        return sample_bsdf<SK, BWD>(R, C, x, ps, it, pv);   // the caller traces the continuation ray (path_continue), after it has put pv away
    }
}

// The continuation ray of a vertex whose path goes on: already traced by path_shade when A::kFuseRays.
template <class A, bool STATS>
ZD void path_continue(const DScene &S, int *lds, const PathState &ps, Hit &h, Counters &cnt) {
    if constexpr (!A::kFuseRays) { COUNT(C_CLOSEST); h = A::closest(S, lds, ps.o, ps.d, 0.0f, 1e30f); }
}

// ---- primary queue -----------------------------------------------------------------------------
// Camera rays are generated and traced in BATCHES by the whole wave — one sample index for all 64
// pixels of the tile per step, every lane busy — classified, and the vertices worth shading are parked
// in a wave-wide FIFO in global memory.  Whenever a lane's path ends it takes the next parked vertex,
// WHATEVER pixel it belongs to: lanes do not own pixels in the flat loop, so no lane idles because
// "its" pixel happened to have short paths (with one pixel per lane a wave ran until its slowest lane
// had finished, 25 % of all trips on cbox), and a path of k shaded vertices costs k trips, not k + 1.
// Entry e of block b: two float4 at queue[(b * CAP * 64 + e % (CAP * 64)) * 2 + {0, 1}]
//   {d.xyz, u} {v, bits(slot), bits(sampler LCG state), bits(pixel-in-tile << 26 | item bank << 25 | sample index)}
// Pushes and pops are compacted with ballot/mbcnt, so both sides touch consecutive entries.
struct PrimaryQueue { float4 *base; uint32_t head, tail; };    // head, tail: wave-uniform entry counters
#define ZDR_QUEUE_ENTRIES (ZDR_RING_CAP * 64)
static_assert(ZDR_RING_BATCH * 64 + 64 <= ZDR_QUEUE_ENTRIES, "a refill must fit behind a queue that could not serve every idle lane");

ZD PrimaryQueue queue_init(const KernelIO &io) {
    PrimaryQueue r; r.base = io.ring + (size_t)blockIdx.x * (ZDR_QUEUE_ENTRIES * 2); r.head = 0; r.tail = 0;
    return r;
}
ZD uint32_t lane_rank(unsigned long long mask) {               // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Generates ZDR_RING_BATCH camera samples for every pixel of the tile (lane = pixel here).  Samples that end
// at the camera ray (miss, emitter, back face) are finished at once: their radiance goes to `sum`,
// this lane's own pixel.
template <int SK, class A, bool BWD, bool STATS, bool ENV>
ZD void primary_refill(const DScene &S, const RenderCfg &R, const SamplerCfg &C, int *lds, int x, int y, bool valid, unsigned long long cam_mask,
                       uint32_t perm_seed, int bank, uint32_t &next_sample, uint32_t s_end, PrimaryQueue &q, f3 &sum, Counters &cnt) {
    for (int b = 0; b < ZDR_RING_BATCH && next_sample < s_end; b++, next_sample++) {   // wave-uniform
        bool park = false;
        float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0;
        if (valid) {
            PathState ps;
            ps.smp = sampler_make<SK>(C, (uint32_t)x, (uint32_t)y, perm_seed, next_sample);
            pixel_ray<SK>(R, C, ps.smp, x, y, ps.o, ps.d);
            ps.beta = mk3(1.0f); ps.L = mk3(0.0f); ps.pdf_bsdf = 1e30f; ps.depth = 0;   // prb.py:20-22
            COUNT(C_SAMPLES); COUNT(C_CLOSEST);
            Hit h = A::closest_camera(S, lds, ps.o, ps.d, cam_mask);
            Interaction it; f3 tl;
            if (path_arrive<false, STATS, ENV>(S, ps, h, it, tl, cnt)) {
                if (!BWD) {                                                     // a path without vertices has no gradient
                    if (!any_nan(ps.L)) sum = sum + clamp_radiance(ps.L);       // integrator.py:27-28
                    else COUNT(C_NAN);
                }
            } else {
                park = true;
                e0 = make_float4(ps.d.x, ps.d.y, ps.d.z, h.u);
                e1 = make_float4(h.v, __int_as_float(h.slot), __uint_as_float(ps.smp.state), __uint_as_float(((uint32_t)threadIdx.x << 26) | ((uint32_t)bank << 25) | next_sample));
            }
        }
        const unsigned long long m = __ballot(park);
        if (park) {
            float4 *e = q.base + (size_t)((q.tail + lane_rank(m)) % ZDR_QUEUE_ENTRIES) * 2;
            e[0] = e0; e[1] = e1;
        }
        q.tail += (uint32_t)__popcll(m);
    }
    // entries are read back by OTHER lanes of this wave: order the stores before the later loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// An idle lane takes the oldest parked vertex that no lower idle lane takes: the path state as it is right
// after the camera ray.  Returns bank * 64 + pixel (lane index within the tile) of the path, or -1.
// lds_perm[bank * 64 + pixel]: CMJ seed of the pixel; lds_origin[bank * 2 + {0, 1}]: first pixel of the bank's tile.
template <int SK>
ZD int primary_pop(const DScene &S, const SamplerCfg &C, bool idle, const uint32_t *lds_perm, const int *lds_origin,
                   PrimaryQueue &q, PathState &ps, Interaction &it) {
    const unsigned long long m = __ballot(idle);
    const uint32_t avail = q.tail - q.head, rank = lane_rank(m), want = (uint32_t)__popcll(m);
    const bool take = idle && rank < avail;
    int pix = -1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (take) {
        const float4 *e = q.base + (size_t)((q.head + rank) % ZDR_QUEUE_ENTRIES) * 2;
        float4 a = e[0], b = e[1];
        Hit h; h.slot = __float_as_int(b.y); h.u = a.w; h.v = b.x; h.t = 0.0f;
        it = surface_interact(S, h);
        ps.d = mk3(a.x, a.y, a.z); ps.o = mk3(0.0f);
        ps.beta = mk3(1.0f); ps.L = mk3(0.0f); ps.pdf_bsdf = 1e30f; ps.depth = 0;
        const uint32_t key = __float_as_uint(b.w);
        const int p = (int)(key >> 26), bank = (int)((key >> 25) & 1u);
        pix = bank * 64 + p;
        ps.smp.px = (uint32_t)(lds_origin[bank * 2] + (p & 7)); ps.smp.py = (uint32_t)(lds_origin[bank * 2 + 1] + (p >> 3));
        ps.smp.sample_index = key & 0x01ffffffu;
        ps.smp.dimension = 2;                                                   // pixel_ray drew one 2-D sample
        ps.smp.permutation_seed = lds_perm ? lds_perm[pix] : ((SK == 0) ? xxhash32_4(ps.smp.px, ps.smp.py, C.seed, 0u) : 0u);   // (no LDS copy: the pixel's seed is hashed again)
        ps.smp.state = __float_as_uint(b.z);
    }
    q.head += (want < avail) ? want : avail;
    return pix;
}

// The vertex as the sweep stores it: four float4 (+ a fifth for vertices at depth >= rr_depth).  The NEE
// part of the gradient does not depend on the rest of the path, so it is finished at once (contracted
// with the pixel cotangent g, MIS-weight derivative included); the BSDF part keeps what multiplies the
// arriving adjoint:
//   a = d f^L[bW g] - (0,0,0,<g, neeM>)  (4)   NEE gradient of this vertex
//   b = {Q = bpq c, r = dfdr / c}        (4)   BSDF gradient = (Q Aeff, r sum(Q Aeff))
//   c = {T, uv.x}, d = {g fLW, uv.y}     (8)   A_k = g fLW + T Aeff
//   e = {b = beta leaving (rr == 2) | (-1,0,0) (rr == 1) | 0, dln(pdf)/dr}
struct PackedVertex { float4 a, b, c, d, e; };

// mode (zdr.h): ZDR_PRB_DETACHED — the MIS weights and the Russian-roulette factors are constants: no neeM, no RR fields, no score;
// ZDR_PRB_LITERAL — as detached, and e.xyz = beta f / pdf for the literal BSDF-sample seed of prb.py:162
ZD PackedVertex pack_vertex(const PathVertex &v, f3 g, int mode = ZDR_PRB_EXPECTATION) {
    PackedVertex p;
    p.a = brdf_grad(v.cL, v.dfLdr, v.bW * g);
    if (mode == ZDR_PRB_EXPECTATION) p.a.w -= dot(g, v.neeM);
    f3 Q = v.bpq * v.c;
    float r = (v.c > 0.0f) ? v.dfdr * rcp(v.c) : 0.0f;
    p.b = make_float4(Q.x, Q.y, Q.z, r);
    p.c = make_float4(v.T.x, v.T.y, v.T.z, v.uv.x);
    f3 gA = g * v.fLW;
    p.d = make_float4(gA.x, gA.y, gA.z, v.uv.y);
    f3 e = (v.rr == 2) ? v.bnorm : ((v.rr == 1) ? mk3(-1.0f, 0.0f, 0.0f) : mk3(0.0f));
    p.e = make_float4(e.x, e.y, e.z, v.dlnp);
    if (mode == ZDR_PRB_DETACHED) p.e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (mode == ZDR_PRB_LITERAL) p.e = make_float4(v.bnorm.x, v.bnorm.y, v.bnorm.z, 0.0f);
    return p;
}

// Carriers of the adjoint sweep (deviation 8; derivation in oracle/zdr_oracle.c path_backward):
//   A  adjoint of the scalar <g, L> w.r.t. the throughput (g-contracted: renormalisation mixes channels)
//   Lv plain arriving radiance times g (the VALUE of what follows, for the score term)
//   s  adjoint of the survival probability;  Z  <b, Lv> of the renormalisation whose 1/pdf this vertex lost
//   tw once: d(MIS weight)/dr factor of the emitter hit that ended the path
struct SweepState { f3 A, Lv; float s, Z, tw; };

// One step (prb.py:105-187 with the corrected weight, App. B-3): consumes a vertex, returns its gradient.
ZD float4 sweep_vertex(const PackedVertex &p, SweepState &S, f2 &uv, int mode = ZDR_PRB_EXPECTATION) {
    const f3 w = mk3(0.212671f, 0.715160f, 0.072169f);           // prb.py:80
    f3 Aeff = S.A;
    if (mode == ZDR_PRB_LITERAL) {                               // prb.py:162: (beta / pdf) * Le_remaining * le_grad, Le_remaining = beta T Li
        Aeff = mk3(p.e.x, p.e.y, p.e.z) * S.Lv;
    } else if (p.e.x < 0.0f) {                                   // stochastic RR vertex
        Aeff = S.A + w * S.s; S.s = 0.0f; S.Z = 0.0f;
    } else if (p.e.x + p.e.y + p.e.z > 0.0f) {                   // renormalising RR vertex
        f3 b = mk3(p.e.x, p.e.y, p.e.z);
        float ba = dot(b, S.A);
        Aeff = S.A - w * ba; S.s += ba; S.Z = dot(b, S.Lv);
    }
    f3 ct = mk3(p.b.x, p.b.y, p.b.z) * Aeff;
    float4 g = make_float4(p.a.x + ct.x, p.a.y + ct.y, p.a.z + ct.z,
                           p.a.w + p.b.w * (ct.x + ct.y + ct.z) + p.e.w * (S.Z + S.tw));
    S.tw = 0.0f;
    f3 T = mk3(p.c.x, p.c.y, p.c.z), gA = mk3(p.d.x, p.d.y, p.d.z);
    S.A = gA + T * Aeff;
    S.Lv = gA + T * S.Lv;
    uv.x = p.c.w; uv.y = p.d.w;
    return g;
}
