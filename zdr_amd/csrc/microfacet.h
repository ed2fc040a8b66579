// microfacet.h — GGX / Trowbridge-Reitz BRDF of the reference (microfacet.py:7-92): eval (times
// cosine), 50/50 cosine + visible-normal sampling, mixed pdf, and the closed-form derivative
// w.r.t. (diffuse, roughness) that replaces luisa.autodiff.
#pragma once
#include "vecmath.h"

#define ZDR_SPECULAR 0.04f      // prb.py:52, direct.py:38, collocated.py:24

ZD float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }

// Everything the eval, the pdf and the derivative share for one (wo, wi, roughness) triple.
struct GgxTerms {
    float a2;          // alpha^2 = r^4
    float D, F, G1i, G1o;
    float t;           // nh^2 (a2 - 1) + 1
    float nh2;         // max(1e-5, h.z)^2
    float ki, ko;      // (1 - nv^2) / nv^2 for wi, wo
    float si, so;      // sqrt(1 + a2 k)
    float inv4;        // 1 / (4 max(1e-5, wi.z) max(1e-5, wo.z))
    float wo_dot_h;
};

ZD GgxTerms ggx_terms(f3 wo, f3 wi, float roughness) {
    GgxTerms g;
    float alpha = roughness * roughness;                      // microfacet.py:25
    g.a2 = alpha * alpha;
    f3 h = normalize(wi + wo);                                // :26
    float nh = fmaxf(0.00001f, h.z);                          // :10
    g.nh2 = nh * nh;
    g.t = g.nh2 * (g.a2 - 1.0f) + 1.0f;
    g.D = g.a2 * rcp(ZDR_PI * (g.t * g.t));                   // :11
    g.wo_dot_h = dot(wo, h);
    float c = clampf(g.wo_dot_h, 0.00001f, 1.0f);             // :28
    g.F = ZDR_SPECULAR + (1.0f - ZDR_SPECULAR) * pow5(1.0f - c);  // :15
    float nvi = fmaxf(0.00001f, wi.z), nvo = fmaxf(0.00001f, wo.z);  // :20
    g.ki = (1.0f - nvi * nvi) * rcp(nvi * nvi);
    g.ko = (1.0f - nvo * nvo) * rcp(nvo * nvo);
    g.si = fsqrt(1.0f + g.a2 * g.ki);
    g.so = fsqrt(1.0f + g.a2 * g.ko);
    g.G1i = 2.0f * rcp(1.0f + g.si);                          // :21
    g.G1o = 2.0f * rcp(1.0f + g.so);
    g.inv4 = rcp(4.0f * nvi * nvo);                           // :30
    return g;
}

// ggx_brdf (microfacet.py:24-30): (D F G / (4 wi.z wo.z) + diffuse / pi) * wi.z
ZD f3 ggx_brdf_from(const GgxTerms &g, f3 wi, f3 diffuse) {
    float s = (g.D * g.F * (g.G1i * g.G1o)) * g.inv4;
    return (mk3(s) + diffuse * ZDR_INV_PI) * wi.z;
}

// ggx_sample_pdf (microfacet.py:52-58, 68-69): 0.5 cos/pi + 0.5 G1(wo)/|wo.z| D |wo.wm| / (4 |wo.wm|)
ZD float ggx_pdf_from(const GgxTerms &g, f3 wo, f3 wi) {
    float a = fabsf(g.wo_dot_h);
    float pdf_wm = g.G1o * rcp(fabsf(wo.z)) * g.D * a;
    float glossy = pdf_wm * rcp(4.0f * a);
    return 0.5f * (wi.z * ZDR_INV_PI) + 0.5f * glossy;
}

// d(f cos)/d roughness, identical for the three channels:
//   4 r^3 F wi.z / (4 ci co) (D' G + D (G1i' G1o + G1i G1o')),  D' = (1 - c(1 + a2)) / (pi t^3),
//   G1' = -k / (s (1 + s)^2).  d(f cos)_c / d diffuse_c = wi.z / pi.
// Also returns d ln(ggx_sample_pdf) / d roughness (only the glossy half G1(wo) D / (4 |wo.z|) depends on
// it): the PRB adjoint needs it where Russian roulette renormalises the throughput (integrators.h).
ZD float ggx_dfdr_from(const GgxTerms &g, f3 wo, f3 wi, float roughness, float &dlnpdf_dr) {
    float t3 = g.t * g.t * g.t;
    float dD = (1.0f - g.nh2 * (1.0f + g.a2)) * rcp(ZDR_PI * t3);
    float opi = 1.0f + g.si, opo = 1.0f + g.so;
    float dG1i = -g.ki * rcp(g.si * (opi * opi));
    float dG1o = -g.ko * rcp(g.so * (opo * opo));
    float dS = dD * (g.G1i * g.G1o) + g.D * (dG1i * g.G1o + g.G1i * dG1o);
    float r3 = roughness * roughness * roughness;
    float dglossy = (dG1o * g.D + g.G1o * dD) * rcp(4.0f * fabsf(wo.z));
    dlnpdf_dr = (0.5f * dglossy * 4.0f * r3) * rcp(ggx_pdf_from(g, wo, wi));
    return 4.0f * r3 * (g.F * wi.z * g.inv4) * dS;
}

// Both lobes of ggx_sample warp the same polar point of the unit disk: r = sqrt(u.x), angle 2 pi u.y
// (cosine_sample_hemisphere, microfacet.py:34-37, and SampleUniformDiskPolar, :61-65).  In a wavefront
// the two branches are both executed (the lobe is chosen per lane), so the disk point — one sincos —
// is computed once, before the branch.
ZD f3 sample_wm_disk(f3 w, float alpha, float px, float py) {      // microfacet.py:72-92 (pbrt-v4 VNDF)
    f3 wh = normalize(mk3(alpha * w.x, alpha * w.y, w.z));
    if (wh.z < 0.0f) wh = -wh;
    f3 T1 = (wh.z < 0.99999f) ? normalize(cross(mk3(0.0f, 0.0f, 1.0f), wh)) : mk3(1.0f, 0.0f, 0.0f);
    f3 T2 = cross(wh, T1);
    float h = fsqrt(1.0f - px * px);
    py = lerpf(h, py, (1.0f + wh.z) * 0.5f);
    float pz = fsqrt(fmaxf(0.0f, 1.0f - (px * px + py * py)));
    f3 nh = px * T1 + py * T2 + pz * wh;
    return normalize(mk3(alpha * nh.x, alpha * nh.y, fmaxf(1e-6f, nh.z)));
}

ZD f3 ggx_sample(f3 wo, float roughness, float u_lobe, f2 u2) {  // microfacet.py:41-49
    float r = fsqrt(u2.x), phi = 2.0f * ZDR_PI * u2.y;
    float sn, cs;
    sincosf(phi, &sn, &cs);
    float px = r * cs, py = r * sn;
    if (u_lobe < 0.5f) return mk3(px, py, fsqrt(1.0f - u2.x));    // cosine lobe
    f3 wm = sample_wm_disk(wo, roughness * roughness, px, py);
    f3 i = -wo;                                                 // reflect(-wo, wm)
    return i - wm * (2.0f * dot(wm, i));
}
