// zdr_api.cpp — host side of libzdr_hip.so: scene assembly (what LuisaCompute's Accel / heap did
// for render.py:73-128), a binned-SAH BVH2 builder, launch configuration, and the C-ABI of
// include/zdr.h.  Compiled with hipcc -ffp-contract=off: per-triangle constants (edges, geometric
// normal, area, camera frame) are then plain IEEE float32 and reproducible on any host.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "internal.h"
#include "zdr.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ZDR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" const char *zdr_version(void) { return ZDR_VERSION_STRING; }
extern "C" int zdr_abi_version(void) { return ZDR_ABI_VERSION; }
extern "C" const char *zdr_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------ host vec3
struct h3 { float x, y, z; };
static inline h3 H3(float x, float y, float z) { return {x, y, z}; }
static inline h3 hsub(h3 a, h3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline h3 hcross(h3 a, h3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float hdot(h3 a, h3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline h3 hscale(h3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline h3 hnormalize(h3 a) { return hscale(a, 1.0f / sqrtf(hdot(a, a))); }
static inline h3 xform_point(const float *m, h3 v) {      // (M * float4(v,1)).xyz, interaction.py:19-21
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7], m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]};
}
static void normal_matrix(const float *m, float *n) {      // inverse(transpose(M3x3)), interaction.py:28
    float a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
    float c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
    float inv = 1.0f / (a * c00 + b * c01 + c * c02);
    n[0] = c00 * inv; n[1] = c01 * inv; n[2] = c02 * inv; n[3] = c10 * inv; n[4] = c11 * inv; n[5] = c12 * inv;
    n[6] = c20 * inv; n[7] = c21 * inv; n[8] = c22 * inv;
}

// ------------------------------------------------------------------------------ BVH builder
// Binned SAH (ZDR_BVH_BINS bins, 3 axes), leaves of <= ZDR_BVH_LEAF triangles, depth bounded by the traversal stack.
#ifndef ZDR_BVH_BINS
#define ZDR_BVH_BINS 32   // 1 M triangles, path fwd / bwd ms at 1024^2 spp 32: 16 bins 33.9 / 44.1, 32 bins 33.5 / 43.7
#endif
// (ZDR_BVH_LEAF lives in scene.h: the walk compiles its generic leaf loop only when leaves can hold more than two triangles)
static_assert(ZDR_BVH_LEAF >= 1 && ZDR_BVH_LEAF <= 6, "the child word holds the leaf size in 3 bits, 7 = unused");
#ifndef ZDR_BVH_BFS_NODES
#define ZDR_BVH_BFS_NODES 341   // root + four levels of a full BVH4: numbered breadth-first, so that "node id < K" is the top of the tree for any K up to here
#endif
struct BNode { float lo[3], hi[3]; int left, right, first, count; };
struct Prim { float lo[3], hi[3], c[3]; int tri; };

struct BvhBuilder {
    std::vector<Prim> prims;
    std::vector<BNode> nodes;
    int max_depth = 0;
    static constexpr int kLeaf = ZDR_BVH_LEAF, kBins = ZDR_BVH_BINS, kDepthLimit = ZDR_BVH_STACK - 2;

    static void grow(float *lo, float *hi, const float *plo, const float *phi) {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], plo[k]); hi[k] = std::max(hi[k], phi[k]); }
    }
    static int bin_of(float f) {     // float -> bin, clamped; a NaN centroid (garbage input) lands in bin 0 instead of an undefined conversion
        return (f >= 0.0f) ? ((f < (float)kBins) ? (int)f : kBins - 1) : 0;
    }
    static float area(const float *lo, const float *hi) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
    int build(int b, int e, int depth) {
        int id = (int)nodes.size();
        nodes.push_back(BNode());
        max_depth = std::max(max_depth, depth);
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        float clo[3] = {3e38f, 3e38f, 3e38f}, chi[3] = {-3e38f, -3e38f, -3e38f};
        for (int i = b; i < e; i++) { grow(lo, hi, prims[i].lo, prims[i].hi); grow(clo, chi, prims[i].c, prims[i].c); }
        BNode nd; memcpy(nd.lo, lo, 12); memcpy(nd.hi, hi, 12); nd.left = nd.right = -1; nd.first = b; nd.count = e - b;
        int n = e - b;
        if (n <= kLeaf) { nodes[id] = nd; return id; }
        int mid = -1;
        // levels still needed if we split at the median from here on
        int need = 0; for (int m = (n + kLeaf - 1) / kLeaf; m > 1; m = (m + 1) / 2) need++;
        bool force_median = depth + need + 1 >= kDepthLimit;
        if (!force_median) {
            float best = 3e38f; int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                float ext = chi[ax] - clo[ax];
                if (!(ext > 0.0f)) continue;
                float blo[kBins][3], bhi[kBins][3]; int bc[kBins];
                for (int k = 0; k < kBins; k++) { bc[k] = 0; for (int j = 0; j < 3; j++) { blo[k][j] = 3e38f; bhi[k][j] = -3e38f; } }
                float scale = (float)kBins / ext;
                for (int i = b; i < e; i++) {
                    int k = bin_of((prims[i].c[ax] - clo[ax]) * scale);
                    bc[k]++; grow(blo[k], bhi[k], prims[i].lo, prims[i].hi);
                }
                float rlo[kBins][3], rhi[kBins][3]; int rc[kBins];
                float alo[3] = {3e38f, 3e38f, 3e38f}, ahi[3] = {-3e38f, -3e38f, -3e38f}; int ac = 0;
                for (int k = kBins - 1; k > 0; k--) { if (bc[k]) grow(alo, ahi, blo[k], bhi[k]); ac += bc[k]; memcpy(rlo[k], alo, 12); memcpy(rhi[k], ahi, 12); rc[k] = ac; }
                float llo[3] = {3e38f, 3e38f, 3e38f}, lhi[3] = {-3e38f, -3e38f, -3e38f}; int lc = 0;
                for (int k = 0; k < kBins - 1; k++) {
                    if (bc[k]) grow(llo, lhi, blo[k], bhi[k]); lc += bc[k];
                    if (lc == 0 || rc[k + 1] == 0) continue;
                    float cost = area(llo, lhi) * (float)lc + area(rlo[k + 1], rhi[k + 1]) * (float)rc[k + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = k; }
                }
            }
            if (best_axis >= 0) {
                float ext = chi[best_axis] - clo[best_axis], scale = (float)kBins / ext;
                auto it = std::partition(prims.begin() + b, prims.begin() + e, [&](const Prim &p) {
                    int k = bin_of((p.c[best_axis] - clo[best_axis]) * scale);
                    return k <= best_bin; });
                mid = (int)(it - prims.begin());
            }
        }
        if (mid <= b || mid >= e) {   // median split along the widest centroid axis
            int ax = 0; float w = chi[0] - clo[0];
            for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > w) { w = chi[k] - clo[k]; ax = k; }
            mid = b + n / 2;
            std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e, [ax](const Prim &p, const Prim &q) { return p.c[ax] < q.c[ax]; });
        }
        int l = build(b, mid, depth + 1);
        int r = build(mid, e, depth + 1);
        nd.left = l; nd.right = r; nd.count = 0;
        nodes[id] = nd;
        return id;
    }
};

// plane-form intersection record of one triangle, float64 -> float32 (accel.h, tri_test)
static void plane_record(const h3 *p, float4 *out, float4 *w_out = nullptr) {
    double p0[3] = {p[0].x, p[0].y, p[0].z};
    double e1[3] = {(double)p[1].x - p0[0], (double)p[1].y - p0[1], (double)p[1].z - p0[2]};
    double e2[3] = {(double)p[2].x - p0[0], (double)p[2].y - p0[1], (double)p[2].z - p0[2]};
    double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    double nu[3] = {(e2[1] * n[2] - e2[2] * n[1]) / nn, (e2[2] * n[0] - e2[0] * n[2]) / nn, (e2[0] * n[1] - e2[1] * n[0]) / nn};
    double nv[3] = {(n[1] * e1[2] - n[2] * e1[1]) / nn, (n[2] * e1[0] - n[0] * e1[2]) / nn, (n[0] * e1[1] - n[1] * e1[0]) / nn};
    out[0] = make_float4((float)n[0], (float)n[1], (float)n[2], (float)(n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]));
    out[1] = make_float4((float)nu[0], (float)nu[1], (float)nu[2], (float)-(nu[0] * p0[0] + nu[1] * p0[1] + nu[2] * p0[2]));
    out[2] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], (float)-(nv[0] * p0[0] + nv[1] * p0[1] + nv[2] * p0[2]));
    if (w_out) {    // w = 1 - u - v as an edge function of its own (the third edge of a triangle that found no partner, find_quads)
        const double du = -(nu[0] * p0[0] + nu[1] * p0[1] + nu[2] * p0[2]), dv = -(nv[0] * p0[0] + nv[1] * p0[1] + nv[2] * p0[2]);
        *w_out = make_float4((float)-(nu[0] + nv[0]), (float)-(nu[1] + nv[1]), (float)-(nu[2] + nv[2]), (float)(1.0 - du - dv));
    }
}

// Brute-force scenes: pairs of triangles that form a PLANAR CONVEX QUAD are tested as one primitive (accel.h, quad test):
// one plane, one hit point, the four outer edge functions — 39 VALU for four triangles instead of 62.  Two triangles
// merge when they share an edge (the same two world positions), lie in one plane (the apex of the second at most 5e-7 of
// the quad's size off the plane of the first: a few float32 ulps of the corner coordinates themselves) on opposite sides of
// that edge, and the quad is convex (the apex of each lies strictly inside the other's two outer edges).  For each input
// triangle `rot` says which corner is the apex (= corner 0 of its slot's records: the diagonal is then the w = 0 edge of
// both triangles and the outer edges are their u = 0 and v = 0 edges); `quads` lists the merged pairs (input indices).
struct QuadPair { int a, b; bool par; };   // par: a parallelogram (the second apex = s1 + s2 - first apex)
static void find_quads(const std::vector<h3> &pos, uint32_t ntris, std::vector<int> &rot, std::vector<QuadPair> &quads) {
    rot.assign(ntris, 0); quads.clear();
    struct EdgeKey { float k[6]; bool operator<(const EdgeKey &o) const { return memcmp(k, o.k, sizeof k) < 0; } };
    auto key = [&](h3 a, h3 b) { EdgeKey e; const float A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z};
                                 const bool sw = memcmp(A, B, sizeof A) > 0; memcpy(e.k, sw ? B : A, 12); memcpy(e.k + 3, sw ? A : B, 12);
                                 for (float &f : e.k) if (f == 0.0f) f = 0.0f;   /* -0 -> +0 */ return e; };
    std::map<EdgeKey, std::vector<std::pair<int, int>>> edges;      // edge -> (triangle, corner opposite)
    for (uint32_t t = 0; t < ntris; t++) for (int e = 0; e < 3; e++) edges[key(pos[3 * (size_t)t + (e + 1) % 3], pos[3 * (size_t)t + (e + 2) % 3])].push_back({(int)t, e});
    auto D = [](h3 v, double *o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; };
    // which pairs COULD merge (t with corner e as its apex, t2 with corner e2)
    struct Cand { int t2, e, e2; bool par; };
    std::vector<std::vector<Cand>> adj(ntris);
    for (uint32_t t = 0; t < ntris; t++) {
        for (int e = 0; e < 3; e++) {
            for (const auto &cand : edges[key(pos[3 * (size_t)t + (e + 1) % 3], pos[3 * (size_t)t + (e + 2) % 3])]) {
                const int t2 = cand.first, e2 = cand.second;
                if (t2 == (int)t) continue;
                double a0[3], s1[3], s2[3], b0[3];
                D(pos[3 * (size_t)t + e], a0); D(pos[3 * (size_t)t + (e + 1) % 3], s1); D(pos[3 * (size_t)t + (e + 2) % 3], s2); D(pos[3 * (size_t)t2 + e2], b0);
                double e1[3], e2v[3], r[3], n[3];
                for (int k = 0; k < 3; k++) { e1[k] = s1[k] - a0[k]; e2v[k] = s2[k] - a0[k]; r[k] = b0[k] - a0[k]; }
                n[0] = e1[1] * e2v[2] - e1[2] * e2v[1]; n[1] = e1[2] * e2v[0] - e1[0] * e2v[2]; n[2] = e1[0] * e2v[1] - e1[1] * e2v[0];
                const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                if (!(nn > 0.0)) continue;
                const double size = sqrt(std::max(std::max(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], e2v[0] * e2v[0] + e2v[1] * e2v[1] + e2v[2] * e2v[2]), r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
                if (fabs(n[0] * r[0] + n[1] * r[1] + n[2] * r[2]) / sqrt(nn) > 5e-7 * size) continue;          // not in one plane
                // barycentrics of b0 in (a0; s1, s2): u = weight of s1, v = weight of s2
                const double u = ((e2v[1] * n[2] - e2v[2] * n[1]) * r[0] + (e2v[2] * n[0] - e2v[0] * n[2]) * r[1] + (e2v[0] * n[1] - e2v[1] * n[0]) * r[2]) / nn;
                const double v = ((n[1] * e1[2] - n[2] * e1[1]) * r[0] + (n[2] * e1[0] - n[0] * e1[2]) * r[1] + (n[0] * e1[1] - n[1] * e1[0]) * r[2]) / nn;
                if (!(u > 1e-6 && v > 1e-6 && 1.0 - u - v < -1e-6)) continue;                                   // not convex, or folded back
                // (the corners at the two apices are corners of a triangle, hence convex: nothing else to check)
                adj[t].push_back({t2, e, e2, fabs(u - 1.0) <= 2e-6 && fabs(v - 1.0) <= 2e-6});
            }
        }
    }
    // greedy matching, the triangles with the fewest possible partners first (each takes its partner with the fewest): a strip
    // of quads pairs up along its own diagonals instead of across its cells
    std::vector<int> by_degree(ntris);
    for (uint32_t t = 0; t < ntris; t++) by_degree[t] = (int)t;
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    std::vector<uint8_t> used(ntris, 0);
    for (int t : by_degree) {
        if (used[t]) continue;
        const Cand *best = nullptr;
        for (const Cand &c : adj[t]) {
            if (used[c.t2]) continue;
            bool mutual = false;                                   // t2 must see t over the same edge (both tolerances hold)
            for (const Cand &d : adj[c.t2]) mutual = mutual || (d.t2 == t && d.e == c.e2 && d.e2 == c.e);
            if (mutual && (!best || adj[c.t2].size() < adj[best->t2].size())) best = &c;
        }
        if (!best) continue;
        used[t] = used[best->t2] = 1; rot[t] = best->e; rot[best->t2] = best->e2;
        quads.push_back({std::min(t, best->t2), std::max(t, best->t2), best->par});
    }
    // parallelograms first (their pairs take the shorter test of accel.h), each group in input order
    std::sort(quads.begin(), quads.end(), [](const QuadPair &x, const QuadPair &y) { return x.par != y.par ? x.par : x.a < y.a; });
}

// Which triangles can NEVER lie between a surface point of the scene and a point of a light — the shadow segments of next-event
// estimation (prb.py:57-59, direct.py:42-44), traced with tmin = 1e-4 and tmax = 0.9999 dist.  A triangle W qualifies when its plane
// SUPPORTS the scene: every vertex of every triangle lies on one side of it (or on it, up to `viol`, the worst violation found)
// and every vertex of every light triangle lies on that side by at least b.  A segment from p (signed distance a >= -viol, as the
// kernel computes it to within r = 4e-7 max|coordinate|) to q on a light (distance >= b) meets the plane at parameter
// t = a dist / (a - b): negative for a > 0, and at most (viol + r) D / b for a < 0, D = the scene's diagonal >= dist.  W is declared
// a non-occluder only if that bound is below HALF of tmin, so the unmasked test could not have accepted it either — the shadow
// walk skips such primitives and returns the same answer, bit for bit (tests/test_gpu_render.py, ZDR_NO_SHADOW_MASK=1).  The walls,
// floor and ceiling of a room lit from inside are the typical case: 3 of the Cornell box's 9 pairs.
static void never_occluders(const std::vector<h3> &pos, uint32_t ntris, const std::vector<uint8_t> &is_light_tri, std::vector<uint8_t> &never) {
    never.assign(ntris, 0);
    bool any_light = false;
    for (uint32_t t = 0; t < ntris; t++) any_light = any_light || is_light_tri[t];
    if (!any_light || ntris > 128) return;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, cmax = 0.0;
    for (size_t i = 0; i < 3 * (size_t)ntris; i++) {
        const double c[3] = {pos[i].x, pos[i].y, pos[i].z};
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], c[k]); hi[k] = std::max(hi[k], c[k]); cmax = std::max(cmax, fabs(c[k])); }
    }
    const double D = sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    if (!(D > 0.0) || !(D < 1e30)) return;
    for (uint32_t w = 0; w < ntris; w++) {
        const h3 *p = &pos[3 * (size_t)w];
        double e1[3] = {(double)p[1].x - p[0].x, (double)p[1].y - p[0].y, (double)p[1].z - p[0].z}, e2[3] = {(double)p[2].x - p[0].x, (double)p[2].y - p[0].y, (double)p[2].z - p[0].z};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(nn > 0.0)) continue;                                 // degenerate: never hit, but left in (it costs nothing to be careful)
        for (int k = 0; k < 3; k++) n[k] /= nn;
        double dmin = 1e300, dmax = -1e300, lmin = 1e300, lmax = -1e300;
        for (uint32_t t = 0; t < ntris; t++)
            for (int k = 0; k < 3; k++) {
                const h3 &v = pos[3 * (size_t)t + k];
                const double d = n[0] * ((double)v.x - p[0].x) + n[1] * ((double)v.y - p[0].y) + n[2] * ((double)v.z - p[0].z);
                dmin = std::min(dmin, d); dmax = std::max(dmax, d);
                if (is_light_tri[t]) { lmin = std::min(lmin, d); lmax = std::max(lmax, d); }
            }
        const double r = 4e-7 * cmax, tmin = 1e-4;
        const bool above = lmin > 0.0 && (std::max(0.0, -dmin) + r) * D <= 0.5 * tmin * lmin;     // scene on the + side, lights strictly
        const bool below = lmax < 0.0 && (std::max(0.0, dmax) + r) * D <= 0.5 * tmin * -lmax;
        never[w] = (above || below) ? 1 : 0;
    }
}

// Slot order of the brute-force walk: the two triangles of quad q in slots 2q and 2q + 1, the single triangles after them.  Within the
// parallelograms, within the other quads and within the single triangles the primitives that can occlude a shadow segment come first
// (`never`, may be null): the shadow walk then skips whole PAIRS of the others.
static uint32_t brute_slot_order(const std::vector<h3> &pos, uint32_t ntris, std::vector<int> &order, std::vector<int> &rot, uint32_t *npar = nullptr,
                                 const std::vector<uint8_t> *never = nullptr) {
    std::vector<QuadPair> quads;
    rot.assign(ntris, 0);
    if (!getenv("ZDR_NO_QUADS")) find_quads(pos, ntris, rot, quads);
    auto nv = [&](int t) { return never && (*never)[t]; };
    std::stable_sort(quads.begin(), quads.end(), [&](const QuadPair &x, const QuadPair &y) {
        if (x.par != y.par) return (bool)x.par;
        return (nv(x.a) && nv(x.b)) < (nv(y.a) && nv(y.b)); });
    std::vector<uint8_t> in_quad(ntris, 0);
    order.resize(ntris);
    uint32_t slot = 0;
    for (const QuadPair &q : quads) { order[slot++] = q.a; order[slot++] = q.b; in_quad[q.a] = in_quad[q.b] = 1; }
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t t = 0; t < ntris; t++) if (!in_quad[t] && (int)nv((int)t) == pass) order[slot++] = (int)t;
    if (npar) { *npar = 0; for (const QuadPair &q : quads) *npar += q.par ? 1u : 0u; }
    return (uint32_t)quads.size();
}

// Builds the acceleration structure over world-space triangles (pos: ntris x 3 corners).
// order[slot] = input triangle; nodes = BVH4 nodes (empty for the brute-force accel or a single leaf).
static int build_accel(const std::vector<h3> &pos, uint32_t ntris, bool use_bvh, std::vector<int> &order,
                       std::vector<float4> &nodes, uint32_t &bvh_nodes, uint32_t &bvh_depth, uint32_t &stack_entries) {
    order.resize(ntris);
    for (uint32_t t = 0; t < ntris; t++) order[t] = (int)t;
    nodes.clear(); bvh_nodes = 0; bvh_depth = 0; stack_entries = 8;
    if (!use_bvh) return ZDR_OK;
    float slo[3] = {3e38f, 3e38f, 3e38f}, shi[3] = {-3e38f, -3e38f, -3e38f};
    BvhBuilder bb;
    bb.prims.resize(ntris);
    for (uint32_t t = 0; t < ntris; t++) {
        Prim &p = bb.prims[t]; p.tri = (int)t;
        for (int k = 0; k < 3; k++) {
            float a = (&pos[3 * (size_t)t].x)[k], b = (&pos[3 * (size_t)t + 1].x)[k], c = (&pos[3 * (size_t)t + 2].x)[k];
            p.lo[k] = std::min(a, std::min(b, c)); p.hi[k] = std::max(a, std::max(b, c));
            p.c[k] = 0.5f * (p.lo[k] + p.hi[k]);
        }
        BvhBuilder::grow(slo, shi, p.lo, p.hi);
    }
    bb.nodes.reserve((size_t)ntris + 16);
    bb.build(0, (int)ntris, 0);
    for (uint32_t t = 0; t < ntris; t++) order[t] = bb.prims[t].tri;
    // conservative padding so that box culling never rejects what the triangle test accepts
    float diag = sqrtf((shi[0] - slo[0]) * (shi[0] - slo[0]) + (shi[1] - slo[1]) * (shi[1] - slo[1]) + (shi[2] - slo[2]) * (shi[2] - slo[2]));
    float pad = 4e-6f * diag + 1e-30f;
    // Collapse to a 4-wide BVH (which binary nodes survive: `collapse` below).  One node = 64 bytes (four dwordx4 loads per visit — the BVH kernels are
    // bound by the number of per-lane vector loads, so bytes per visit is what counts): the child boxes are
    // quantised to 8 bits per plane on the node's own grid,
    //   {origin.xyz, scale.x} {scale.y, scale.z, qlo.x[4], qlo.y[4]} {qlo.z[4], qhi.x[4], qhi.y[4], qhi.z[4]} {child[4]}
    // plane = origin + scale * q, rounded outwards (a quantised box always contains the padded float box);
    // child = index << 3 | count: count 0 = node index, 1..4 = first slot of a leaf, 7 = unused child.
    // Node 0 is the root; a scene that is one leaf has no nodes.
    // Which binary nodes survive as 4-wide nodes: the choice that minimises the summed surface area of the surviving nodes
    // (= the expected number of node visits of a random ray, the SAH with leaves fixed), by dynamic programming over the
    // binary tree (Ylitie, Karras, Laine 2017, section 3.1 specialised to fixed leaves): cost[n][k] = least area below n when n
    // may occupy up to k child slots of its parent.  The greedy "adopt the grandchildren of the largest child" it replaces
    // filled 3.0 of 4 slots on the 1 M triangle scene; this fills 3.4 (251 k -> 212 k nodes, 2.5 MB less to keep in L2) — at
    // the same speed: the path kernels are bound by the VALU work of the visits, and the visits saved are few.
    struct Collapse {
        const std::vector<BNode> &bn;
        std::vector<float> cost;            // 4 per binary node
        std::vector<uint8_t> split;         // 4 per binary node: 0 = as with one slot fewer, j = left gets j slots
        explicit Collapse(const std::vector<BNode> &b) : bn(b), cost(4 * b.size(), 0.0f), split(4 * b.size(), 0) {
            for (int n = (int)bn.size() - 1; n >= 0; n--) {     // children carry larger indices than their parent
                if (bn[n].count) continue;                      // leaf: constant cost, left out
                const int l = bn[n].left, r = bn[n].right;
                float best4 = 3e38f;
                for (int j = 1; j <= 3; j++) best4 = std::min(best4, cost[4 * l + j - 1] + cost[4 * r + 3 - j]);
                cost[4 * n] = BvhBuilder::area(bn[n].lo, bn[n].hi) + best4;
                for (int k = 2; k <= 4; k++) {
                    float c = cost[4 * n + k - 2]; uint8_t sp = 0;
                    for (int j = 1; j < k; j++) { float v = cost[4 * l + j - 1] + cost[4 * r + k - j - 1]; if (v < c) { c = v; sp = (uint8_t)j; } }
                    cost[4 * n + k - 1] = c; split[4 * n + k - 1] = sp;
                }
            }
        }
        void collect(int n, int k, int *kids, int &nk) const {
            while (k > 1 && !bn[n].count && split[4 * n + k - 1] == 0) k--;
            if (bn[n].count || k == 1) { kids[nk++] = n; return; }
            const int j = split[4 * n + k - 1];
            collect(bn[n].left, j, kids, nk); collect(bn[n].right, k - j, kids, nk);
        }
        void children(int n, int *kids, int &nk) const {       // the children of the wide node made from binary node n
            const int l = bn[n].left, r = bn[n].right;
            int bj = 1; float best = 3e38f;
            for (int j = 1; j <= 3; j++) { float v = cost[4 * l + j - 1] + cost[4 * r + 3 - j]; if (v < best) { best = v; bj = j; } }
            nk = 0; collect(l, bj, kids, nk); collect(r, 4 - bj, kids, nk);
        }
    } collapse(bb.nodes);
    int ninner = 0, worst_stack = 0;
    if (bb.nodes[0].count == 0) {
        // Node numbering: the top of the tree breadth-first (nodes 0 .. ZDR_BVH_BFS_NODES - 1 are the root, its children, their children ...:
        // the kernels keep the first few in LDS, accel.h), everything below depth-first (a subtree stays together in memory).
        struct Item { int bnode, id4, stack; };
        std::vector<Item> todo; todo.push_back({0, 0, 0});
        size_t head = 0;                                        // todo[head ..): pending; taken from the front while numbering breadth-first
        ninner = 1;
        nodes.assign(4, make_float4(0, 0, 0, 0));
        while (head < todo.size()) {
            Item it;
            if (ninner < ZDR_BVH_BFS_NODES) it = todo[head++];
            else { it = todo.back(); todo.pop_back(); }
            int kids[4], nk = 0;
            collapse.children(it.bnode, kids, nk);
            float lo[3][4], hi[3][4]; int child[4], cnt[4];
            for (int k = 0; k < 4; k++) {
                if (k < nk) {
                    const BNode &n = bb.nodes[kids[k]];
                    for (int ax = 0; ax < 3; ax++) { lo[ax][k] = n.lo[ax] - pad; hi[ax][k] = n.hi[ax] + pad; }
                    if (n.count) { child[k] = n.first; cnt[k] = n.count; }
                    else {
                        child[k] = ninner++; cnt[k] = 0;
                        nodes.resize(4 * (size_t)ninner, make_float4(0, 0, 0, 0));
                        todo.push_back({kids[k], child[k], it.stack + nk - 1});
                    }
                } else {
                    for (int ax = 0; ax < 3; ax++) { lo[ax][k] = 3e38f; hi[ax][k] = 3e38f; }
                    child[k] = 0; cnt[k] = -1;     // unused slot: an inverted box (below) that no finite ray enters; count 7 ends a walk that gets there anyway
                }
            }
            worst_stack = std::max(worst_stack, it.stack + nk - 1);
            // quantise on the node's grid
            float org[3], scl[3]; uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int ax = 0; ax < 3; ax++) {
                float nlo = 3e38f, nhi = -3e38f;
                for (int k = 0; k < nk; k++) { nlo = std::min(nlo, lo[ax][k]); nhi = std::max(nhi, hi[ax][k]); }
                org[ax] = nlo;
                scl[ax] = std::max(nextafterf((nhi - nlo) / 255.0f, 3e38f), 1e-30f);
                for (int k = 0; k < nk; k++) {
                    // outward rounding with a margin, then checked against the float32 plane the device reconstructs
                    double margin = 1e-3 + 8.0 * (double)(nextafterf(std::max(fabsf(nlo), fabsf(nhi)), 3e38f) - std::max(fabsf(nlo), fabsf(nhi))) / scl[ax];
                    int a = (int)std::floor(((double)lo[ax][k] - org[ax]) / scl[ax] - margin);
                    int b = (int)std::ceil(((double)hi[ax][k] - org[ax]) / scl[ax] + margin);
                    a = std::min(std::max(a, 0), 255); b = std::min(std::max(b, 0), 255);
                    while (a > 0 && fmaf((float)a, scl[ax], org[ax]) > lo[ax][k]) a--;
                    while (b < 255 && fmaf((float)b, scl[ax], org[ax]) < hi[ax][k]) b++;
                    if (fmaf((float)a, scl[ax], org[ax]) > lo[ax][k] || fmaf((float)b, scl[ax], org[ax]) < hi[ax][k])
                        return fail(ZDR_E_INVALID, "internal error: BVH box quantisation is not conservative");
                    qlo[ax] |= (uint32_t)a << (8 * k); qhi[ax] |= (uint32_t)b << (8 * k);
                }
                // unused child slots: low plane 255, high plane 0 — whatever the sign of the direction the ray would have to enter
                // after it has left (q_near A + B > q_far A + B for every finite A != 0), so the walk needs no test for them
                for (int k = nk; k < 4; k++) qlo[ax] |= 255u << (8 * k);
            }
            uint32_t cw[4];
            for (int k = 0; k < 4; k++) cw[k] = (cnt[k] < 0) ? 7u : (((uint32_t)child[k] << 3) | (uint32_t)cnt[k]);
            auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
            float4 *o = &nodes[4 * (size_t)it.id4];
            o[0] = make_float4(org[0], org[1], org[2], scl[0]);
            o[1] = make_float4(scl[1], scl[2], bits(qlo[0]), bits(qlo[1]));
            o[2] = make_float4(bits(qlo[2]), bits(qhi[0]), bits(qhi[1]), bits(qhi[2]));
            o[3] = make_float4(bits(cw[0]), bits(cw[1]), bits(cw[2]), bits(cw[3]));
        }
    }
    if (worst_stack + 5 > ZDR_BVH_STACK) { return fail(ZDR_E_UNSUPPORTED, "BVH needs a deeper traversal stack than ZDR_BVH_STACK"); }   // + 4 slots of slack for the unconditional stores
    {   // structural self-check before anything reaches the GPU: the nodes form a tree rooted at 0,
        // every node is referenced once, the leaves tile [0, ntris) exactly, counts fit the 3-bit field
        std::vector<uint8_t> seen_node(ninner, 0), seen_tri(ntris, 0);
        bool ok = true;
        if (ninner) seen_node[0] = 1;
        for (int i = 0; i < ninner && ok; i++) {
            uint32_t cw[4]; memcpy(cw, &nodes[4 * (size_t)i + 3], 16);
            int ch[4], ct[4];
            for (int k = 0; k < 4; k++) { ch[k] = (int)(cw[k] >> 3); ct[k] = (int)(cw[k] & 7u); if (ct[k] == 7) ct[k] = -1; }
            for (int k = 0; k < 4 && ok; k++) {
                if (ct[k] < 0) continue;
                if (ct[k] == 0) { ok = ch[k] > i && ch[k] < ninner && !seen_node[ch[k]]; if (ok) seen_node[ch[k]] = 1; }
                else {
                    ok = ct[k] <= ZDR_BVH_LEAF && ch[k] >= 0 && (uint32_t)(ch[k] + ct[k]) <= ntris;
                    for (int q = 0; q < ct[k] && ok; q++) { ok = !seen_tri[ch[k] + q]; seen_tri[ch[k] + q] = 1; }
                }
            }
        }
        for (int i = 0; i < ninner && ok; i++) ok = seen_node[i] != 0;
        if (ninner) for (uint32_t t = 0; t < ntris && ok; t++) ok = seen_tri[t] != 0;
        if (!ok) { return fail(ZDR_E_INVALID, "internal error: BVH failed its structural self-check"); }
    }
    // Child words as the device reads them (accel.h, fetch): the BYTE OFFSET of what the child names inside the one allocation that
    // holds the nodes (64 bytes each, first) and the plane records (48 bytes per slot, behind them), | the count in the low bits
    // (offsets are multiples of 16).  The structural check above read the index form.
    if ((size_t)ninner * 64 + (size_t)ntris * 48 >= (1ull << 32)) return fail(ZDR_E_UNSUPPORTED, "BVH nodes and triangle records exceed 4 GiB");
    for (int i = 0; i < ninner; i++) {
        uint32_t cw[4]; memcpy(cw, &nodes[4 * (size_t)i + 3], 16);
        for (int k = 0; k < 4; k++) {
            const uint32_t idx = cw[k] >> 3, ct = cw[k] & 7u;
            cw[k] = (ct == 7u) ? 7u : ((ct == 0u ? idx * 64u : (uint32_t)ninner * 64u + idx * 48u) | ct);
        }
        memcpy(&nodes[4 * (size_t)i + 3], cw, 16);
    }
    bvh_nodes = (uint32_t)ninner; bvh_depth = (uint32_t)bb.max_depth;
    stack_entries = (uint32_t)((worst_stack + 5 + 3) & ~3);   // deepest pending set + the four unconditionally stored slots
    return ZDR_OK;
}

// ----------------------------------------------------------------------------------- scene
struct zdr_scene {
    int device = 0;
    int accel_is_bvh = 0;
    uint32_t ntris = 0, nverts = 0, ninst = 0;
    uint32_t bvh_nodes = 0, bvh_depth = 0, stack_entries = 8;
    std::vector<int32_t> inst_tri_begin;
    std::vector<float> emission;
    float4 *d_isect = nullptr, *d_pairs = nullptr, *d_shade = nullptr, *d_nodes = nullptr;
    bool isect_in_nodes = false;            // BVH: d_isect points into the d_nodes allocation (freed once)
    std::vector<int> slot_tri;              // brute force: slot -> input triangle
    unsigned long long shadow_pairs = ~0ull;   // brute force: the pairs of the pair walk a shadow segment can meet (never_occluders; rebuilt when the lights change)
    uint32_t nquads = 0, nquads2 = 0, npar = 0;   // brute force: primitives of the pair walk, how many of them are merged quads (find_quads), how many of those parallelograms
    float4 *d_ppairs = nullptr;
    float *d_emission = nullptr;
    // flat light table (scene.h): one 80-byte entry per triangle of every emitting instance + {first entry, count} per light
    std::vector<float4> tri_geo;            // host copy, 4 float4 per INPUT triangle: p0, p1, p2, {ng, area} (lights may change)
    float4 *d_light_tris = nullptr; size_t light_tris_cap = 0;
    int32_t *d_light_range = nullptr;       // 2 ints per instance slot
    float4 *d_emission4 = nullptr;          // ninst x {e.rgb, 0}
    int32_t *d_light_insts = nullptr, *d_inst_tri_begin = nullptr, *d_slot_of_tri = nullptr;
    uint32_t *d_pmj = nullptr; uint16_t *d_bn = nullptr; SamplerTables tab{};
    float4 *d_env_tex = nullptr; float *d_alias_prob = nullptr, *d_env_pdf = nullptr; int32_t *d_alias_idx = nullptr;
    float4 *d_partial = nullptr; size_t partial_bytes = 0;
    unsigned long long *d_tile_masks = nullptr; size_t tile_mask_bytes = 0;   // camera-ray candidate pairs per tile (k_tile_masks)
    float tile_mask_key[24]; bool tile_mask_key_set = false;                  // camera + tile grid the masks in the buffer were built for (all tiles of the rectangle, whatever the shard)
    unsigned int *d_work_counters = nullptr;
    float4 *d_ring = nullptr; size_t ring_bytes = 0;     // primary rings of the path kernels (integrators.h)
    float *d_cells = nullptr; size_t cells_bytes = 0;       // backward staging cells, (tex_h+1) x (tex_w+1) x 16 floats
    unsigned long long *d_counters = nullptr;
    unsigned int *d_error = nullptr;                        // device error word (scene.h, ZDR_DEVERR_*), sticky until read
    uint64_t device_bytes = 0;
    // A render call recorded while its stream was CAPTURING (hipStreamBeginCapture; torch.cuda.graph) bakes this handle's workspace
    // pointers into a graph that may be replayed at any later time.  From then on the handle never frees a buffer a kernel can read or
    // write — one that has to grow is retired (kept until zdr_scene_destroy) and replaced — and never trusts the tile masks in the
    // buffer, which a replay rebuilds for ITS view behind the host's back (render_common).
    bool captured = false;
    std::vector<void *> retired;
    DScene ds{};
};

static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

// frees a buffer of the handle that no kernel in flight reads — unless a captured graph may still name it
static void release_buffer(zdr_scene *s, void *p) {
    if (!p) return;
    if (s->captured) s->retired.push_back(p); else (void)hipFree(p);
}

template <class T>
static hipError_t upload(T **dst, const void *src, size_t bytes, uint64_t *acc) {
    hipError_t e = hipMalloc((void **)dst, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return e;
    if (acc) *acc += bytes;
    return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
}

static void light_list(const std::vector<float> &em, uint32_t ninst, std::vector<int32_t> &out, int &count) {
    out.assign(ninst, 0); count = 0;     // render.py:89-90,118-121,146-148
    for (uint32_t i = 0; i < ninst; i++)
        if (em[3 * i] > 0.0f || em[3 * i + 1] > 0.0f || em[3 * i + 2] > 0.0f) out[count++] = (int32_t)i;
}

// Light table for sample_light (scene.h): the reference walks light -> instance -> triangle range -> triangle ->
// emission through four dependent lookups (light.py:33-48); on the GPU that is four memory round trips and eleven
// per-lane loads per path vertex.  Flattened here to {first entry, T} per light and one 80-byte entry per light
// triangle {p0} {p1} {p2} {ng, area} {emission, 0}, the same floats the shade records hold.
// the pairs of the brute-force walk that hold at least one primitive a shadow segment can meet, for the current lights
static unsigned long long shadow_pair_mask(const zdr_scene *s, const std::vector<int32_t> &lights, int count) {
    if (s->accel_is_bvh || s->slot_tri.empty() || s->ntris > 128 || getenv("ZDR_NO_SHADOW_MASK")) return ~0ull;
    std::vector<h3> pos(3 * (size_t)s->ntris);
    for (uint32_t t = 0; t < s->ntris; t++) for (int k = 0; k < 3; k++) { const float4 &g = s->tri_geo[4 * (size_t)t + k]; pos[3 * (size_t)t + k] = H3(g.x, g.y, g.z); }
    std::vector<uint8_t> is_light(s->ntris, 0), never;
    for (int l = 0; l < count; l++) for (int t = s->inst_tri_begin[lights[l]]; t < s->inst_tri_begin[lights[l] + 1]; t++) is_light[t] = 1;
    never_occluders(pos, s->ntris, is_light, never);
    unsigned long long m = 0ull;
    for (uint32_t q = 0; q < s->nquads; q++) {                  // primitive q: slots 2q, 2q + 1 (a quad) or slot q + nquads2 (a single triangle)
        const bool quad = q < s->nquads2;
        const uint32_t a = quad ? 2 * q : q + s->nquads2;
        const bool skip = never[s->slot_tri[a]] && (!quad || never[s->slot_tri[a + 1]]);
        if (!skip) m |= 1ull << (q / 2);
    }
    return m;
}

static int upload_light_table(zdr_scene *s, const std::vector<int32_t> &lights, int count, hipStream_t st) {
    s->shadow_pairs = shadow_pair_mask(s, lights, count);
    std::vector<float4> tab; std::vector<int32_t> range(2 * (size_t)s->ninst, 0);
    for (int l = 0; l < count; l++) {
        const int inst = lights[l], b = s->inst_tri_begin[inst], T = s->inst_tri_begin[inst + 1] - b;
        range[2 * l] = (int32_t)(tab.size() / 5); range[2 * l + 1] = T;
        for (int t = b; t < b + T; t++) {
            for (int k = 0; k < 4; k++) tab.push_back(s->tri_geo[4 * (size_t)t + k]);
            tab.push_back(make_float4(s->emission[3 * (size_t)inst], s->emission[3 * (size_t)inst + 1], s->emission[3 * (size_t)inst + 2], 0.0f));
        }
    }
    if (tab.empty()) tab.push_back(make_float4(0, 0, 0, 0));
    if (tab.size() > s->light_tris_cap) {
        HIPCHK(hipStreamSynchronize(st));
        release_buffer(s, s->d_light_tris); s->d_light_tris = nullptr; s->light_tris_cap = 0;
        HIPCHK(hipMalloc((void **)&s->d_light_tris, tab.size() * sizeof(float4)));
        s->light_tris_cap = tab.size();
    }
    if (!s->d_light_range) HIPCHK(hipMalloc((void **)&s->d_light_range, range.size() * sizeof(int32_t)));
    if (!s->d_emission4) HIPCHK(hipMalloc((void **)&s->d_emission4, (size_t)s->ninst * sizeof(float4)));
    std::vector<float4> e4(s->ninst);
    for (uint32_t i = 0; i < s->ninst; i++) e4[i] = make_float4(s->emission[3 * (size_t)i], s->emission[3 * (size_t)i + 1], s->emission[3 * (size_t)i + 2], 0.0f);
    HIPCHK(hipMemcpyAsync(s->d_light_tris, tab.data(), tab.size() * sizeof(float4), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_light_range, range.data(), range.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_emission4, e4.data(), e4.size() * sizeof(float4), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));       // the host vectors go out of scope
    s->ds.light_tris = s->d_light_tris; s->ds.light_range = s->d_light_range; s->ds.emission4 = s->d_emission4;
    s->ds.light0_T = count > 0 ? range[1] : 0;
    return ZDR_OK;
}

extern "C" int zdr_scene_create(const float *verts8, uint32_t nverts, const int32_t *tris, uint32_t ntris,
                                const int32_t *inst_tri_begin, const float *inst_xform, const float *inst_emission,
                                uint32_t ninst, int device, int accel, zdr_scene **out) {
    if (!verts8 || !tris || !inst_tri_begin || !inst_emission || !out) return fail(ZDR_E_INVALID, "null argument");
    if (ntris == 0 || nverts == 0 || ninst == 0) return fail(ZDR_E_INVALID, "empty scene");
    if (ninst > 10000) return fail(ZDR_E_INVALID, "exceeding maximum number of mesh instances");   // render.py:114-115
    if (ntris >= (1u << 28)) return fail(ZDR_E_UNSUPPORTED, "too many triangles");
    if (inst_tri_begin[0] != 0 || (uint32_t)inst_tri_begin[ninst] != ntris) return fail(ZDR_E_INVALID, "inst_tri_begin must span [0, ntris]");
    for (uint32_t i = 0; i < ninst; i++) if (inst_tri_begin[i + 1] < inst_tri_begin[i]) return fail(ZDR_E_INVALID, "inst_tri_begin must be non-decreasing");
    for (size_t i = 0; i < (size_t)ntris * 3; i++) if (tris[i] < 0 || (uint32_t)tris[i] >= nverts) return fail(ZDR_E_INVALID, "triangle index out of range");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ZDR_E_INVALID, "no such HIP device");
    HIPCHK(hipSetDevice(device));

    zdr_scene *s = new zdr_scene();
    s->device = device; s->ntris = ntris; s->nverts = nverts; s->ninst = ninst;
    s->inst_tri_begin.assign(inst_tri_begin, inst_tri_begin + ninst + 1);
    s->emission.assign(inst_emission, inst_emission + 3 * (size_t)ninst);

    // world-space per-triangle records in input order
    struct TriRec { h3 p[3], n[3]; float uv[3][2]; h3 ng; float area; int inst, prim; };
    std::vector<TriRec> rec(ntris);
    static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float slo[3] = {3e38f, 3e38f, 3e38f}, shi[3] = {-3e38f, -3e38f, -3e38f};
    for (uint32_t i = 0; i < ninst; i++) {
        const float *m = inst_xform ? inst_xform + 16 * (size_t)i : ident;
        float nm[9]; normal_matrix(m, nm);
        for (int t = inst_tri_begin[i]; t < inst_tri_begin[i + 1]; t++) {
            TriRec &r = rec[t];
            for (int k = 0; k < 3; k++) {
                const float *v = verts8 + 8 * (size_t)tris[3 * (size_t)t + k];
                r.p[k] = xform_point(m, H3(v[0], v[1], v[2]));
                r.uv[k][0] = v[3]; r.uv[k][1] = v[4];
                r.n[k] = H3(nm[0] * v[5] + nm[1] * v[6] + nm[2] * v[7], nm[3] * v[5] + nm[4] * v[6] + nm[5] * v[7], nm[6] * v[5] + nm[7] * v[6] + nm[8] * v[7]);
                const float pc[3] = {r.p[k].x, r.p[k].y, r.p[k].z};
                BvhBuilder::grow(slo, shi, pc, pc);
            }
            h3 c = hcross(hsub(r.p[1], r.p[0]), hsub(r.p[2], r.p[0]));
            r.ng = hnormalize(c);                       // interaction.py:29, light.py:68
            r.area = sqrtf(hdot(c, c)) / 2.0f;          // light.py:72
            r.inst = (int)i; r.prim = t - inst_tri_begin[i];
        }
    }

    // acceleration structure: slot order + nodes
    std::vector<int> order;
    std::vector<float4> nodes;
    bool use_bvh = (accel == ZDR_ACCEL_BVH) || (accel == ZDR_ACCEL_AUTO && ntris > 64);
    {
        std::vector<h3> pos(3 * (size_t)ntris);
        for (uint32_t t = 0; t < ntris; t++) for (int k = 0; k < 3; k++) pos[3 * (size_t)t + k] = rec[t].p[k];
        int rc = build_accel(pos, ntris, use_bvh, order, nodes, s->bvh_nodes, s->bvh_depth, s->stack_entries);
        if (rc) { delete s; return rc; }
    }
    s->accel_is_bvh = use_bvh ? 1 : 0;
    // brute force: merge coplanar triangle pairs into quads; slots 2q, 2q + 1 = quad q, the single triangles follow
    std::vector<int> rot(ntris, 0);
    if (!use_bvh) {
        std::vector<h3> pos(3 * (size_t)ntris);
        for (uint32_t t = 0; t < ntris; t++) for (int k = 0; k < 3; k++) pos[3 * (size_t)t + k] = rec[t].p[k];
        std::vector<uint8_t> is_light(ntris, 0), never;       // (the lights of the moment only ORDER the primitives; the mask itself follows update_lights)
        for (uint32_t t = 0; t < ntris; t++) { const float *e = &s->emission[3 * (size_t)rec[t].inst]; is_light[t] = (e[0] > 0.0f || e[1] > 0.0f || e[2] > 0.0f) ? 1 : 0; }
        never_occluders(pos, ntris, is_light, never);
        s->nquads2 = brute_slot_order(pos, ntris, order, rot, &s->npar, &never); s->nquads = ntris - s->nquads2;
        s->slot_tri = order;
    }

    std::vector<float4> isect(3 * (size_t)ntris + 3, make_float4(0, 0, 0, 0)), shade(8 * (size_t)ntris);   // + one record: the BVH walk fetches four float4 behind a leaf's first triangle
    std::vector<int32_t> slot_of_tri(ntris);
    std::vector<float4> wedge(ntris);       // third edge function of every slot (single triangles of the brute-force walk)
    for (uint32_t slot = 0; slot < ntris; slot++) {
        TriRec r = rec[order[slot]];
        slot_of_tri[order[slot]] = (int32_t)slot;
        const int ro = rot[order[slot]];     // the slot's corner 0 is input corner `ro` (cyclic: same triangle, same orientation)
        if (ro) {
            const TriRec in = r;
            for (int k = 0; k < 3; k++) { r.p[k] = in.p[(k + ro) % 3]; r.n[k] = in.n[(k + ro) % 3]; r.uv[k][0] = in.uv[(k + ro) % 3][0]; r.uv[k][1] = in.uv[(k + ro) % 3][1]; }
        }
        plane_record(r.p, &isect[3 * (size_t)slot], &wedge[slot]);
        float4 *q = &shade[8 * (size_t)slot];
        q[0] = make_float4(r.p[0].x, r.p[0].y, r.p[0].z, r.uv[0][0]);
        q[1] = make_float4(r.p[1].x, r.p[1].y, r.p[1].z, r.uv[0][1]);
        q[2] = make_float4(r.p[2].x, r.p[2].y, r.p[2].z, r.uv[1][0]);
        q[3] = make_float4(r.n[0].x, r.n[0].y, r.n[0].z, r.uv[1][1]);
        q[4] = make_float4(r.n[1].x, r.n[1].y, r.n[1].z, r.uv[2][0]);
        q[5] = make_float4(r.n[2].x, r.n[2].y, r.n[2].z, r.uv[2][1]);
        float fi, fp; memcpy(&fi, &r.inst, 4); memcpy(&fp, &r.prim, 4);
        q[6] = make_float4(r.ng.x, r.ng.y, r.ng.z, fi);
        float fr; memcpy(&fr, &ro, 4);
        q[7] = make_float4(r.area, fp, fr, 0.0f);          // .z: rotation of the slot's corners against the input triangle's
    }
    std::vector<int32_t> lights; int light_count = 0;
    light_list(s->emission, ninst, lights, light_count);
    s->tri_geo.resize(4 * (size_t)ntris);
    for (uint32_t t = 0; t < ntris; t++) {
        const TriRec &r = rec[t];
        s->tri_geo[4 * (size_t)t] = make_float4(r.p[0].x, r.p[0].y, r.p[0].z, 0.0f);
        s->tri_geo[4 * (size_t)t + 1] = make_float4(r.p[1].x, r.p[1].y, r.p[1].z, 0.0f);
        s->tri_geo[4 * (size_t)t + 2] = make_float4(r.p[2].x, r.p[2].y, r.p[2].z, 0.0f);
        s->tri_geo[4 * (size_t)t + 3] = make_float4(r.ng.x, r.ng.y, r.ng.z, r.area);
    }

    hipError_t e = hipSuccess;
    auto up = [&](auto **dst, const void *src, size_t bytes) { if (e == hipSuccess) e = upload(dst, src, bytes, &s->device_bytes); };
    // BVH: nodes and plane records share ONE allocation, nodes first — the walk then addresses whatever a lane stands on as
    // base + a 32-bit byte offset (accel.h, fetch): no 64-bit address arithmetic, no branch between "node" and "triangle" pointers.
    const bool one_block = use_bvh;
    if (one_block) {
        if ((nodes.size() + isect.size()) * sizeof(float4) >= (1ull << 32)) { zdr_scene_destroy(s); return fail(ZDR_E_UNSUPPORTED, "BVH nodes and triangle records exceed 4 GiB"); }
        std::vector<float4> both(nodes); both.insert(both.end(), isect.begin(), isect.end());
        up(&s->d_nodes, both.data(), both.size() * sizeof(float4));
        s->d_isect = s->d_nodes ? s->d_nodes + nodes.size() : nullptr;
        s->isect_in_nodes = true;
    } else up(&s->d_isect, isect.data(), isect.size() * sizeof(float4));
    if (!use_bvh) {
        // brute-force loops test two PRIMITIVES per trip with packed fp32 math, a primitive being a quad (slots 2q, 2q + 1)
        // or a single triangle: plane N of the (first) triangle and four edge functions — u and v of both triangles of a quad,
        // or u, v, w, w of a single triangle — five float4, interleaved for the two halves of a float2
        size_t npairs = ((size_t)s->nquads + 1) / 2;
        std::vector<float> pr(40 * npairs, 0.0f);
        for (uint32_t q = 0; q < s->nquads; q++) {
            const uint32_t a = q < s->nquads2 ? 2 * q : q + s->nquads2;
            float4 src[5] = {isect[3 * (size_t)a], isect[3 * (size_t)a + 1], isect[3 * (size_t)a + 2], wedge[a], wedge[a]};
            if (q < s->nquads2) { src[3] = isect[3 * (size_t)(a + 1) + 1]; src[4] = isect[3 * (size_t)(a + 1) + 2]; }
            float *dst = &pr[40 * (size_t)(q / 2) + (q & 1)];
            for (int k = 0; k < 20; k++) dst[2 * k] = ((const float *)src)[k];
        }
        up(&s->d_pairs, pr.data(), pr.size() * sizeof(float));
        // pairs of two PARALLELOGRAMS (they come first): the outer edges of the second triangle are 1 - u and 1 - v of the first,
        // so the record is the plane and two edge functions — six float4
        const size_t nppairs = s->npar / 2;
        std::vector<float> ppr(24 * std::max<size_t>(nppairs, 1), 0.0f);
        for (uint32_t q = 0; q < 2 * nppairs; q++) {
            const float *src = (const float *)&isect[3 * (size_t)(2 * q)];
            float *dst = &ppr[24 * (size_t)(q / 2) + (q & 1)];
            for (int k = 0; k < 12; k++) dst[2 * k] = src[k];
        }
        up(&s->d_ppairs, ppr.data(), ppr.size() * sizeof(float));
        s->ds.nppairs = getenv("ZDR_NO_PARALLELOGRAMS") ? 0 : (int32_t)nppairs;
    }
    up(&s->d_shade, shade.data(), shade.size() * sizeof(float4));
    if (!one_block) up(&s->d_nodes, nodes.data(), nodes.size() * sizeof(float4));
    up(&s->d_emission, s->emission.data(), s->emission.size() * sizeof(float));
    up(&s->d_light_insts, lights.data(), lights.size() * sizeof(int32_t));
    up(&s->d_inst_tri_begin, s->inst_tri_begin.data(), s->inst_tri_begin.size() * sizeof(int32_t));
    up(&s->d_slot_of_tri, slot_of_tri.data(), slot_of_tri.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_counters, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_error, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(s->d_error, 0, sizeof(unsigned int));
    if (e != hipSuccess) { std::string m = hipGetErrorString(e); zdr_scene_destroy(s); return fail(ZDR_E_HIP, "scene upload: " + m); }
    s->ds.isect = s->d_isect; s->ds.pairs = s->d_pairs; s->ds.ppairs = s->d_ppairs; s->ds.shade = s->d_shade; s->ds.nodes = s->d_nodes; s->ds.emission = s->d_emission;
    s->ds.light_insts = s->d_light_insts; s->ds.inst_tri_begin = s->d_inst_tri_begin; s->ds.slot_of_tri = s->d_slot_of_tri;
    s->ds.error_word = s->d_error;
    if (const char *e = getenv("ZDR_DEBUG_BVH_BUDGET")) s->ds.debug_bvh_budget = atoi(e);
    s->ds.nquads = (int32_t)s->nquads; s->ds.nquads2 = (int32_t)s->nquads2;
    s->ds.ntris = (int32_t)ntris; s->ds.ninst = (int32_t)ninst; s->ds.light_count = light_count; s->ds.nnodes = (int32_t)s->bvh_nodes; s->ds.stack_entries = (int32_t)s->stack_entries;
    s->ds.walk_base = (const char *)(one_block ? s->d_nodes : s->d_isect); s->ds.isect_off = one_block ? (uint32_t)(nodes.size() * sizeof(float4)) : 0u;   // accel.h, fetch
    { int rc = upload_light_table(s, lights, light_count, nullptr); if (rc) { zdr_scene_destroy(s); return rc; } }
    *out = s;
    return ZDR_OK;
}

// Host-only view of the acceleration structure (no GPU needed): lets the CPU test-suite run an
// emulation of the device traversal on exactly the data the kernels would see.
extern "C" int zdr_debug_build_accel(const float *tri_xyz, uint32_t ntris, int accel, float *nodes_out, uint32_t nodes_cap,
                                     uint32_t *nnodes, uint32_t *stack_entries, int32_t *order_out, float *isect_out) {
    if (!tri_xyz || !ntris || !nnodes || !order_out || !isect_out) return fail(ZDR_E_INVALID, "null argument");
    std::vector<h3> pos(3 * (size_t)ntris);
    for (size_t i = 0; i < 3 * (size_t)ntris; i++) pos[i] = H3(tri_xyz[3 * i], tri_xyz[3 * i + 1], tri_xyz[3 * i + 2]);
    std::vector<int> order; std::vector<float4> nodes; uint32_t nn = 0, depth = 0, se = 8;
    bool use_bvh = (accel == ZDR_ACCEL_BVH) || (accel == ZDR_ACCEL_AUTO && ntris > 64);
    int rc = build_accel(pos, ntris, use_bvh, order, nodes, nn, depth, se); if (rc) return rc;
    std::vector<int> rot(ntris, 0);
    uint32_t npar = 0;
    if (!use_bvh) { nn = brute_slot_order(pos, ntris, order, rot, &npar); se = npar; }   // brute force: *nnodes = merged quads (slots 2q, 2q + 1), *stack_entries = how many of them are parallelograms (they come first); no nodes
    *nnodes = nn;
    if (stack_entries) *stack_entries = se;
    if (use_bvh) {
        if (nn > nodes_cap) return fail(ZDR_E_NOMEM, "nodes_out too small");
        if (nn) memcpy(nodes_out, nodes.data(), (size_t)nn * 4 * sizeof(float4));
    }
    for (uint32_t slot = 0; slot < ntris; slot++) {
        order_out[slot] = order[slot];
        const int ro = rot[order[slot]];
        h3 c[3]; for (int k = 0; k < 3; k++) c[k] = pos[3 * (size_t)order[slot] + (k + ro) % 3];
        float4 q[3]; plane_record(c, q);
        memcpy(isect_out + 12 * (size_t)slot, q, sizeof q);
    }
    return ZDR_OK;
}

// Host-only: the classification behind the shadow walk's pair mask (never_occluders above), for the CPU test-suite.
extern "C" int zdr_debug_never_occluders(const float *tri_xyz, uint32_t ntris, const uint8_t *is_light_tri, uint8_t *never_out) {
    if (!tri_xyz || !ntris || !is_light_tri || !never_out) return fail(ZDR_E_INVALID, "null argument");
    std::vector<h3> pos(3 * (size_t)ntris);
    for (size_t i = 0; i < 3 * (size_t)ntris; i++) pos[i] = H3(tri_xyz[3 * i], tri_xyz[3 * i + 1], tri_xyz[3 * i + 2]);
    std::vector<uint8_t> light(is_light_tri, is_light_tri + ntris), never;
    never_occluders(pos, ntris, light, never);
    memcpy(never_out, never.data(), ntris);
    return ZDR_OK;
}

extern "C" int zdr_scene_destroy(zdr_scene *s) {
    if (!s) return ZDR_OK;
    (void)hipSetDevice(s->device);
    if (!s->isect_in_nodes) (void)hipFree(s->d_isect); (void)hipFree(s->d_pairs); (void)hipFree(s->d_ppairs); (void)hipFree(s->d_shade); (void)hipFree(s->d_nodes); (void)hipFree(s->d_emission); (void)hipFree(s->d_light_insts); (void)hipFree(s->d_light_tris); (void)hipFree(s->d_light_range); (void)hipFree(s->d_emission4);
    (void)hipFree(s->d_inst_tri_begin); (void)hipFree(s->d_slot_of_tri); (void)hipFree(s->d_pmj); (void)hipFree(s->d_bn); (void)hipFree(s->d_env_tex); (void)hipFree(s->d_alias_prob); (void)hipFree(s->d_alias_idx); (void)hipFree(s->d_env_pdf); (void)hipFree(s->d_partial); (void)hipFree(s->d_ring); (void)hipFree(s->d_work_counters); (void)hipFree(s->d_tile_masks); (void)hipFree(s->d_cells); (void)hipFree(s->d_counters); (void)hipFree(s->d_error);
    for (void *p : s->retired) (void)hipFree(p);
    delete s;
    return ZDR_OK;
}

extern "C" int zdr_scene_info(const zdr_scene *s, zdr_scene_info_t *info) {
    if (!s || !info) return fail(ZDR_E_INVALID, "null argument");
    info->ntris = s->ntris; info->nverts = s->nverts; info->ninst = s->ninst; info->light_count = (uint32_t)s->ds.light_count;
    info->accel = s->accel_is_bvh ? ZDR_ACCEL_BVH : ZDR_ACCEL_BRUTE;
    info->bvh_nodes = s->bvh_nodes; info->bvh_max_depth = s->bvh_depth; info->bvh_stack_entries = s->stack_entries; info->device = s->device; info->device_bytes = s->device_bytes;
    return ZDR_OK;
}

extern "C" int zdr_scene_set_emissions(zdr_scene *s, const float *inst_emission, void *stream) {
    if (!s || !inst_emission) return fail(ZDR_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(s->device));
    s->emission.assign(inst_emission, inst_emission + 3 * (size_t)s->ninst);
    std::vector<int32_t> lights; int count = 0;
    light_list(s->emission, s->ninst, lights, count);
    hipStream_t st = (hipStream_t)stream;
    // pageable sources: these copies return once the data is staged, so the vectors may go out of scope
    HIPCHK(hipMemcpyAsync(s->d_emission, s->emission.data(), s->emission.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_light_insts, lights.data(), lights.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    s->ds.light_count = count;
    return upload_light_table(s, lights, count, st);
}

extern "C" int zdr_scene_set_envmap(zdr_scene *s, const float *tex, uint32_t tex_h, uint32_t tex_w, const float *alias_prob,
                                    const int32_t *alias_idx, const float *pdf, uint32_t map_w, uint32_t map_h) {
    if (!s) return fail(ZDR_E_INVALID, "null scene");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());     // nothing in flight may still read the old tables
    release_buffer(s, s->d_env_tex); release_buffer(s, s->d_alias_prob); release_buffer(s, s->d_alias_idx); release_buffer(s, s->d_env_pdf);
    s->d_env_tex = nullptr; s->d_alias_prob = nullptr; s->d_alias_idx = nullptr; s->d_env_pdf = nullptr;
    s->ds.env_count = 0; s->ds.env_tex = nullptr; s->ds.alias_prob = nullptr; s->ds.alias_idx = nullptr; s->ds.env_pdf = nullptr;
    if (!tex) return ZDR_OK;
    if (!alias_prob || !alias_idx || !pdf || !tex_h || !tex_w || !map_w || !map_h) return fail(ZDR_E_INVALID, "bad envmap arguments");
    size_t n_alias = (size_t)map_h + (size_t)map_h * map_w;
    for (size_t i = 0; i < n_alias; i++) {
        size_t lim = i < map_h ? map_h : map_w;
        if (alias_idx[i] < 0 || (size_t)alias_idx[i] >= lim) return fail(ZDR_E_INVALID, "alias index out of range");
    }
    HIPCHK(upload(&s->d_env_tex, tex, (size_t)tex_h * tex_w * sizeof(float4), &s->device_bytes));
    HIPCHK(upload(&s->d_alias_prob, alias_prob, n_alias * sizeof(float), &s->device_bytes));
    HIPCHK(upload(&s->d_alias_idx, alias_idx, n_alias * sizeof(int32_t), &s->device_bytes));
    HIPCHK(upload(&s->d_env_pdf, pdf, (size_t)map_w * map_h * sizeof(float), &s->device_bytes));
    s->ds.env_tex = s->d_env_tex; s->ds.alias_prob = s->d_alias_prob; s->ds.alias_idx = s->d_alias_idx; s->ds.env_pdf = s->d_env_pdf;
    s->ds.env_h = (int32_t)tex_h; s->ds.env_w = (int32_t)tex_w; s->ds.map_w = (int32_t)map_w; s->ds.map_h = (int32_t)map_h;
    s->ds.env_count = 1;
    return ZDR_OK;
}

extern "C" int zdr_scene_set_pmj02bn_tables(zdr_scene *s, const uint32_t *pmj, uint32_t nsets, uint32_t nsamples,
                                            const uint16_t *bn, uint32_t ntex, uint32_t bnres) {
    if (!s || !pmj || !bn || !nsets || !nsamples || !ntex || !bnres) return fail(ZDR_E_INVALID, "bad table arguments");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());     // nothing in flight may still read the old tables
    // only the sampler tables belong to this setter (the environment buffers are zdr_scene_set_envmap's)
    release_buffer(s, s->d_pmj); release_buffer(s, s->d_bn); s->d_pmj = nullptr; s->d_bn = nullptr;
    memset(&s->tab, 0, sizeof s->tab);
    HIPCHK(upload(&s->d_pmj, pmj, (size_t)nsets * nsamples * 2 * sizeof(uint32_t), &s->device_bytes));
    HIPCHK(upload(&s->d_bn, bn, (size_t)ntex * bnres * bnres * sizeof(uint16_t), &s->device_bytes));
    s->tab.pmj = s->d_pmj; s->tab.bn = s->d_bn; s->tab.nsets = nsets; s->tab.nsamples = nsamples; s->tab.ntex = ntex; s->tab.bnres = bnres;
    return ZDR_OK;
}

// ------------------------------------------------------------------------- launch set-up
static uint32_t smear(uint32_t w) { w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16; return w; }
static bool is_pow2(uint32_t x) { return x && !(x & (x - 1)); }

static int make_sampler_cfg(const zdr_scene *s, int32_t sampler, uint32_t seed, uint32_t spp, SamplerCfg &C) {
    if (spp == 0) return fail(ZDR_E_INVALID, "spp must be positive");
    memset(&C, 0, sizeof C);
    C.kind = sampler; C.seed = seed; C.spp = spp; C.w = smear(spp - 1);
    // CMJ strata grid: res x res for perfect squares (corrmj.py:67), else res_x * res_y >= spp
    uint32_t res = (uint32_t)(int)sqrtf((float)spp + 0.4f);
    if (res < 1) res = 1;
    if (res * res == spp) { C.res_x = C.res_y = res; }
    else if (is_pow2(spp)) { uint32_t lg = 0; while ((1u << lg) < spp) lg++; C.res_x = 1u << ((lg + 1) / 2); C.res_y = spp / C.res_x; }
    else { uint32_t m = res; while (m * m < spp) m++; C.res_x = m; C.res_y = (spp + m - 1) / m; }
    C.resw_x = smear(C.res_x - 1); C.resw_y = smear(C.res_y - 1);
    C.spp_pow2 = is_pow2(spp); C.res_pow2 = is_pow2(C.res_x) && is_pow2(C.res_y);
    C.inv_spp = 1.0f / (float)spp; C.inv_res_x = 1.0f / (float)C.res_x; C.inv_res_y = 1.0f / (float)C.res_y;
    C.res_x_shift = 0; while ((1u << C.res_x_shift) < C.res_x) C.res_x_shift++;
    if (sampler == ZDR_SAMPLER_PMJ02BN) {
        if (!s->d_pmj || !s->d_bn) return fail(ZDR_E_UNSUPPORTED, "PMJ02bn sampler needs tables: call zdr_scene_set_pmj02bn_tables (the reference's own tables are absent)");
        if (spp > s->tab.nsamples) return fail(ZDR_E_INVALID, "spp exceeds the PMJ02bn table length");
        C.tab = s->tab;
    } else if (sampler != ZDR_SAMPLER_CMJ) return fail(ZDR_E_INVALID, "unknown sampler");
    return ZDR_OK;
}

// The struct grows with the library: a caller built against another header says so in struct_size instead of having the
// library read past (or short of) what it passed.  Checked before anything else looks at the parameters.
static int check_params_abi(const zdr_render_params *p) {
    if (p->struct_size != sizeof(zdr_render_params))
        return fail(ZDR_E_INVALID, "zdr_render_params.struct_size is " + std::to_string(p->struct_size) + ", this library (ABI " + std::to_string(ZDR_ABI_VERSION) +
                                   ") expects " + std::to_string(sizeof(zdr_render_params)) + ": caller and library were built from different include/zdr.h");
    return ZDR_OK;
}

static int make_render_cfg(const zdr_render_params *p, bool backward, RenderCfg &R) {
    { int rc = check_params_abi(p); if (rc) return rc; }
    if (p->integrator < 0 || p->integrator > ZDR_UVGRAD) return fail(ZDR_E_INVALID, "unknown integrator");
    if (p->integrator == ZDR_UVGRAD && backward) return fail(ZDR_E_UNSUPPORTED, "render_duvdxy has no backward pass");
    if (p->width <= 0 || p->height <= 0) return fail(ZDR_E_INVALID, "bad resolution");
    if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 > p->x1 || p->y0 > p->y1) return fail(ZDR_E_INVALID, "bad pixel rectangle");
    if (p->sample_begin > p->sample_end || p->sample_end > p->spp) return fail(ZDR_E_INVALID, "bad sample range");
    if (p->tex_h <= 0 || p->tex_w <= 0) return fail(ZDR_E_INVALID, "bad texture size");
    if (p->max_depth < 1) return fail(ZDR_E_INVALID, "max_depth must be >= 1");
    if (p->tile_shard_count > 1 && (p->tile_shard_index < 0 || p->tile_shard_index >= p->tile_shard_count)) return fail(ZDR_E_INVALID, "tile_shard_index must lie in [0, tile_shard_count)");
    if (backward && p->integrator == ZDR_PATH && p->max_depth > ZDR_MAX_RECORDED_DEPTH) return fail(ZDR_E_UNSUPPORTED, "path backward records at most 16 vertices (prb.py:15)");
    memset(&R, 0, sizeof R);
    R.width = p->width; R.height = p->height; R.x0 = p->x0; R.y0 = p->y0; R.x1 = p->x1; R.y1 = p->y1;
    R.sample_begin = p->sample_begin; R.sample_end = p->sample_end;
    R.use_tent = p->use_tent; R.max_depth = p->max_depth; R.rr_depth = p->rr_depth; R.tex_h = p->tex_h; R.tex_w = p->tex_w;
    {   // few texels: replicate the staging cells so that the launch's atomics do not pile up on a handful of addresses (scene.h)
        const size_t ncells = (size_t)(p->tex_h + 1) * (size_t)(p->tex_w + 1);
        R.cell_copies = (ncells < (1u << 16)) ? (int32_t)std::min<size_t>(ZDR_MAX_CELL_COPIES, (1u << 20) / ncells) : 1;
    }
    R.two_over_w = 2.0f / (float)p->width; R.two_over_h = 2.0f / (float)p->height;
    R.aspect = (float)p->height / (float)p->width;
    R.inv_spp = 1.0f / (float)p->spp;
    R.alpha = (float)(p->sample_end - p->sample_begin) / (float)p->spp;
    h3 origin = H3(p->camera.origin[0], p->camera.origin[1], p->camera.origin[2]);
    h3 target = H3(p->camera.target[0], p->camera.target[1], p->camera.target[2]);
    h3 up = H3(p->camera.up[0], p->camera.up[1], p->camera.up[2]);
    h3 fwd = hnormalize(hsub(target, origin));                    // camera.py:12-14
    h3 right = hnormalize(hcross(fwd, up));
    h3 upp = hcross(right, fwd);
    R.cam_o[0] = origin.x; R.cam_o[1] = origin.y; R.cam_o[2] = origin.z;
    R.cam_fwd[0] = fwd.x; R.cam_fwd[1] = fwd.y; R.cam_fwd[2] = fwd.z;
    R.cam_right[0] = right.x; R.cam_right[1] = right.y; R.cam_right[2] = right.z;
    R.cam_upp[0] = upp.x; R.cam_upp[1] = upp.y; R.cam_upp[2] = upp.z;
    R.cam_tan = tanf(0.5f * p->camera.fov);                       // camera.py:15
    // work decomposition: 8x8 pixel tiles x sample chunks, enough single-wave workgroups to keep
    // 256 CUs x 4 SIMDs busy with several waves each and to let the dispatcher balance uneven tiles
    R.tiles_x = (p->x1 - p->x0 + 7) / 8; R.tiles_y = (p->y1 - p->y0 + 7) / 8;
    uint32_t ns = p->sample_end - p->sample_begin;
    if (p->prb_mode != ZDR_PRB_EXPECTATION && p->prb_mode != ZDR_PRB_DETACHED && p->prb_mode != ZDR_PRB_LITERAL) return fail(ZDR_E_INVALID, "unknown prb_mode");
    R.prb_mode = p->prb_mode;
    R.shard_count = p->tile_shard_count > 1 ? p->tile_shard_count : 1;
    R.shard_index = p->tile_shard_count > 1 ? p->tile_shard_index : 0;
    R.shard_skew = R.shard_count > 1 ? 1 : 0;
    const long all_tiles = (long)R.tiles_x * R.tiles_y;
    R.ntiles = (int32_t)(all_tiles > R.shard_index ? (all_tiles - R.shard_index + R.shard_count - 1) / R.shard_count : 0);
    long tiles = R.ntiles;
    long target_waves = (backward && p->integrator == ZDR_PATH) ? 131072 : 65536;   // (path backward, 4,096 resident waves since the kernel holds 16 per CU: 131072 11.22 ms, 65536 11.40; forward: the finer split is no faster) work items (tile x sample chunk) the persistent waves draw; cbox 512^2 spp 256: 16384 11.3/18.5 ms, 32768 10.7/17.7, 65536 10.25/17.2, 131072 10.2/17.1
    if (const char *e = getenv("ZDR_TARGET_WAVES")) target_waves = std::max(1L, atol(e));
    uint32_t min_chunk = 8;               // = one refill batch; cbox 512^2 forward / backward ms with 16 / 8 / 4: spp 16 1.10 / 0.99 / 1.06, 1.63 / 1.41 / 1.41; spp 64 2.90 / 2.86 / 2.92, 4.35 / 4.12 / 4.16; spp 256 unchanged
    if (const char *e = getenv("ZDR_MIN_CHUNK")) min_chunk = (uint32_t)std::max(1L, atol(e));
    long want = tiles > 0 ? (target_waves + tiles - 1) / tiles : 1;
    long maxc = std::max<long>(1, ns / min_chunk);
    long nchunks = std::max<long>(1, std::min(want, maxc));
    if (p->integrator == ZDR_UVGRAD) nchunks = 1;        // four-channel output, written directly
#ifdef ZDR_MEASURE   // timing-only ablations exist in measurement builds alone (zdr_kernels.hip, ZDR_ABLATE)
    if (const char *e = getenv("ZDR_DEBUG_NO_SCATTER")) R.debug_no_scatter = atoi(e);
#endif
    R.chunk = ns ? (uint32_t)((ns + nchunks - 1) / nchunks) : 1;
    R.nchunks = ns ? (int32_t)((ns + R.chunk - 1) / R.chunk) : 0;
    return ZDR_OK;
}

// A workspace that has to grow: not while the stream is capturing (an allocation cannot be recorded, and the capture would name a
// buffer that does not exist yet) — one eager call of the same kind and size beforehand sizes everything.
static int may_allocate(bool capturing, const char *what) {
    if (!capturing) return ZDR_OK;
    return fail(ZDR_E_UNSUPPORTED, std::string("the ") + what + " workspace would have to be allocated while the stream is capturing: make one eager call of this kind, resolution and spp on the handle first");
}

static int ensure_partial(zdr_scene *s, const RenderCfg &R, bool capturing) {
    if (R.nchunks <= 1) return ZDR_OK;
    size_t need = (size_t)R.nchunks * (size_t)R.ntiles * 64 * sizeof(float4);   // [chunk][tile of the shard][lane]
    if (need > s->partial_bytes) {
        if (int rc = may_allocate(capturing, "chunk-partial")) return rc;
        release_buffer(s, s->d_partial); s->d_partial = nullptr; s->partial_bytes = 0;
        HIPCHK(hipMalloc((void **)&s->d_partial, need));
        s->partial_bytes = need;
    }
    return ZDR_OK;
}

// Workspace of the path kernels: one FIFO of ZDR_RING_CAP x 64 parked camera-ray vertices (two float4 each,
// 32 KiB) per persistent workgroup, and the eight item counters the workgroups draw from (zeroed per launch).
static int ensure_ring(zdr_scene *s, hipStream_t st, bool capturing) {
    size_t need = (size_t)ZDR_MAX_PERSISTENT_BLOCKS * ZDR_RING_CAP * 2 * 64 * sizeof(float4);
    if (need > s->ring_bytes) {
        if (int rc = may_allocate(capturing, "parked-vertex FIFO")) return rc;
        release_buffer(s, s->d_ring); s->d_ring = nullptr; s->ring_bytes = 0;
        HIPCHK(hipMalloc((void **)&s->d_ring, need));
        s->ring_bytes = need;
    }
    if (!s->d_work_counters) { if (int rc = may_allocate(capturing, "work-counter")) return rc; HIPCHK(hipMalloc((void **)&s->d_work_counters, 8 * sizeof(unsigned int))); }
    if (zdr_launch_zero(s->d_work_counters, 8 * sizeof(unsigned int), st)) return fail(ZDR_E_HIP, "zero-fill launch failed");
    return ZDR_OK;
}

static int ensure_cells(zdr_scene *s, const RenderCfg &R, hipStream_t st, bool capturing) {
    size_t need = (size_t)R.cell_copies * (size_t)(R.tex_h + 1) * (R.tex_w + 1) * 16 * sizeof(float);
    if (need > s->cells_bytes) {
        if (int rc = may_allocate(capturing, "staging-cell")) return rc;
        release_buffer(s, s->d_cells); s->d_cells = nullptr; s->cells_bytes = 0;
        HIPCHK(hipMalloc((void **)&s->d_cells, need));
        s->cells_bytes = need;
    }
    if (zdr_launch_zero(s->d_cells, need, st)) return fail(ZDR_E_HIP, "zero-fill launch failed");
    return ZDR_OK;
}

// Reads (and clears) the device error word; the stream is synchronised first.  A set bit means a watchdog ended
// work early (zdr_kernels.hip: stall; accel.h: BVH budget): the image / gradient of the calls since the last
// check is incomplete.
static int check_device_error(zdr_scene *s, hipStream_t st) {
    unsigned int h = 0;
    HIPCHK(hipMemcpyAsync(&h, s->d_error, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!h) return ZDR_OK;
    HIPCHK(hipMemsetAsync(s->d_error, 0, sizeof h, st));
    std::string m = "device watchdog tripped:";
    if (h & ZDR_DEVERR_STALL) m += " a persistent path wave stopped without draining its work items;";
    if (h & ZDR_DEVERR_BVH_BUDGET) m += " a BVH walk exceeded its iteration budget;";
    if (h & ZDR_DEVERR_POOL) m += " a backward wave ended with record-pool slots leaked or handed out twice;";
    return fail(ZDR_E_HIP, m + " results since the last check are incomplete");
}

extern "C" int zdr_scene_check(zdr_scene *s, void *stream) {
    if (!s) return fail(ZDR_E_INVALID, "null scene");
    HIPCHK(hipSetDevice(s->device));
    return check_device_error(s, (hipStream_t)stream);
}

static int render_common(zdr_scene *s, const zdr_render_params *p, const float *material, float *image, const float *d_image,
                         float *d_material, int backward, int stats, void *stream) {
    if (!s || !p || !material) return fail(ZDR_E_INVALID, "null argument");
    int rc = check_params_abi(p); if (rc) return rc;
    HIPCHK(hipSetDevice(s->device));
    RenderCfg R; SamplerCfg C;
    rc = make_render_cfg(p, backward != 0, R); if (rc) return rc;
    rc = make_sampler_cfg(s, p->sampler, p->seed, p->spp, C); if (rc) return rc;
    const bool capturing = stream_is_capturing((hipStream_t)stream);
    if (capturing) s->captured = true;                  // sticky: a graph may name this handle's buffers from now on (zdr_scene)
    if (backward) { rc = ensure_cells(s, R, (hipStream_t)stream, capturing); if (rc) return rc; }
    else if (!stats) { rc = ensure_partial(s, R, capturing); if (rc) return rc; }
    if (p->integrator == ZDR_PATH) {
        if (p->spp > (1u << 25)) return fail(ZDR_E_UNSUPPORTED, "path integrator: spp above 2^25");   // queue entries pack pixel << 26 | bank << 25 | sample
        rc = ensure_ring(s, (hipStream_t)stream, capturing); if (rc) return rc;
    }
    KernelIO io; memset(&io, 0, sizeof io);
    io.ring = s->d_ring; io.work_counters = s->d_work_counters;
    // brute-force scenes of at most 64 triangle pairs: camera rays test only the pairs their tile can see
    const bool masks = !s->accel_is_bvh && s->ds.ntris <= 128 && p->integrator != ZDR_UVGRAD && !getenv("ZDR_NO_TILE_MASKS");
    if (masks) {
        size_t need = (size_t)R.tiles_x * R.tiles_y * sizeof(unsigned long long);
        if (need > s->tile_mask_bytes) {
            if (int rc2 = may_allocate(capturing, "tile-mask")) return rc2;
            release_buffer(s, s->d_tile_masks); s->d_tile_masks = nullptr; s->tile_mask_bytes = 0; s->tile_mask_key_set = false;
            HIPCHK(hipMalloc((void **)&s->d_tile_masks, need));
            s->tile_mask_bytes = need;
        }
        io.tile_masks = s->d_tile_masks;
        // the masks depend on the camera and the tile grid only (the geometry of a scene handle never changes):
        // an optimisation loop that renders the same view again does not rebuild them
        float key[24] = {(float)R.x0, (float)R.y0, (float)R.x1, (float)R.y1, (float)R.width, (float)R.height, R.two_over_w, R.two_over_h, R.aspect, R.cam_tan,
                         R.cam_o[0], R.cam_o[1], R.cam_o[2], R.cam_fwd[0], R.cam_fwd[1], R.cam_fwd[2], R.cam_right[0], R.cam_right[1], R.cam_right[2],
                         R.cam_upp[0], R.cam_upp[1], R.cam_upp[2], (float)R.tiles_x, (float)R.tiles_y};
        io.tile_masks_valid = (s->tile_mask_key_set && memcmp(key, s->tile_mask_key, sizeof key) == 0) ? 1 : 0;
        // A captured call must carry its own k_tile_masks (a replay may follow an eager render of another view on this handle), and
        // once a graph exists its replays rewrite the buffer unseen by the host: the key is never trusted again.
        if (s->captured) io.tile_masks_valid = 0;
        // the key is recorded only when k_tile_masks is really going to run: a shard that owns no tile (more shards than
        // tiles) launches nothing, and must not leave the key of masks nobody built behind for the next call
        if ((long)R.ntiles * R.nchunks > 0) { memcpy(s->tile_mask_key, key, sizeof key); s->tile_mask_key_set = true; }
    }
    io.material = (const float4 *)material; io.image = (float4 *)image; io.partial = s->d_partial;
    io.d_image = (const float4 *)d_image; io.d_material = d_material; io.cells = s->d_cells; io.counters = s->d_counters;
    // every pointer a kernel variant dereferences must be live before anything is launched
    if (backward && (!io.d_image || !io.d_material || !io.cells)) return fail(ZDR_E_INVALID, "backward needs d_image, d_material and the staging cells");
    if (!backward && !stats && !io.image) return fail(ZDR_E_INVALID, "forward needs an image");
    if (!backward && !stats && R.nchunks > 1 && !io.partial) return fail(ZDR_E_NOMEM, "chunk workspace missing");
    if (stats && !io.counters) return fail(ZDR_E_NOMEM, "counter buffer missing");
    if (p->integrator == ZDR_PATH && (!io.ring || !io.work_counters)) return fail(ZDR_E_NOMEM, "path workspace missing");
    if (stats && p->integrator == ZDR_UVGRAD) return fail(ZDR_E_UNSUPPORTED, "no statistics for render_duvdxy");
    DScene S = s->ds;
    // shadow segments end on a light's surface; an environment light sends them to infinity, where nothing can be ruled out
    S.shadow_pairs = (S.env_count > 0) ? ~0ull : s->shadow_pairs;
    if (zdr_launch_render(S, R, C, io, p->integrator, s->accel_is_bvh, backward, stats, (hipStream_t)stream))
        return fail(ZDR_E_HIP, std::string("kernel launch: ") + hipGetErrorString(hipGetLastError()));
    static const bool check_every_call = getenv("ZDR_CHECK") && atoi(getenv("ZDR_CHECK")) != 0;   // opt-in: costs a synchronise per call
    if (check_every_call && !stats && !capturing) return check_device_error(s, (hipStream_t)stream);   // (a synchronise cannot be captured: zdr_scene_check after the replay instead)
    return ZDR_OK;
}

extern "C" int zdr_render_forward(zdr_scene *s, const zdr_render_params *p, const float *material, float *image, void *stream) {
    if (!image) return fail(ZDR_E_INVALID, "null image");
    return render_common(s, p, material, image, nullptr, nullptr, 0, 0, stream);
}

extern "C" int zdr_render_backward(zdr_scene *s, const zdr_render_params *p, const float *d_image, const float *material,
                                   float *d_material, void *stream) {
    if (!d_image || !d_material) return fail(ZDR_E_INVALID, "null gradient buffer");
#ifdef ZDR_MEASURE_STATS   // measurement build (tools/bwd_stats.sh): the path backward kernel counts its trips, sweep iterations and flushes
    if (s) { (void)hipSetDevice(s->device); (void)hipMemsetAsync(s->d_counters, 0, 8 * sizeof(unsigned long long), (hipStream_t)stream); }
    int rc = render_common(s, p, material, nullptr, d_image, d_material, 1, 0, stream);
    if (!rc) {
        unsigned long long h[8];
        (void)hipMemcpyAsync(h, s->d_counters, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream); (void)hipStreamSynchronize((hipStream_t)stream);
        fprintf(stderr, "[bwd stats] trips %llu shaded %llu finished %llu sweep_iterations %llu sweep_steps %llu flushes %llu entries %llu duplicate_cells %llu\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return rc;
#else
    return render_common(s, p, material, nullptr, d_image, d_material, 1, 0, stream);
#endif
}

extern "C" int zdr_render_stats(zdr_scene *s, const zdr_render_params *p, const float *material, uint64_t counters[8], void *stream) {
    if (!s || !counters) return fail(ZDR_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(s->d_counters, 0, 8 * sizeof(unsigned long long), st));
    int rc = render_common(s, p, material, nullptr, nullptr, nullptr, 0, 1, stream); if (rc) return rc;
    unsigned long long h[8];
    HIPCHK(hipMemcpyAsync(h, s->d_counters, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) counters[i] = h[i];
    return check_device_error(s, st);
}

extern "C" int zdr_trace_closest(zdr_scene *s, const float *rays, uint32_t n, int32_t *inst_prim, float *bary_t, void *stream) {
    if (!s || !rays || !inst_prim || !bary_t) return fail(ZDR_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(s->device));
    if (zdr_launch_trace(s->ds, s->accel_is_bvh, 0, rays, n, inst_prim, bary_t, (hipStream_t)stream)) return fail(ZDR_E_HIP, "trace launch failed");
    return ZDR_OK;
}

extern "C" int zdr_trace_any(zdr_scene *s, const float *rays, uint32_t n, int32_t *occluded, void *stream) {
    if (!s || !rays || !occluded) return fail(ZDR_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(s->device));
    if (zdr_launch_trace(s->ds, s->accel_is_bvh, 1, rays, n, occluded, nullptr, (hipStream_t)stream)) return fail(ZDR_E_HIP, "trace launch failed");
    return ZDR_OK;
}

extern "C" int zdr_path_dump(zdr_scene *s, const zdr_render_params *p, const float *material, const float *d_image,
                             const int32_t *queries, uint32_t n, int32_t maxv, float *out, void *stream) {
    if (!s || !p || !material || !queries || !out) return fail(ZDR_E_INVALID, "null argument");
    if (int rc = check_params_abi(p)) return rc;
    if (p->integrator != ZDR_PATH) return fail(ZDR_E_UNSUPPORTED, "path traces exist for the path integrator only");
    if (maxv < 1 || maxv > ZDR_MAX_RECORDED_DEPTH) return fail(ZDR_E_INVALID, "maxv must lie in [1, 16]");
    HIPCHK(hipSetDevice(s->device));
    RenderCfg R; SamplerCfg C;
    int rc = make_render_cfg(p, true, R); if (rc) return rc;
    rc = make_sampler_cfg(s, p->sampler, p->seed, p->spp, C); if (rc) return rc;
    KernelIO io; memset(&io, 0, sizeof io);
    io.material = (const float4 *)material; io.d_image = (const float4 *)d_image;
    DScene S = s->ds;
    S.shadow_pairs = (S.env_count > 0) ? ~0ull : s->shadow_pairs;
    if (zdr_launch_path_dump(S, R, C, io, s->accel_is_bvh, queries, n, maxv, out, (hipStream_t)stream)) return fail(ZDR_E_HIP, "path dump launch failed");
    return ZDR_OK;
}

extern "C" int zdr_sampler_dump(zdr_scene *s, int32_t sampler, uint32_t seed, uint32_t spp, const int32_t *queries, uint32_t n,
                                int32_t nvert, int32_t rr_depth, float *out, void *stream) {
    if (!s || !queries || !out || nvert < 0) return fail(ZDR_E_INVALID, "bad argument");
    HIPCHK(hipSetDevice(s->device));
    SamplerCfg C;
    int rc = make_sampler_cfg(s, sampler, seed, spp, C); if (rc) return rc;
    if (zdr_launch_sampler_dump(C, queries, n, nvert, rr_depth, out, 0, nullptr, (hipStream_t)stream)) return fail(ZDR_E_HIP, "sampler dump launch failed");
    return ZDR_OK;
}

extern "C" int zdr_vertex_sampler_dump(zdr_scene *s, int32_t integrator, int32_t sampler, uint32_t seed, uint32_t spp, const int32_t *queries, uint32_t n,
                                       int32_t nvert, int32_t rr_depth, float *out, int32_t *batched, void *stream) {
    if (!s || !queries || !out || nvert < 0) return fail(ZDR_E_INVALID, "bad argument");
    if (integrator != ZDR_PATH && integrator != ZDR_DIRECT) return fail(ZDR_E_INVALID, "the path and the direct integrator have draw routes of their own");
    HIPCHK(hipSetDevice(s->device));
    SamplerCfg C;
    int rc = make_sampler_cfg(s, sampler, seed, spp, C); if (rc) return rc;
    int b = 0;
    if (zdr_launch_sampler_dump(C, queries, n, nvert, rr_depth, out, integrator == ZDR_DIRECT ? 2 : 1, &b, (hipStream_t)stream)) return fail(ZDR_E_HIP, "sampler dump launch failed");
    if (batched) *batched = b;
    return ZDR_OK;
}
