// vecmath.h — float3 helpers for gfx950 device code (wave64; no fast-math: the integrators rely
// on IEEE NaN/inf propagation, integrator.py:27, prb.py:100,179).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZD __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

ZD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
ZD f3 mk3(float s) { return mk3(s, s, s); }
ZD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
ZD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
ZD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
ZD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
ZD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
ZD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
ZD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ZD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// 1-ulp hardware reciprocal / rsqrt / sqrt (v_rcp_f32, v_rsq_f32, v_sqrt_f32): same special-value
// behaviour as IEEE division for 0, inf and NaN operands, which is what the NaN policy needs.
ZD float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
ZD float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
ZD float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
ZD f3 operator/(f3 a, float s) { float r = rcp(s); return mk3(a.x * r, a.y * r, a.z * r); }
ZD f3 normalize(f3 a) { return a * rsq(dot(a, a)); }
ZD float length(f3 a) { return fsqrt(dot(a, a)); }
ZD bool any_nan(f3 a) { return (a.x != a.x) | (a.y != a.y) | (a.z != a.z); }
ZD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
ZD int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }
ZD float lerpf(float a, float b, float t) { return a + t * (b - a); }
ZD f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }

#define ZDR_PI 3.14159265358979323846f
#define ZDR_INV_PI 0.31830988618379067154f
