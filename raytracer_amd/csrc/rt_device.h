// rt_device.h -- device-side arithmetic of the trace loop: vector helpers, glm::perlin, the texture functions,
// primitive intersection and hit reconstruction.  Every routine cites the reference lines whose result it must
// reproduce.  This translation unit is compiled with -ffp-contract=off so that `a*b+c` stays two roundings like
// the reference's x86-64 build; where fusing cannot change which primitive wins (box slabs) fmaf is explicit.
// hipcc's default correctly-rounded f32 divide/sqrt is relied upon (1/sqrt in normalize, t = num/den).
#pragma once
#include <hip/hip_runtime.h>
#include "rt_types.h"

#define RT_DEV __device__ __forceinline__

struct V2 { float x, y; };
struct V3 { float x, y, z; };
RT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 v3s(float s) { return V3{s, s, s}; }
RT_DEV V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
RT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RT_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RT_DEV V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
RT_DEV V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
// glm/detail/func_geometric.inl:48-55, 70-78, 10-13, 82-90
RT_DEV float dot(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
RT_DEV V3 cross(V3 x, V3 y) { return V3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
RT_DEV float length(V3 v) { return sqrtf(dot(v, v)); }
RT_DEV V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
RT_DEV float gmin(float x, float y) { return (y < x) ? y : x; }  // glm::min
RT_DEV float gmax(float x, float y) { return (x < y) ? y : x; }  // glm::max
RT_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
RT_DEV float smax(float a, float b) { return (a < b) ? b : a; }  // std::max(a, b)

// xyz of mat4 * (v, w), glm/detail/type_mat4x4.inl:561-572: (m0*v0 + m1*v1) + (m2*v2 + m3*w)
RT_DEV V3 mulPoint(const float *m, V3 v, float w) {
    V3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * w);
    r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * w);
    r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * w);
    return r;
}

// ---- shared counter-based jitter (same formula in oracle/ref_harness.cpp and oracle/rt_oracle.cpp) ----------
RT_DEV uint32_t rtHash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
RT_DEV void rtHashJitter(uint32_t pixel, uint32_t sample, uint32_t seed, float &jx, float &jy) {
    uint32_t a = rtHash32(pixel ^ (seed * 0x85ebca6bU + 0x9e3779b9U));
    uint32_t b = rtHash32(a + 2u * sample + 1u), c = rtHash32(a + 2u * sample + 2u);
    jx = ((float)(b >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.5f;
    jy = ((float)(c >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.5f;
}

// ---- glm::perlin(vec3): glm/gtc/noise.inl:65-132 + glm/detail/_noise.hpp:8-78 ---------------------------------
// Pure ALU: the hash is the polynomial permute(x) = mod289((34x+1)x); there is no permutation table to stage
// (SURVEY.md F2).
RT_DEV float pMod289(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
RT_DEV float pPermute(float x) { return pMod289(((x * 34.0f) + 1.0f) * x); }
RT_DEV float pFract(float x) { return x - floorf(x); }
RT_DEV float pStep(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
RT_DEV float pMix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RT_DEV float pFade(float t) { return (t * t * t) * (t * (t * 6.0f - 15.0f) + 10.0f); }
RT_DEV float pTaylorInvSqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }
RT_DEV V3 pGradient(float h) {  // one lattice corner: noise.inl:82-88
    const float seventh = (float)(1.0 / 7.0);
    float gx = h * seventh;
    float gy = pFract(floorf(gx) * seventh) - 0.5f;
    gx = pFract(gx);
    float gz = 0.5f - fabsf(gx) - fabsf(gy);
    float sz = pStep(gz, 0.0f);
    gx -= sz * (pStep(0.0f, gx) - 0.5f);
    gy -= sz * (pStep(0.0f, gy) - 0.5f);
    V3 g = v3(gx, gy, gz);
    return g * pTaylorInvSqrt(dot(g, g));  // noise.inl:107-116
}

RT_DEV float rtPerlin3(float px, float py, float pz) {
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    const float x0 = pMod289(flx), y0 = pMod289(fly), z0 = pMod289(flz);
    const float x1 = pMod289(flx + 1.0f), y1 = pMod289(fly + 1.0f), z1 = pMod289(flz + 1.0f);
    const float fx0 = px - flx, fy0 = py - fly, fz0 = pz - flz;
    const float fx1 = fx0 - 1.0f, fy1 = fy0 - 1.0f, fz1 = fz0 - 1.0f;
    const float px0 = pPermute(x0), px1 = pPermute(x1);
    const float ixy00 = pPermute(px0 + y0), ixy10 = pPermute(px1 + y0);
    const float ixy01 = pPermute(px0 + y1), ixy11 = pPermute(px1 + y1);
    const V3 g000 = pGradient(pPermute(ixy00 + z0)), g100 = pGradient(pPermute(ixy10 + z0));
    const V3 g010 = pGradient(pPermute(ixy01 + z0)), g110 = pGradient(pPermute(ixy11 + z0));
    const V3 g001 = pGradient(pPermute(ixy00 + z1)), g101 = pGradient(pPermute(ixy10 + z1));
    const V3 g011 = pGradient(pPermute(ixy01 + z1)), g111 = pGradient(pPermute(ixy11 + z1));
    const float n000 = dot(g000, v3(fx0, fy0, fz0)), n100 = dot(g100, v3(fx1, fy0, fz0));
    const float n010 = dot(g010, v3(fx0, fy1, fz0)), n110 = dot(g110, v3(fx1, fy1, fz0));
    const float n001 = dot(g001, v3(fx0, fy0, fz1)), n101 = dot(g101, v3(fx1, fy0, fz1));
    const float n011 = dot(g011, v3(fx0, fy1, fz1)), n111 = dot(g111, v3(fx1, fy1, fz1));
    const float ux = pFade(fx0), uy = pFade(fy0), uz = pFade(fz0);
    const float nz0 = pMix(n000, n001, uz), nz1 = pMix(n100, n101, uz), nz2 = pMix(n010, n011, uz), nz3 = pMix(n110, n111, uz);
    const float nyz0 = pMix(nz0, nz2, uy), nyz1 = pMix(nz1, nz3, uy);
    return 2.2f * pMix(nyz0, nyz1, ux);
}

// ---- texture functions, Textures.h:12-188 ---------------------------------------------------------------------
RT_DEV V3 rtLerpCols(V3 p, V3 c1, V3 c2) {  // Textures.h:94-96
    return v3(p.x * c1.x + ((1.0f - p.x) * c2.x), p.y * c1.y + ((1.0f - p.y) * c2.y), p.z * c1.z + ((1.0f - p.z) * c2.z));
}
// pixelAt, Textures.h:156-166: nearest texel, bottom-up v, BGR -> RGB, /255.  Row index == height (fmod(v,1) == 0)
// is an out-of-bounds read in the reference; here it repeats the last row (same choice as the oracle).
RT_DEV V3 rtPixelAt(const RtSceneDev &sc, int slot, V2 uv) {
    const int W = sc.bmp_w[slot], H = sc.bmp_h[slot], C = sc.bmp_c[slot];
    if (W == 0) return v3s(0.0f);
    int x = (int)(W * fmod((double)uv.x, 1.0)) & 0xFFFF;
    int y = (int)(H - H * fmod((double)uv.y, 1.0)) & 0xFFFF;
    if (y >= H) y = H - 1;
    if (x >= W) x = W - 1;
    const uint8_t *p = sc.bmp[slot] + (size_t)C * ((size_t)W * y + x);
    return v3(p[2] / 255.f, p[1] / 255.f, p[0] / 255.f);
}

// Inlined at its call sites (diffuse colour, normal map, roughness, occlusion).  Measured on k_shade: one out-of-line
// copy costs the caller its caller-saved registers around every call (222 ms), one inlined copy inside a rolled loop
// over the four textures is no better (220 ms), four inlined copies are fastest (210 ms; 6.9 K VALU instructions,
// still inside the instruction cache).  The Perlin evaluation sits in a rolled loop (opal needs three).
// FAST: the caller guarantees that no BMP map is sampled (fast k_shade: such materials go to the general kernel)
template <bool FAST = false>
__device__ inline V3 rtEvalTexture(const RtSceneDev &sc, int id, V2 uv) {
    if (!FAST && id >= RT_TEX_BMP_COLOR) {  // Textures.h:173-188
        V3 c = rtPixelAt(sc, id - RT_TEX_BMP_COLOR, uv);
        return (id >= RT_TEX_BMP_ROUGHNESS) ? v3s(c.x) : c;
    }
    if (id == RT_TEX_QWILFISH) {  // Textures.h:123-132
        float y = 0.5f * (uv.y + 3.0f);
        if (y < 1) return v3((float)0.937, (float)0.922, (float)0.392);
        return v3(0.0f, (float)0.416, (float)0.42);
    }
    if (id == RT_TEX_NONE) return v3s(0.0f);
    // Perlin family.  perlinCalculations (Textures.h:12-19) evaluates three channels; where the reference's
    // channel arguments coincide (terrain, ice, snow) or a channel is unused (perlinNormal) or is perlin(0,0,0) = 0
    // (the r,g channels of perlinWater: u/v scales and offset 0), one evaluation gives the same values.
    float us = 50.0f, vs = 50.0f, off = 50.0f;  // perlinNormal / perlinWater, b channel
    int n = 1;
    if (id == RT_TEX_OPAL) { n = 3; vs = 20.0f; }
    else if (id == RT_TEX_PERLIN_TERRAIN) { us = 30.0f; vs = 30.0f; off = 0.0f; }
    else if (id == RT_TEX_PERLIN_ICE_TERRAIN || id == RT_TEX_SNOW_TERRAIN) { us = 10.0f; vs = 10.0f; off = 10.0f; }
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
#pragma nounroll
    for (int k = 0; k < n; k++) {
        if (id == RT_TEX_OPAL) {  // Textures.h:56-58
            us = (k == 0) ? 20.0f : (k == 1 ? 10.0f : 15.0f);
            off = (k == 0) ? 0.2f : (k == 1 ? 0.3f : 0.4f);
        }
        float q = rtPerlin3(uv.x * us, uv.y * vs, off);
        if (k == 0) q0 = q; else if (k == 1) q1 = q; else q2 = q;
    }
    switch (id) {
        case RT_TEX_PERLIN_NORMAL: {  // Textures.h:22-36
            float pz = 2.0f * q0 - 1.0f;
            pz = -pz;
            pz = (pz + 1.0f) / 2.0f;
            return normalize(v3(uv.x, uv.y, pz));
        }
        case RT_TEX_PERLIN_WATER: {  // Textures.h:39-52 with p.r = p.g = 0
            V3 p = v3(0.0f, 0.0f, q0);
            p = 2.0f * p - v3s(1.0f);
            p = normalize(p);
            p = (p + v3s(1.0f)) / 2.0f;
            return normalize(v3(uv.x, uv.y, p.z));
        }
        case RT_TEX_OPAL: {  // Textures.h:55-73
            V3 p = v3(q0, q1, q2);
            p = 2.0f * p - v3s(1.0f);
            p = normalize(p);
            p = (p + v3s(1.0f)) / 2.0f;
            return v3(0.5f + 0.4f * p.x, 0.5f + 0.3f * p.y, 0.5f + 0.2f * p.z);
        }
        case RT_TEX_PERLIN_TERRAIN:  // Textures.h:76-84
            return v3(0.2f + 0.4f * q0, 0.2f + 0.3f * q0, 0.2f + 0.2f * q0);
        case RT_TEX_PERLIN_ICE_TERRAIN:  // Textures.h:87-99
            return rtLerpCols(v3s(q0), v3((float)0.722, (float)0.961, (float)0.937), v3((float)0.075, (float)0.482, (float)0.631));
        case RT_TEX_SNOW_TERRAIN:  // Textures.h:102-114
            return rtLerpCols(v3s(q0), v3((float)0.722, (float)0.961, (float)0.937), v3((float)0.212, (float)0.51, (float)0.62));
    }
    return v3s(0.0f);
}

// ---- primitive intersection ------------------------------------------------------------------------------------
struct RtRay { V3 o, d; };

// Sphere::intersect up to the distance, Objects.h:38-72.  Returns false on a miss.
// `ni` = object-space hit point (needed again by the shading reconstruction).
RT_DEV bool rtSphereHit(const RtSphere &sp, const RtRay &r, float &dist, V3 &ni, V3 &P) {
    V3 no = mulPoint(sp.Minv, r.o, 1.0f);
    V3 nd = normalize(mulPoint(sp.Minv, r.d, 0.0f));
    V3 c = v3s(0.0f) - no;
    float cdotc = dot(c, c), cdotd = dot(c, nd);
    float Df = 0.0f;
    if (cdotc > cdotd * cdotd) Df = sqrtf(cdotc - cdotd * cdotd);
    if (!(Df <= 1.0f)) return false;
    // the reference evaluates the roots in double (Objects.h:51-59)
    const double D = (double)Df;
    const double sq = sqrt(1.0 - D * D);
    float t1 = (float)((double)cdotd - sq);
    float t2 = (float)((double)cdotd + sq);
    float t = (t1 < 0) ? t2 : t1;
    if (t < 0) return false;
    ni = no + t * nd;
    P = mulPoint(sp.M, ni, 1.0f);
    dist = length(P - r.o);
    return true;
}

// Cone::intersect up to the object-space hit, Objects.h:212-254: unit cone x^2 + z^2 = y^2 for 0 <= y <= 1 (apex at the
// origin) closed by the disc y = 1.  ni = object-space intersection, nn = object-space normal, P = world intersection.
RT_DEV bool rtConeHit(const RtSphere &sp, const RtRay &r, float &dist, V3 &ni, V3 &nn, V3 &P) {
    const float height = 1.0f;
    const V3 no = mulPoint(sp.Minv, r.o, 1.0f);
    const V3 nd = normalize(mulPoint(sp.Minv, r.d, 0.0f));
    const float a = nd.x * nd.x + nd.z * nd.z - nd.y * nd.y;
    const float b = 2 * (no.x * nd.x + no.z * nd.z - no.y * nd.y);
    const float c = no.x * no.x + no.z * no.z - no.y * no.y;
    const float delta = b * b - 4 * a * c;
    if (delta < 0) return false;
    const float t1 = (-b - sqrtf(delta)) / (2 * a);
    const float t2 = (-b + sqrtf(delta)) / (2 * a);
    float t = t1;
    ni = no + t * nd;
    if (t < 0 || ni.y > height || ni.y < 0) {
        t = t2;
        ni = no + t * nd;
        if (t < 0 || ni.y > height || ni.y < 0) return false;
    }
    nn = normalize(v3(ni.x, -ni.y, ni.z));
    // the base, a Plane(point (0,1,0), normal (0,1,0)) intersected in object space (Objects.h:138-158, :246-253):
    // it replaces the side hit when it is closer and inside the unit disc -- only rays that also hit the side get here
    const V3 bn = v3(0.0f, 1.0f, 0.0f);
    const float ddotN = dot(nd, bn);
    if (ddotN != 0) {
        const float podotN = dot(bn - no, bn);
        const float tb = podotN / ddotN;
        if (!(tb < 0)) {
            const V3 bi = no + tb * nd;
            if (tb < t && length(bi - bn) <= 1.0f) {
                ni = bi;
                nn = bn;
            }
        }
    }
    P = mulPoint(sp.M, ni, 1.0f);
    dist = length(P - r.o);
    return true;
}

// Plane::intersect, Objects.h:138-158
RT_DEV bool rtPlaneHit(const RtPlane &pl, const RtRay &r, float &t) {
    V3 n = ld3(pl.normal);
    float ddotN = dot(r.d, n);
    if (ddotN == 0) return false;
    float podotN = dot(ld3(pl.point) - r.o, n);
    t = podotN / ddotN;
    if (t < 0) return false;
    return true;
}

// either transformed unit primitive, by kind
RT_DEV bool rtUnitPrimHit(const RtSphere &sp, const RtRay &r, float &dist, V3 &ni, V3 &P) {
    if (sp.kind == RT_KIND_CONE) {
        V3 nn;
        return rtConeHit(sp, r, dist, ni, nn, P);
    }
    return rtSphereHit(sp, r, dist, ni, P);
}

struct RtTri { V3 A, B, C, n; };
RT_DEV RtTri rtLoadTri(const RtTriGeom *g, uint32_t slot) {
    const float4 *p = reinterpret_cast<const float4 *>(g + slot);
    float4 q0 = p[0], q1 = p[1], q2 = p[2];
    RtTri T;
    T.A = v3(q0.x, q0.y, q0.z);
    T.B = v3(q1.x, q1.y, q1.z);
    T.C = v3(q2.x, q2.y, q2.z);
    T.n = v3(q0.w, q1.w, q2.w);
    return T;
}

// Triangle::intersect, MeshLoader.h:70-97.  `tlimit`: candidates with t > tlimit cannot be selected by the callers'
// strict-min / <= tests, so they are dropped before the inside tests (same result, less work).
// w0..w2 / d0..d2 are returned for the attribute reconstruction.
RT_DEV bool rtTriHit(const RtTri &T, const RtRay &r, float tlimit, float &t) {
    float ddotN = dot(r.d, T.n);
    if (ddotN == 0) return false;
    float podotN = dot(T.A - r.o, T.n);
    t = podotN / ddotN;
    if (t < 0) return false;
    if (!(t <= tlimit)) return false;  // also drops NaN distances, which never win in the reference either
    V3 td = r.o + t * r.d;
    V3 w0 = cross(T.B - td, T.C - td);
    V3 w1 = cross(T.C - td, T.A - td);
    V3 w2 = cross(T.A - td, T.B - td);
    if (dot(T.n, w0) < 0 || dot(T.n, w1) < 0 || dot(T.n, w2) < 0) return false;
    return true;
}

// ---- the full Hit record of the reference (Core.h:34-57) rebuilt from (ray, t, id) -----------------------------
struct RtFullHit {
    V3 P, normal, ns, T, B;
    V2 uv;
    int32_t object, material;
    uint32_t prim;     // reported primitive id
    int32_t triangle;  // slot (for meshes) or -1
};

template <bool FAST = false>
RT_DEV V3 rtApplyNormalMap(const RtSceneDev &sc, const rt_material &m, V3 tangent, V3 bitangent, V3 normal, V2 uv) {
    V3 nm = normalize(rtEvalTexture<FAST>(sc, m.normal_map, uv));
    // mat3(T,B,N) * nm, glm/detail/type_mat3x3.inl:468-474
    V3 r = v3(tangent.x * nm.x + bitangent.x * nm.y + normal.x * nm.z, tangent.y * nm.x + bitangent.y * nm.y + normal.y * nm.z,
              tangent.z * nm.x + bitangent.z * nm.y + normal.z * nm.z);
    return normalize(r);
}

// FAST: the caller guarantees that the hit is a triangle or a plane without BMP maps (fast k_shade)
template <bool FAST = false>
RT_DEV void rtReconstructHit(const RtSceneDev &sc, const RtRay &r, float t, uint32_t id, RtFullHit &h) {
    h.T = v3s(0.0f);
    h.B = v3s(0.0f);
    h.uv = V2{0.0f, 0.0f};
    h.triangle = -1;
    const uint32_t kind = id & RT_ID_KIND_MASK;
    if (!FAST && kind == RT_ID_SPHERE) {  // Objects.h:67-95
        const RtSphere &sp = sc.spheres[id & 0xFFFF];
        float dist;
        V3 ni, nn;
        if (sp.kind == RT_KIND_CONE) {  // Objects.h:255-275: no normal map on cones, uv from the object-space normal
            rtConeHit(sp, r, dist, ni, nn, h.P);
        } else {
            rtSphereHit(sp, r, dist, ni, h.P);
            nn = normalize(ni - v3s(0.0f));
        }
        V3 ng = normalize(mulPoint(sp.N, nn, 0.0f));
        h.normal = ng;
        h.ns = ng;
        const double kPi = 3.14159265358979323846;
        h.uv.x = (float)((asin((double)nn.y) + kPi / 2) / kPi);
        h.uv.y = (float)((atan2((double)nn.z, (double)nn.x) + kPi) / (2 * kPi));
        h.object = sp.object;
        h.material = sp.material;
        h.prim = RT_PRIM_ANALYTIC | (uint32_t)sp.object;
        const rt_material &m = sc.materials[sp.material];
        if (m.has_normal_map && sp.kind == RT_KIND_SPHERE) {
            h.T = normalize(cross(v3(0.0f, 1.0f, 0.0f), h.P));
            h.B = normalize(cross(h.normal, h.T));
            h.ns = rtApplyNormalMap(sc, m, h.T, h.B, h.normal, h.uv);
        }
    } else if (kind == RT_ID_PLANE) {  // Objects.h:156-179
        const RtPlane &pl = sc.planes[id & 0xFFFF];
        h.P = r.o + t * r.d;
        h.normal = ld3(pl.normal);
        h.ns = h.normal;
        h.uv.x = 0.1f * h.P.x;
        h.uv.y = 0.1f * h.P.z;
        h.object = pl.object;
        h.material = pl.material;
        h.prim = RT_PRIM_ANALYTIC | (uint32_t)pl.object;
        const rt_material &m = sc.materials[pl.material];
        if (m.has_normal_map) {
            h.T = v3(0, 0, 1);
            h.B = v3(1, 0, 0);
            h.ns = rtApplyNormalMap<FAST>(sc, m, h.T, h.B, h.normal, h.uv);
        }
    } else {  // MeshLoader.h:89-121
        id &= RT_ID_TRI_MASK;
        const RtTri T = rtLoadTri(sc.tri_geom, id);
        const RtTriShade &s = sc.tri_shade[id];
        V3 td = r.o + t * r.d;
        V3 w0 = cross(T.B - td, T.C - td);
        V3 w1 = cross(T.C - td, T.A - td);
        V3 w2 = cross(T.A - td, T.B - td);
        float d0 = dot(T.n, w0), d1 = dot(T.n, w1), d2 = dot(T.n, w2);
        if (s.flags & 1u) {
            float a0 = length(w0) * (d0 >= 0 ? 1.0f : -1.0f) * 0.5f;
            float a1 = length(w1) * (d1 >= 0 ? 1.0f : -1.0f) * 0.5f;
            float a2 = length(w2) * (d2 >= 0 ? 1.0f : -1.0f) * 0.5f;
            float totA = a0 + a1 + a2;
            h.normal = normalize((a0 / totA) * ld3(s.nA) + (a1 / totA) * ld3(s.nB) + (a2 / totA) * ld3(s.nC));
        } else {
            h.normal = T.n;
        }
        h.P = td;
        h.ns = T.n;
        if (s.flags & 2u) {
            float alpha = d0 / length(w0), beta = d1 / length(w1), gamma = d2 / length(w2);
            h.uv = alpha * V2{s.tA[0], s.tA[1]} + beta * V2{s.tB[0], s.tB[1]} + gamma * V2{s.tC[0], s.tC[1]};
        }
        h.object = s.object;
        h.material = s.material;
        h.prim = s.prim;
        h.triangle = (int32_t)id;
    }
}

// glm::reflect / refract / angle: glm/detail/func_geometric.inl:104-123, glm/gtx/vector_angle.inl:16-21
RT_DEV V3 rtReflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
RT_DEV bool rtRefract(V3 I, V3 N, float eta, V3 &out) {
    float dv = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - dv * dv);
    if (k >= 0.0f) {
        out = eta * I - (eta * dv + sqrtf(k)) * N;
        return true;
    }
    out = v3s(0.0f);
    return false;
}
RT_DEV float rtAngle(V3 x, V3 y) { return acosf(gclamp(dot(x, y), -1.0f, 1.0f)); }

// toneMapping, main.cpp:240-256
RT_DEV float rtToneMap1(float x) {
    const float alpha = 1.5f, beta = 1.8f, gamma = 2.2f, oog = 1.0f / gamma;
    float n = powf(alpha * powf(x, beta), oog);
    return gclamp(gmin(n, 1.0f), 0.0f, 1.0f);
}

// fixed-point accumulator conversion (DESIGN.md "deterministic accumulation")
RT_DEV long long rtToFixed(float x) {
    if (!(x == x)) return 0;  // NaN contributions are dropped (the reference would poison the pixel)
    x = fminf(fmaxf(x, -RT_ACC_LIMIT), RT_ACC_LIMIT);
    // x = hi + fr exactly (|x| <= 2^24, fr in [0,1)); fr * 2^32 is an integer below 2^32 whenever fr >= 2^-8, and is
    // truncated to one otherwise: the result is x * 2^32 rounded towards -inf, with float-only arithmetic
    const float hi = floorf(x), fr = x - hi;
    // (fr can round up to 1.0f for tiny negative x: clamp below 2^32 so the conversion stays defined)
    return ((long long)(int)hi << 32) + (long long)(unsigned int)fminf(fr * 4294967296.0f, 4294967040.0f);
}
RT_DEV float rtFromFixed(long long a) { return (float)((double)a * (1.0 / RT_ACC_SCALE)); }
